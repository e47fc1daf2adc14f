"""Lint of the device assembly for the hazard the hand-written asynchronous LDS reads carry.

The pipelined GEMM kernels issue their fragment reads as inline-asm `ds_read_b128` and wait for them with hand-counted
`s_waitcnt lgkmcnt(N)`.  The compiler believes an asm output is complete the moment the statement ends, so it is free to read (copy,
spill, use) or overwrite the destination registers BEFORE the data has landed -- it did exactly that when a kernel grew a second copy
of its k-loop (register moves at the loop entry).  This script replays the LGKM queue over the generated code:

  * every inline-asm `ds_read*` (between ;;#ASMSTART / ;;#ASMEND) enters the in-order queue with its destination registers;
    compiler-issued LDS / SMEM operations enter it too (they count in lgkmcnt; SMEM returns out of order, so only lgkmcnt(0) clears a
    queue that holds one);
  * `s_waitcnt lgkmcnt(N)` retires all but the N youngest entries;
  * any instruction that reads or writes a register of a still-pending inline-asm read is reported (except another LDS read that
    only overwrites it: LDS returns in order, so the younger data lands last -- that is how the compiler recycles the registers of
    fragment reads whose results a tail copy of the loop never uses);
  * the queue is dropped at labels (join points: the other predecessors are checked on their own paths).
Third rule: no `s_barrier` while an LDS write of the wave is still in the queue (a raw barrier carries no wait of its own).
Second rule: an LDS-DMA instruction (`global_load_lds_*`) must not issue directly behind a write of M0 (the compiler's hazard
recogniser guarantees that for the builtin form, not inside inline asm).

Fourth rule (round 4, igemm_wreg.hip): the same replay for the VM queue.  The weights-in-registers GEMM issues its weight fragments as
inline-asm `global_load_dwordx4` and waits for them with hand-counted `s_waitcnt vmcnt(N)`; every VMEM operation (loads, LDS-DMA,
stores, atomics) enters the in-order queue, asm loads with their destination registers, `vmcnt(N)` retires all but the N youngest,
and any instruction that touches a pending asm destination is reported (a younger VMEM load that only overwrites it is consistent:
returns are in order).  Unlike the LGKM queue this one survives labels -- the weights stay in flight across loop back edges: at a
label reached by fall-through the state is kept, behind an unconditional branch it is the state of the first forward branch to that
label, and every backward branch replays its loop body once with the state it arrives with (the steady state of the k-loop).

Fifth rule (round 5): the write-through publication protocol (lint_publication): every `sc0 sc1` store is covered by `vmcnt(0)` before the next
atomic RMW (the ticket), and a publishing function has an `sc1`-load / `buffer_inv` reader path.

Sixth rule (round 5): no `scratch_*` (spill) traffic between the first and the last MFMA of a kernel that carries inline asm (lint_scratch): a spill is a
VMEM operation and would shift the hand-counted `vmcnt` waits.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -x hip --cuda-device-only -S csrc/igemm_glds.hip -o /tmp/glds.s
    python tools/asm_lint.py /tmp/glds.s            -> exit status 1 if anything is reported
"""
import re
import sys

REG = re.compile(r"\b([vs])(?:\[(\d+):(\d+)\]|(\d+)\b)")
NO_DST = ("global_store", "buffer_store", "flat_store", "scratch_store", "ds_write", "s_cbranch", "s_branch", "s_barrier", "s_nop", "s_waitcnt",
          "s_endpgm", "s_setprio", "s_sleep", "s_cmp", "s_bitcmp", "global_load_lds", "buffer_wbl2", "buffer_inv", "s_sendmsg", "s_setreg")


def regs(op):
    out = set()
    for m in REG.finditer(op):
        kind = m.group(1)
        if m.group(4) is not None:
            out.add((kind, int(m.group(4))))
        else:
            out.update((kind, r) for r in range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def lint(path):
    findings, kernel, in_asm, queue, prev_writes_m0 = [], None, False, [], False      # queue entries: dict(dst=set, asm=bool, smem=bool, line=int, text=str)
    for ln, raw in enumerate(open(path), 1):
        line = raw.strip()
        if not line:
            continue
        if line.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if line.startswith(";") or line.startswith("."):
            if re.match(r"\.LBB\d+_\d+:", line):
                queue = [e for e in queue if e.get("write")]       # (a pending LDS write stays pending whichever path joins here)
            continue
        if re.match(r"^[A-Za-z_][\w$.]*:", line):            # function label
            kernel, queue = line.split(":")[0], []
            continue
        code = line.split(";")[0].strip()
        if not code:
            continue
        parts = code.split(None, 1)
        mn, ops = parts[0], [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        if mn == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", code)
            if m:
                n = int(m.group(1))
                if n == 0:
                    queue = []
                else:
                    # LDS operations return in order among themselves, SMEM in any order: lgkmcnt(N) leaves at most N operations
                    # outstanding, hence at most N LDS operations -- the oldest LDS entries beyond that have landed whatever the
                    # SMEM entries did; the SMEM entries themselves stay pending until lgkmcnt(0)
                    lds = [e for e in queue if not e["smem"]]
                    done = set(id(e) for e in lds[:max(0, len(lds) - n)])
                    queue = [e for e in queue if id(e) not in done]
            continue
        # third rule: a raw s_barrier waits for nothing -- an LDS write of this wave that is still in the queue may not have landed when
        # the barrier releases its readers
        if mn == "s_barrier":
            w = next((e for e in queue if e.get("write")), None)
            if w:
                findings.append(f"{kernel}: line {ln}: s_barrier with the LDS write `{w['text']}` (line {w['line']}) still pending: no lgkmcnt wait between them")
            continue
        if mn == "s_endpgm":
            queue = []
            continue
        # second rule: an LDS-DMA instruction must not issue in the wait state right behind a write of M0
        if mn.startswith("global_load_lds") or (mn.startswith("buffer_load") and " lds" in code):
            if prev_writes_m0:
                findings.append(f"{kernel}: line {ln}: `{code}` issues directly behind a write of M0 (one wait state is required)")
        prev_writes_m0 = bool(ops) and ops[0] == "m0" and mn.startswith("s_")
        has_dst = not mn.startswith(NO_DST)
        written = regs(ops[0]) if (has_dst and ops) else set()
        read = set().union(*[regs(o) for o in (ops[1:] if has_dst else ops)]) if ops else set()
        is_lds_read = mn.startswith(("ds_read", "ds_bpermute", "ds_swizzle", "ds_permute"))
        for e in queue:
            # (an LDS read that only OVERWRITES a pending destination is consistent: LDS returns in order, the younger data lands last)
            hit = (read & e["dst"]) | (set() if is_lds_read else (written & e["dst"]))
            if e["asm"] and hit:
                what = "reads" if read & e["dst"] else "overwrites"
                findings.append(f"{kernel}: line {ln}: `{code}` {what} {sorted(r for r in (read | written) & e['dst'])[:4]} of the inline-asm "
                                f"`{e['text']}` (line {e['line']}) before any lgkmcnt wait covers it")
                break
        if mn.startswith(("ds_read", "ds_bpermute", "ds_swizzle", "ds_permute")):
            queue.append(dict(dst=written, asm=in_asm, smem=False, line=ln, text=code))
        elif mn.startswith(("ds_write", "ds_add", "ds_max", "ds_min", "ds_or", "ds_and")):
            queue.append(dict(dst=set(), asm=in_asm, smem=False, line=ln, text=code, write=True))
        elif mn.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime")):
            queue.append(dict(dst=set(), asm=in_asm, smem=True, line=ln, text=code))
    return findings


VMEM = ("global_load", "global_store", "global_atomic", "buffer_load", "buffer_store", "buffer_atomic", "flat_load", "flat_store",
        "flat_atomic", "scratch_load", "scratch_store")


def lint_vm(path):
    """fourth rule: pending inline-asm global loads vs hand-counted vmcnt (see the module docstring)"""
    funcs, cur, in_asm = [], None, False
    for ln, raw in enumerate(open(path), 1):
        line = raw.strip()
        if not line:
            continue
        if line.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"(\.LBB\d+_\d+):", line)
        if m:
            if cur is not None:
                cur["ins"].append(dict(label=m.group(1), line=ln))
            continue
        if line.startswith(";") or line.startswith("."):
            continue
        m = re.match(r"^([A-Za-z_][\w$.]*):", line)
        if m:
            cur = dict(name=m.group(1), ins=[])
            funcs.append(cur)
            continue
        code = line.split(";")[0].strip()
        if code and cur is not None:
            parts = code.split(None, 1)
            cur["ins"].append(dict(mn=parts[0], ops=[o.strip() for o in parts[1].split(",")] if len(parts) > 1 else [], code=code, line=ln, asm=in_asm))
    findings = []
    for f in funcs:
        ins = f["ins"]
        if not any(i.get("asm") and i.get("mn", "").startswith("global_load") for i in ins):
            continue
        label_at = {i["label"]: k for k, i in enumerate(ins) if "label" in i}
        loop_heads = set()
        for k, i in enumerate(ins):
            if i.get("mn", "").startswith(("s_cbranch", "s_branch")) and i["ops"] and i["ops"][0] in label_at and label_at[i["ops"][0]] < k:
                loop_heads.add(i["ops"][0])
        replayed, seen = set(), set()

        def step(k, q):
            i = ins[k]
            mn, ops, code = i["mn"], i["ops"], i["code"]
            if mn == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", code)
                if m:
                    n = int(m.group(1))
                    del q[:max(0, len(q) - n)]
                return
            has_dst = not mn.startswith(NO_DST)
            written = regs(ops[0]) if (has_dst and ops) else set()
            read = set().union(*[regs(o) for o in (ops[1:] if has_dst else ops)]) if ops else set()
            is_vm = mn.startswith(VMEM)
            is_vm_load = is_vm and "_load" in mn and "lds" not in mn
            for e in q:
                hit = (read & e["dst"]) | (set() if is_vm_load else (written & e["dst"]))
                if e["asm"] and hit:
                    key = (i["line"], e["line"])
                    if key not in seen:
                        seen.add(key)
                        what = "reads" if read & e["dst"] else "overwrites"
                        findings.append(f"{f['name']}: line {i['line']}: `{code}` {what} {sorted(hit)[:4]} of the inline-asm `{e['text']}` "
                                        f"(line {e['line']}) before any vmcnt wait covers it")
                    break
            if is_vm:
                q.append(dict(dst=written if (is_vm_load and i["asm"]) else set(), asm=i["asm"] and is_vm_load, line=i["line"], text=code))

        # join points of forward branches drop the queue (as the LGKM rule does: which of the predecessors ran is not known, and a
        # guessed one reports hazards on paths that cannot execute); a loop header keeps the state it is entered with, and every
        # backward branch replays its body once with the state it arrives with -- the steady state of the k-loop
        def walk(k0, k1, q, top):
            k = k0
            while k < k1:
                i = ins[k]
                if "label" in i:
                    if i["label"] not in loop_heads or k != k0 and not top:
                        q[:] = []
                    elif i["label"] not in loop_heads:
                        q[:] = []
                    k += 1
                    continue
                mn = i["mn"]
                if mn in ("s_endpgm", "s_setpc_b64", "s_branch"):
                    q[:] = []
                elif mn.startswith("s_cbranch"):
                    tgt = i["ops"][0] if i["ops"] else None
                    if tgt in label_at and label_at[tgt] < k and top and k not in replayed:
                        replayed.add(k)
                        walk(label_at[tgt] + 1, k, [dict(e) for e in q], False)
                else:
                    step(k, q)
                k += 1

        walk(0, len(ins), [], True)
    return findings


def lint_publication(path):
    """Fifth rule (round 5): the fence-free cross-workgroup publication of round 4 (attention key halves, split-K slabs).  A workgroup publishes with
    inline-asm WRITE-THROUGH stores (`global_store_* ... sc0 sc1`), waits for them (`s_waitcnt vmcnt(0)`), then draws its ticket (an atomic RMW); the
    reader takes the partner's image with L1-bypassing loads (`global_load_* ... sc1`) or behind an acquire fence.  Checked per function, in program order:
      * between the last write-through store and the next atomic RMW there is an `s_waitcnt vmcnt(0)` -- a ticket drawn over stores still in flight
        publishes bytes that have not left;
      * a function that publishes this way and also holds a ticket reader path contains at least one `sc1` load or a `buffer_inv` / acquire
        (`buffer_inv sc1`, or the compiler's acquire sequence) behind an atomic -- a plain load there could be served by a stale L1 line."""
    findings, kernel = [], None
    pending, saw_wt, saw_atomic_after_wt, saw_reader = None, False, False, False

    def close():
        if kernel and saw_wt and saw_atomic_after_wt and not saw_reader:
            findings.append(f"{kernel}: publishes with write-through stores and draws a ticket, but no `sc1` load / `buffer_inv` reader path was found")

    for ln, raw in enumerate(open(path), 1):
        line = raw.strip()
        if not line or line.startswith(";") or line.startswith("."):
            continue
        m = re.match(r"^([A-Za-z_][\w$.]*):", line)
        if m:
            close()
            kernel, pending, saw_wt, saw_atomic_after_wt, saw_reader = m.group(1), None, False, False, False
            continue
        code = line.split(";")[0].strip()
        if not code:
            continue
        mn = code.split(None, 1)[0]
        if mn.startswith(("global_store", "flat_store")) and " sc0" in code and " sc1" in code:
            pending, saw_wt = (ln, code), True
        elif mn == "s_waitcnt" and re.search(r"vmcnt\(0\)", code):
            pending = None
        elif mn.startswith(("global_atomic", "flat_atomic", "buffer_atomic")):
            if pending is not None:
                findings.append(f"{kernel}: line {ln}: `{code}` (ticket) issues with the write-through store `{pending[1]}` (line {pending[0]}) not covered "
                                f"by an `s_waitcnt vmcnt(0)`")
                pending = None
            if saw_wt:
                saw_atomic_after_wt = True
        elif saw_atomic_after_wt and ((mn.startswith(("global_load", "flat_load")) and " sc1" in code) or mn.startswith("buffer_inv")):
            saw_reader = True
    close()
    return findings



def lint_scratch(path):
    """Sixth rule (round 5): no scratch traffic inside the matrix loop of a kernel with hand-counted waits.  A spill load or store is a VMEM operation:
    it enters the VM queue the hand-counted `s_waitcnt vmcnt(N)` of the DMA rings index into, so a `scratch_*` between the first and the last MFMA of a
    function that carries inline asm shifts every count behind it (and costs a memory round trip per k-step).  Spills the compiler places in a prologue
    or epilogue -- it waits for those itself -- are reported by tools/kernel_resources.py, not here."""
    findings = []
    kernel, has_asm, first_mfma, last_mfma, scratch = None, False, None, None, []

    def close():
        if kernel and has_asm and first_mfma is not None:
            for ln, code in scratch:
                if first_mfma < ln < last_mfma:
                    findings.append(f"{kernel}: line {ln}: `{code}` inside the matrix loop (MFMAs at lines {first_mfma} ... {last_mfma}) of a kernel with hand-counted waits")

    for ln, raw in enumerate(open(path), 1):
        line = raw.strip()
        if line.startswith(";;#ASMSTART"):
            has_asm = True
            continue
        if not line or line.startswith(";") or line.startswith("."):
            continue
        m = re.match(r"^([A-Za-z_][\w$.]*):", line)
        if m:
            close()
            kernel, has_asm, first_mfma, last_mfma, scratch = m.group(1), False, None, None, []
            continue
        code = line.split(";")[0].strip()
        if not code:
            continue
        mn = code.split(None, 1)[0]
        if mn.startswith("v_mfma"):
            first_mfma = ln if first_mfma is None else first_mfma
            last_mfma = ln
        elif mn.startswith("scratch_"):
            scratch.append((ln, code))
    close()
    return findings

if __name__ == "__main__":
    bad = []
    for p in sys.argv[1:]:
        f = lint(p) + lint_vm(p) + lint_publication(p) + lint_scratch(p)
        print(f"{p}: {len(f)} finding(s)")
        for x in f[:40]:
            print("  " + x)
        bad += f
    sys.exit(1 if bad else 0)

"""Instruction mix of the loops of one kernel in a hipcc -S listing: per basic block that sits in a loop, the instruction counts.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -x hip --cuda-device-only -S csrc/attention.hip -o /tmp/a.s
    python tools/asm_loop_mix.py /tmp/a.s attn_d64_v2_kernelILi3
"""
import collections, re, sys
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(key), l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
blk, inloop, blocks = None, False, collections.OrderedDict()
for l in lines[start:end]:
    t = l.strip()
    m = re.match(r"(\.LBB\d+_\d+):(.*)", t)
    if m:
        blk = m.group(1); inloop = "Loop" in m.group(2); blocks[blk] = (inloop, [])
        continue
    if not t or t.startswith(";") or t.startswith(".") or blk is None: continue
    blocks[blk][1].append(t.split()[0])
tot = collections.Counter()
for b, (lp, ins) in blocks.items():
    if not lp: continue
    c = collections.Counter(ins)
    v = sum(n for k, n in c.items() if k.startswith("v_") and "mfma" not in k)
    print(f"{b:10s} {len(ins):4d} instructions, {v:3d} VALU (no MFMA): " + ", ".join(f"{k} {n}" for k, n in c.most_common(14)))

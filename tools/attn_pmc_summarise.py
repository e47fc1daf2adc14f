"""Per-kernel means of the counters tools/attn_pmc.sh collects (one row per kernel name; values are per DISPATCH, summed over the
chip as rocprofv3 reports them), plus the ratios that say what a SIMD spends its time on."""
import collections, csv, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sys.argv[1:]:
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "attn" not in r["Kernel_Name"]: continue
        per[(r["Kernel_Name"], r.get("Dispatch_Id", ""))][r["Counter_Name"]] += float(r["Counter_Value"])
    for (k, _), d in per.items():
        for c, v in d.items(): agg[k][c].append(v)
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    print(k[:90])
    for c in sorted(m): print(f"   {c:32s} {m[c]:16.0f}")
    g = m.get("GRBM_GUI_ACTIVE")
    if g:
        simd_cycles = g / 8.0 * 1024.0          # shader cycles of the launch x 1024 SIMDs
        print(f"   launch = {g/8:.0f} shader cycles;  per SIMD-cycle of the launch:")
        for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_VALU_MFMA_COEXEC_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_ANY",
                  "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_DATA_FIFO_FULL", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES"):
            if c in m: print(f"      {c:30s} {m[c]/simd_cycles:8.3f}")

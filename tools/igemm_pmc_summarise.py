"""Per (kernel, grid) means of the counters tools/igemm_pmc.sh collects, normalised per SIMD-cycle of the launch."""
import collections, csv, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sys.argv[1:]:
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "igemm" not in r["Kernel_Name"]: continue
        per[(r["Kernel_Name"], r.get("Grid_Size", ""), r.get("Dispatch_Id", ""))][r["Counter_Name"]] += float(r["Counter_Value"])
    for (k, g, _), d in per.items():
        for c, v in d.items(): agg[(k, g)][c].append(v)
Q = ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES")
for (k, g), d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    ga = m.get("GRBM_GUI_ACTIVE")
    print(f"{k[:70]} grid {g}  dispatches {len(next(iter(d.values())))}")
    if not ga: continue
    sc = ga / 8.0 * 1024.0
    print(f"   launch {ga/8:.0f} shader cycles; MFMA insts {m.get('SQ_INSTS_MFMA', 0):.0f}, VALU insts {m.get('SQ_INSTS_VALU', 0):.0f}, SALU {m.get('SQ_INSTS_SALU', 0):.0f}, LDS {m.get('SQ_INSTS_LDS', 0):.0f}")
    print("   per SIMD-cycle: mfma_busy %.3f coexec %.3f | " % (m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / sc, m.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0) / sc) +
          " ".join(f"{c[3:].lower()} {4 * m[c] / sc:.3f}" for c in Q if c in m) + f" | lds_conflict/lds_active {m.get('SQ_LDS_BANK_CONFLICT', 0) / max(1.0, m.get('SQ_LDS_IDX_ACTIVE', 1)):.2f}")

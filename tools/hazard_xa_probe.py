"""DESIGN 9.2 follow-up (verdict r3 item 6): with hazards 2 and 3 fixed, does the ORIGINAL 64-lane form of the fused cross-attention
epilogue's per-column vector loads still deliver a wrong value now and then?  Measure build (SDXL_MEASURE_LIB=1), knob xa_vec64:
repeated launches of the fused projection on identical inputs, bit-compared -- 0 = scalar-cache form (production), 1 = 64-lane form.
    SDXL_MEASURE_LIB=1 python tools/hazard_xa_probe.py [launches per setting]"""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package(); ctx = pkg.Context(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
def seeded(*s, seed): return torch.randn(*s, generator=torch.Generator().manual_seed(seed)).cuda()
for (B, Nq, C) in ((2, 4096, 640), (2, 1024, 1280)):          # the 256x128 and 96x128 fused projections of the CFG pair
    x = seeded(B, Nq, C, seed=12); g, be = 1 + 0.1 * seeded(C, seed=13), 0.1 * seeded(C, seed=14)
    wq = seeded(C, C, seed=15) / math.sqrt(C); k, v = seeded(B, 77, C, seed=16), seeded(B, 77, C, seed=17)
    for knob in (0, 1, 0, 1):
        pkg.debug_set("xa_vec64", knob)
        ref, bad, worst = None, 0, 0.0
        for r in range(N):
            if r % 16 == 0:        # disturb the caches between bursts, as the original probe did
                junk = torch.full((64 << 20,), float("nan"), device="cuda"); del junk
            o = pkg.ln_query_cross_attention(ctx, x, g, be, wq, k, v, 1e-5, True)[0]
            if ref is None: ref = o.clone()
            elif not torch.equal(o, ref):
                bad += 1; worst = max(worst, float((o - ref).abs().max()))
        print(f"B{B} Nq{Nq} C{C} xa_vec64={knob}: {bad} of {N} launches differ from the first (max |diff| {worst:.3e})", flush=True)
pkg.debug_set("xa_vec64", 0)

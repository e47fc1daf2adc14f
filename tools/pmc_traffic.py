"""Turns the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE csv files of `tools/profile_step.py` into profiles/<tag>_pmc_traffic.json:
HBM-side bytes of the implicit-GEMM kernels per UNet CFG step, corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE is
reported at half the bytes of wide coalesced reads on gfx950 -> doubled; WRITE_SIZE taken as reported, uncalibrated).
usage: pmc_traffic.py out.json fetch.csv write.csv n_steps_profiled"""
import csv, json, sys, collections
out, fcsv, wcsv, nsteps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
def load(path, counter):
    tot = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter: continue
        k = r["Kernel_Name"].split("(")[0]
        tot[k] += float(r["Counter_Value"]); n[k] += 1
    return tot, n
f, fn = load(fcsv, "FETCH_SIZE"); w, wn = load(wcsv, "WRITE_SIZE")
# kernels that run once per MODEL BUILD / per prompt (weight packing, synthetic fill, context caches), not once per step: they are
# listed apart, with their whole-run totals -- dividing them by the step count (r4) printed "17.4 GB per step" for pack_linear_kernel
BUILD = ("pack_", "synth_fill", "repack_wfrag", "colsum_packed", "beta_dot", "absmax", "f16_exact", "round_f16", "xattn_pack", "fill_zero")
res = {"unit": "bytes per UNet CFG step (B=2, 1024x1024)", "steps_profiled": nsteps, "kernels": {},
       "model_build_kernels": {"unit": "bytes over the whole profiled run (once per model build / prompt, NOT per step)"}}
tf = tw = 0.0; launches = 0
for k in sorted(f, key=lambda k: -f[k]):
    if any(b in k for b in BUILD):
        res["model_build_kernels"][k] = {"launches": fn[k], "fetch_bytes": 2.0 * f[k] * 1024.0, "write_bytes": w.get(k, 0.0) * 1024.0}
        continue
    fb = 2.0 * f[k] * 1024.0 / nsteps      # FETCH_SIZE is in KB; x2 gfx950 correction
    wb = w.get(k, 0.0) * 1024.0 / nsteps
    res["kernels"][k] = {"launches_per_step": fn[k] / nsteps, "fetch_bytes": fb, "write_bytes": wb}
    if "igemm" in k: tf += fb; tw += wb; launches += fn[k] / nsteps
res["igemm_total"] = {"launches_per_step": launches, "fetch_bytes": tf, "write_bytes": tw,
                      "bytes_per_launch": (tf + tw) / max(launches, 1)}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res["igemm_total"]))

"""Instruction-cache behaviour per kernel family of one profiled UNet CFG step (rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES
SQC_ICACHE_MISSES_DUPLICATE [SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES ...] -- tools/profile_step.py):
    python tools/icache_summarise.py <counter_collection.csv> [...] > gpurun_out/r06_icache.txt
For every kernel name (template arguments kept, parameter list cut): dispatches and per counter the sum over dispatches; miss rate = MISSES / REQ (a request is one
64-byte line fetch), misses per dispatch and per workgroup-wave are what tell a kernel whose loop does not fit the 64 KiB instruction cache (it misses on every
iteration: misses grow with the trip count) from one that only takes the cold misses of its first pass (misses ~ code size / 64 B per CU-group, independent of K)."""
import collections, csv, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"])[:110]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
cs = sorted({c for d in agg.values() for c in d})
print(f"{'kernel':110s} {'disp':>5s} " + " ".join(f"{c[-22:]:>22s}" for c in cs) + "  miss_rate  misses/dispatch")
for k in sorted(agg, key=lambda k: -agg[k].get("SQC_ICACHE_REQ", 0.0)):
    d = agg[k]; req, mis = d.get("SQC_ICACHE_REQ", 0.0), d.get("SQC_ICACHE_MISSES", 0.0)
    print(f"{k:110s} {len(n[k]):5d} " + " ".join(f"{d.get(c, 0.0):22.0f}" for c in cs) + f"  {mis / req if req else 0.0:9.4f}  {mis / max(len(n[k]), 1):12.0f}")

"""Kernel-to-kernel gaps of the replayed sampling loop, from a rocprofv3 --kernel-trace CSV.
usage: trace_gaps.py <kernel_trace.csv> [out.json]
Sorts the dispatches by start time and reports, over the busiest contiguous window (the timed image): the summed kernel
durations, the summed idle gaps between one kernel's end and the next one's start, and the gap histogram."""
import csv, json, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
n = len(rows)
# last 40 % of the dispatches = the timed image(s) of bench.py (warm-up first)
seg = rows[int(n * 0.6):]
busy = sum(e - s for s, e, _ in seg)
gaps = [max(0, seg[i + 1][0] - seg[i][1]) for i in range(len(seg) - 1)]
overlap = sum(1 for i in range(len(seg) - 1) if seg[i + 1][0] < seg[i][1])
span = seg[-1][1] - seg[0][0]
hist = {}
for g in gaps:
    k = "<1us" if g < 1000 else "1-2us" if g < 2000 else "2-4us" if g < 4000 else "4-8us" if g < 8000 else "8-50us" if g < 50000 else ">50us"
    hist[k] = hist.get(k, 0) + 1
by = {}
for i, g in enumerate(gaps):
    nm = seg[i + 1][2].split("(")[0][:60]
    a = by.setdefault(nm, [0, 0]); a[0] += 1; a[1] += g
top = sorted(by.items(), key=lambda kv: -kv[1][1])[:12]
out = {"dispatches": len(seg), "span_ms": span / 1e6, "busy_ms": busy / 1e6, "gap_ms": sum(gaps) / 1e6,
       "gap_frac_of_span": sum(gaps) / span, "median_gap_ns": sorted(gaps)[len(gaps) // 2], "overlapping_pairs": overlap,
       "gap_histogram": hist, "gap_before_kernel_top": [{"kernel": k, "n": v[0], "gap_ms": v[1] / 1e6, "avg_ns": v[1] / v[0]} for k, v in top]}
s = json.dumps(out, indent=1)
print(s)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(s)

"""Per-kernel duration summary of ONE replayed UNet step (between two ddim_kernel dispatches) from a rocprofv3 kernel trace.
usage: trace_step_summary.py <kernel_trace.csv>"""
import csv, sys
from collections import defaultdict
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Grid_Size_X", "")))
rows.sort()
idx = [i for i, r in enumerate(rows) if r[2].endswith("ddim_kernel")]
a, b = idx[-3], idx[-2]
seg = rows[a + 1:b + 1]
print("step span %.3f ms, %d dispatches" % ((seg[-1][1] - rows[a][1]) / 1e6, len(seg)))
by = defaultdict(list)
for s, e, n, g in seg:
    by[(n[-64:], g)].append((e - s) / 1e3)
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print("%4d  %9.1f us  avg %7.1f  min %7.1f  %s grid %s" % (len(v), sum(v), sum(v) / len(v), min(v), k[0], k[1]))

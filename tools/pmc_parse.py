"""prints per-kernel averages of every rocprofv3 counter_collection csv given on the command line"""
import csv, collections, sys
for f in sys.argv[1:]:
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in agg.items():
        if 'igemm' not in k and 'attn' not in k: continue
        print(f, k, {c: round(sum(v) / len(v)) for c, v in d.items()}, 'n=', len(next(iter(d.values()))))

"""per-shape A/B of one UNet step's launches between two debug-knob settings: python tools/launch_ab.py a.csv b.csv"""
import csv, collections, sys
def load(p):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(p)):
        k = (r['class'], r['M'], r['N'], r['K'], r['ksize'])
        agg[k][0] += 1; agg[k][1] += float(r['ms'])
    return agg
a, b = load(sys.argv[1]), load(sys.argv[2])
tot = [0.0, 0.0]
rows = []
for k in a:
    if k in b:
        rows.append((b[k][1] - a[k][1], k, a[k][0], a[k][1], b[k][1]))
        tot[0] += a[k][1]; tot[1] += b[k][1]
for d, k, n, ta, tb in sorted(rows):
    print(f"{k} x{n}: {ta*1e3/n:7.1f} -> {tb*1e3/n:7.1f} us  (total {d*1e3:+8.1f} us)")
print(f"total {tot[0]:.3f} -> {tot[1]:.3f} ms")

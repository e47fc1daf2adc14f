"""aggregates a SDXL_PROFILE_DUMP per-launch csv (class,M,N,K,ksize,ms) by shape"""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
agg = collections.OrderedDict()
for r in rows:
    k = (r['class'], r['M'], r['N'], r['K'], r['ksize'])
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += float(r['ms'])
tot = sum(a[1] for a in agg.values())
print("total ms", round(tot, 3))
for k, a in sorted(agg.items(), key=lambda x: -x[1][1])[:top]:
    cls, M, N, K, ks = k
    M, N, K = int(M), int(N), int(K)
    fl = 2.0 * M * N * K if cls == '0' else (4.0 * M * N * K * 64 if cls == '1' else 0)
    print(f"cls{cls} M={M:6d} N={N:6d} K={K:6d} ks={ks}  n={a[0]:4d}  ms={a[1]:7.3f} ({100*a[1]/tot:4.1f}%)  avg={1e3*a[1]/a[0]:7.1f}us  TF/s={fl*a[0]/a[1]/1e9 if a[1] else 0:7.0f}")

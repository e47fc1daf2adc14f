"""Per-kernel sums of rocprofv3 --pmc counter files (tools/wreg_tcc_ab.sh): arguments are knob:path pairs; prints, per knob setting and kernel family,
launches and the counter totals per launch; derived: L2 hit rate = TCC_HIT / (TCC_HIT + TCC_MISS), fetched bytes = 2 x FETCH_SIZE KB (gfx950 correction,
MI355X_MICROARCH.md 'HBM')."""
import collections, csv, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for a in sys.argv[1:]:
    knob, path = a.split(":", 1)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        fam = ("wreg<96>" if "igemm_wreg_kernel<96" in k or "igemm_wreg_kernelILi96" in k else "wreg<64>" if "igemm_wreg" in k else "igemm_wide" if "igemm_wide" in k
               else "igemm_pipe" if "igemm_pipe" in k else "attention" if "attn" in k else None)
        if fam is None: continue
        acc[(knob, fam)][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(knob, fam)][r["Counter_Name"]] += 1
for key in sorted(acc):
    c, n = acc[key], cnt[key]
    line = f"wreg_xcd2d={key[0]} {key[1]:11s}"
    for name in sorted(c):
        line += f" | {name} {c[name] / max(n[name], 1):.4g}/launch x{n[name]}"
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        line += f" | L2 hit rate {c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.3f}"
    if "FETCH_SIZE" in c:
        line += f" | fetched {2.0 * c['FETCH_SIZE'] * 1024 / max(n['FETCH_SIZE'], 1) / 1e6:.1f} MB/launch"
    print(line)

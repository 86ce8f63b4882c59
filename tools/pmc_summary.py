import csv, glob, sys, collections
root = sys.argv[1]
for f in sorted(glob.glob(root + '/pmc_*/*/*_counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:60]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in agg.items():
        if not any(x in k for x in ('k_p2g', 'k_g2p', 'k_grid', 'k_vol')): continue
        print(k, {c: f"{sum(v)/len(v):.4g}" for c, v in cs.items()}, 'n=', len(next(iter(cs.values()))))

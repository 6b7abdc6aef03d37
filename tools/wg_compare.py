import csv, collections, sys
res={}
for tag in sys.argv[1:]:
    rows=list(csv.DictReader(open(f'gpurun_out/ops_wg{tag}.csv')))
    agg=collections.defaultdict(float)
    for r in rows:
        if r['op']=='WGRAD': agg[r['i']]+=float(r['ms'])
    res[tag]=agg
keys=sorted(res[sys.argv[1]], key=lambda k:-res[sys.argv[1]][k])
print('shape'.ljust(60), *[t.rjust(8) for t in sys.argv[1:]])
for k in keys: print(k.ljust(60), *[f"{res[t].get(k,0):8.3f}" for t in sys.argv[1:]])
print('total'.ljust(60), *[f"{sum(res[t].values()):8.3f}" for t in sys.argv[1:]])

import csv, glob, collections, sys
O = sys.argv[1]; pat = sys.argv[2]
for d in sorted(glob.glob(f'{O}/*/')):
    fs = glob.glob(d + '*/*_counter_collection.csv')
    if not fs: continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if pat in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for c, v in sorted(agg.items()): print(f'{c:40s} n={len(v):4d} mean={sum(v)/len(v):18.1f}')

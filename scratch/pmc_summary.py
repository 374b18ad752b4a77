import csv, sys, glob, collections
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_traverse"
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: [0.0, 0])
for fn in f:
    for r in csv.DictReader(open(fn)):
        if pat in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (v, c) in sorted(acc.items()):
    print(f"{k:32s} per-dispatch {v / c:16.1f}   (dispatches {c})")

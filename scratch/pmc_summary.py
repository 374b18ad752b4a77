"""Per-dispatch mean of every counter of one rocprofv3 --pmc run, for the kernels whose name
contains a pattern.  With --steady only dispatches of steady-state steps count: everything
before the second k_integrate dispatch (the spawn's sort/build and the first, cold step) is
dropped.  --json prints {"counter": mean, ..., "dispatches": n}."""
import csv, sys, glob, collections, json

args = [a for a in sys.argv[1:] if not a.startswith("--")]
steady = "--steady" in sys.argv
as_json = "--json" in sys.argv
d = args[0]
pat = args[1] if len(args) > 1 else "k_traverse"
rows = []
for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(fn)))
first = 0
if steady:
    ids = sorted({int(r["Dispatch_Id"]) for r in rows if "k_integrate" in r["Kernel_Name"] or "k64_integrate" in r["Kernel_Name"]})
    if len(ids) < 2:
        sys.exit("--steady needs at least two steps in the run")
    first = ids[1]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in rows:
    if pat in r["Kernel_Name"] and int(r["Dispatch_Id"]) >= first:
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
if as_json:
    print(json.dumps({**{k: v / c for k, (v, c) in acc.items()}, "dispatches": max([c for _, c in acc.values()] or [0])}))
else:
    for k, (v, c) in sorted(acc.items()):
        print(f"{k:32s} per-dispatch {v / c:16.1f}   (dispatches {c})")

import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in sys.argv[2:]):
        print(f"   {r['Name'][:36]:36s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us")

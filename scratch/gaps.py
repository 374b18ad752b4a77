import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# steps delimited by k_integrate_morton
gaps = collections.defaultdict(list); busy = []; step_gap = []
cur_busy = cur_gap = 0; prev_end = None; prev_name = None; started = False
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void nb::", "").replace("nb::", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if "k_integrate_morton" in name:
        if started: busy.append(cur_busy); step_gap.append(cur_gap)
        started = True; cur_busy = cur_gap = 0; prev_end = None
    if not started: continue
    if prev_end is not None:
        g = s - prev_end
        gaps[prev_name[:28] + " -> " + name[:28]].append(g); cur_gap += max(g, 0)
    cur_busy += e - s; prev_end = e; prev_name = name
n = len(busy)
print(f"steps {n}: mean kernel-busy {sum(busy)/n/1e3:.1f} us, mean inter-kernel gap total {sum(step_gap)/n/1e3:.1f} us")
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:25]:
    print(f"{k:62s} n={len(v):4d} mean gap {sum(v)/len(v)/1e3:7.2f} us")

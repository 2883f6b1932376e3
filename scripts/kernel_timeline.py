"""Dev aid: one iteration of a rocprofv3 --kernel-trace CSV as a timeline (start, gap to the previous kernel, duration, name).
    rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python bench.py --workload c2 --steps 10 --warmup 3 --no-cpu-baseline --no-profile
    python scripts/kernel_timeline.py /tmp/prof [launches of the heaviest kernel per iteration]"""
import csv, sys, glob, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find the period: the index distance between the last two launches of the most expensive kernel
names = [r["Kernel_Name"] for r in rows]
dur = collections.Counter()
for r in rows: dur[r["Kernel_Name"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
top = dur.most_common(1)[0][0]
dmax = max(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if r["Kernel_Name"] == top)
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"] == top and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 0.3 * dmax]   # (the same template also runs tiny products)
per = int(sys.argv[2]) if len(sys.argv) > 2 else 1
a, b = idx[-1 - 2 * per], idx[-1 - per]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f  +%6.1f gap  %8.1f us  %s  grid %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:90], "x".join(str(r[k]) for k in r if k.lower().startswith("grid"))))
    prev_end = e
print("period %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))

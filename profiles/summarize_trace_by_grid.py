#!/usr/bin/env python3
"""Per-(kernel, grid) durations from a rocprofv3 --kernel-trace --output-format csv run: one template often plays several roles in an iteration (a big pass and a
K x K Gram product share `fused_kernel<128, true, 0, true>`), which the per-name averages of --stats mix.  The first `skip` launches of every group are left out
of the steady-state columns (warm-up, clock ramp).
    python profiles/summarize_trace_by_grid.py <dir with *kernel_trace.csv> "<command>" [skip=40] > profiles/rN_xx_kernel_stats_by_grid.md"""
import collections
import csv
import glob
import statistics
import sys

d, cmd = sys.argv[1], sys.argv[2]
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 40
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
g = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    gk = [k for k in r if k.lower().startswith("grid")]
    grid = "x".join(str(r[k]) for k in gk) if gk else "?"
    g[(r["Kernel_Name"], grid)].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
print("# rocprofv3 --kernel-trace, durations per (kernel, grid)\n\ncommand: `%s`\n" % cmd)
print("| kernel | grid | calls | avg ms (all) | steady: calls | avg ms | median ms | min ms |\n|---|---|---|---|---|---|---|---|")
tot = sum(sum(x[1] for x in v) for v in g.values())
for (name, grid), v in sorted(g.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
    if "nmfx::" not in name or sum(x[1] for x in v) < 0.002 * tot:
        continue
    v.sort()
    dur = [x[1] for x in v]
    st = dur[skip:] if len(dur) > 2 * skip else dur
    print("| `%s` | %s | %d | %.4f | %d | %.4f | %.4f | %.4f |" % (name[:100], grid, len(dur), statistics.mean(dur), len(st), statistics.mean(st), statistics.median(st), min(st)))

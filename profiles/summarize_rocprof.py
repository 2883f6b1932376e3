#!/usr/bin/env python3
"""Turn a rocprofv3 results .db (rocpd sqlite, ROCm 7.2 default output) into the per-kernel summary we commit.

    python profiles/summarize_rocprof.py gpurun_out/prof_x/name_results.db "command line that was profiled" > profiles/r1_x.md
"""
import sqlite3
import sys


def main():
    db, cmd = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print("# rocprofv3 --kernel-trace --stats summary\n")
    if cmd:
        print("command: `%s`\n" % cmd)
    print("| kernel | calls | total ms | avg ms | % |")
    print("|---|---|---|---|---|")
    for name, calls, total, avg, pct in rows:
        print("| `%s` | %d | %.3f | %.4f | %.2f |" % (name[:110], calls, total / 1e3, avg / 1e3, pct))
    try:
        pmc = list(c.execute("select * from counters_collection limit 1"))
        if pmc:
            print("\n(counters present: see summarize_pmc)")
    except Exception:
        pass


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Per-kernel PMC averages from rocprofv3 results .db files (one --pmc pass each), as markdown.

    python profiles/summarize_pmc.py "<command>" fetch.db write.db sq.db > profiles/rN_xx_pmc.md

HBM bytes follow MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
exactly 1/2 of the bytes of a wide coalesced streaming read, so read bytes = 2 * FETCH_SIZE * 1024.
"""
import sqlite3
import sys
from collections import defaultdict


def main():
    cmd, dbs = sys.argv[1], sys.argv[2:]
    agg = defaultdict(dict)
    for db in dbs:
        c = sqlite3.connect(db)
        cols = [r[1] for r in c.execute("PRAGMA table_info(counters_collection)")]
        if "grid_size" in cols:   # one template can play several roles in an iteration (big pass, K x K Gram product): told apart by their grids
            q = ("select kernel_name || '  [grid ' || grid_size || ']', counter_name, count(*), avg(value) from counters_collection group by kernel_name, grid_size, counter_name")
        else:
            q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name")
        for k, cn, n, v in c.execute(q):
            agg[k][cn] = (n, v)
    print("# rocprofv3 PMC passes (separate runs per counter group)\n\ncommand: `%s`\n" % cmd)
    for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0))[1]):
        if "nmfx::" not in k:
            continue
        print("## `%s`\n" % (k[:120] + (k[k.rfind("  [grid"):] if "  [grid" in k and k.rfind("  [grid") >= 120 else "")))
        print("| counter | dispatches | mean per dispatch |\n|---|---|---|")
        for cn, (n, v) in sorted(d.items()):
            print("| %s | %d | %.6g |" % (cn, n, v))
        if "FETCH_SIZE" in d:
            rd = 2.0 * d["FETCH_SIZE"][1] * 1024.0
            wr = d.get("WRITE_SIZE", (0, 0.0))[1] * 1024.0
            print("\nHBM traffic per dispatch (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE): read %.4g B + write %.4g B = **%.4g B**" % (rd, wr, rd + wr))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "SQ_WAVE_CYCLES" in d:
            # SQ_WAVE_CYCLES counts quad-cycles per wave; with one wave per SIMD, 4*SQ_WAVE_CYCLES = SIMD-cycles
            util = d["SQ_VALU_MFMA_BUSY_CYCLES"][1] / (4.0 * d["SQ_WAVE_CYCLES"][1])
            print("\nMFMA pipe busy / (4 x SQ_WAVE_CYCLES) = **%.3f** (valid when one wave per SIMD is resident)" % util)
            tot = d["SQ_WAVE_CYCLES"][1]
            parts = ["%s %.1f%%" % (n_, 100.0 * d[n_][1] / tot) for n_ in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if n_ in d]
            print("\nwave-cycle split: " + ", ".join(parts))
        print()


if __name__ == "__main__":
    main()

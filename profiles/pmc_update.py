#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the PMC summaries of one run:   python profiles/pmc_update.py <tag, e.g. r6_29> [dir, default gpurun_out]   then   python profiles/pmc_stamp.py
Each bench tag is the HBM traffic per launch of one kernel (matched by a substring of its name in profiles/<tag>_<workload>_pmc.md), or the mean over two kernels'
launches where a tag brackets two passes (K > 192 dual maps: the S pass + the second-map pass; K > 256: the chain block + the last block)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, C = "fused:W-step (S=W*H -> R -> R*H')", "fused:H-step (S=W*H -> R -> W'*R + update)", "fused:cost pass (S=W*H -> D(V||S))"
N, G = "H-step numerator Gn=W'*A (two-operand GEMM, or the stationary kernel over V')", "gemm:Gp=W'*B"
GP = "gemm_pipe_kernel<128, 128, true, true, false, true, false>"
MAP = {
    "c3": {W: ["fused_kernel<256, true, 3, true,"], H: ["fused_kernel<256, false, 2, true,"], C: ["fused_kernel<256, true, 3, false,"]},
    "c2": {W: ["fused_kernel<128, true, 0, true, 0,"], N: ["fused_kernel<128, true, 0, true, 2,"], C: ["fused_kernel<128, true, 1, false,"]},
    "c4": {W: ["fused_kernel<512, true, 0, true, 0, true, 8>"], N: ["fused_kernel<128, true, 0, true, 2,"], C: ["fused_kernel<512, true, 1, false,"]},
    "c4kl": {W: ["fused_kernel<512, true, 0, true, 0, true, 8>"], C: ["fused_kernel<512, true, 3, false,"], N: [GP]},
    "c5": {"H-step terms": ["fused_kernel<128, false, 6, true,"], "W-step terms": ["fused_kernel<128, true, 0, true, 0,"], "fused:objective pass": ["fused_kernel<128, true, 1, false,"]},
    "c4sc": {"fused:objective pass": ["fused_kernel<512, true, 21, false,"]},   # (H-step / W-step tags: sums over several kernels per outer iteration, kept as composed in r6_13)
    "c2is256": {W: ["fused_kernel<256, true, 15, true,", "fused_kernel<256, true, 0, true, 0,"], H: ["fused_kernel<256, true, 15, true,", "fused_kernel<256, true, 0, true, 2,"], C: ["fused_kernel<256, true, 11, false,"]},
    "c4is": {W: ["fused_kernel<512, true, 0, true, 0, true, 8>"], C: ["fused_kernel<512, true, 11, false,"], N: [GP], G: [GP]},
    "c2is512": {W: ["fused_kernel<256, true, 0, true, 0,"], C: ["fused_kernel<256, true, 19, false,", "fused_kernel<256, true, 7, false,"], N: [GP], G: [GP]},
}
tag = sys.argv[1]
src = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out")
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
pm = json.load(open(path))
for wl, tags in MAP.items():
    secs = open(os.path.join(src, "%s_%s_pmc.md" % (tag, wl))).read().split("\n## ")[1:]
    def traffic(pat):
        hit = [s for s in secs if pat in s.split("\n")[0]]   # (the same instantiation may also run small products at other grids: the pass is the launch with the most traffic)
        assert hit, (wl, pat)
        return max(float(re.search(r"\*\*([0-9.e+]+) B\*\*", h).group(1)) for h in hit)
    for name, pats in tags.items():
        v = sum(traffic(p) for p in pats) / len(pats)
        old = pm.get(wl, {}).get(name)
        pm.setdefault(wl, {})[name] = int(round(v))
        print("%-8s %-78s %.4g  (was %s)" % (wl, name[:78], v, "%.4g" % old if old else "-"))
pm["_comment"] = ("HBM bytes per launch (c4sc: per OUTER ITERATION of a tag's launches) from separate rocprofv3 --pmc passes of the same bench.py command: 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 "
                  "(gfx950 FETCH_SIZE correction per MI355X_MICROARCH.md; scripts/pmc_passes.sh).  ALL workloads measured in ONE run on the round's final kernel sources: profiles/%s_<workload>_pmc.md "
                  "(scripts/r6_final_job.sh), written by profiles/pmc_update.py (tag -> kernel map there; a tag over two passes holds the mean over their launches).  c4sc's H-step / W-step tags: "
                  "the r6_13 composition, their kernels re-measured unchanged.  c3 against 4.446e9 algorithmic: H tiles re-read per row block." % tag)
json.dump(pm, open(path, "w"), indent=1)

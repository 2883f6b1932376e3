#!/usr/bin/env python3
"""Stamp profiles/pmc_traffic.json with the hash of the kernel sources it was measured on (bench.py withholds `roofline.traffic` when the
stamp does not match the tree).  Run after scripts/pmc_passes.sh and after the numbers in the file have been brought up to date:
    python profiles/pmc_stamp.py [--check]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
pm = json.load(open(path))
have = bench.kernel_sources_sha16()
if "--check" in sys.argv:
    print("stamp %s, tree %s: %s" % (pm.get("_kernel_sources_sha16"), have, "ok" if pm.get("_kernel_sources_sha16") == have else "STALE"))
    sys.exit(0 if pm.get("_kernel_sources_sha16") == have else 1)
pm["_kernel_sources_sha16"] = have
pm["_kernel_sources"] = list(bench.PMC_KERNEL_SOURCES)
json.dump(pm, open(path, "w"), indent=1)
print("stamped", have)

#!/usr/bin/env python3
"""Build-time sanity check of the fused kernels: every instantiation must hold exactly the MFMAs its fully unrolled tile body needs
(K for the first product, 32*K/32 per contraction of the second).  A partially unrolled body (clang's `#pragma unroll` budget) indexes
the accumulator arrays dynamically -- slow, and it produced wrong results once (K = 96 dual H-step kernel, round 2).
    python scripts/check_mfma_counts.py [fused_k32_96.hip ...]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nmf_toolbox_amd", "csrc")
files = sys.argv[1:] or sorted(f for f in os.listdir(CSRC) if f.startswith("fused_") and f.endswith(".hip"))
bad = 0
for f in files:
    with tempfile.TemporaryDirectory() as d:
        asm = os.path.join(d, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-pragma-unroll-threshold=1000000", "-I", os.path.join(ROOT, "include"),
                               "-I", CSRC, "--offload-device-only", "-S", "-c", os.path.join(CSRC, f), "-o", asm], stderr=subprocess.DEVNULL)
        lines = open(asm).read().split("\n")
    starts = [i for i, l in enumerate(lines) if l.startswith("_ZN4nmfx12fused_kernel") and l.split(";")[0].rstrip().endswith(":")]
    for s in starts:
        key = lines[s].split(":")[0]
        end = next(i for i in range(s, len(lines)) if lines[i].startswith(".Lfunc_end"))   # (a body may hold several s_endpgm: early exits)
        m = re.search(r"ILi(\d+)ELb([01])ELi(\d+)ELb([01])ELi(\d)", key)
        K, func, g2 = int(m.group(1)), int(m.group(3)), int(m.group(4))
        if func == 2 and not g2:
            continue                                  # instantiated but never launched (no cost, no second product)
        want = (K if func != 0 else 0) + (32 * (K // 32) * (2 if func in (4, 5) else 1) if g2 else 0)
        got = sum("v_mfma" in l for l in lines[s:end])
        if got != want:
            bad += 1
            print("MISMATCH %s %s: %d MFMAs, expected %d" % (f, key, got, want))
    print("%s: %d kernels checked" % (f, len(starts)))
sys.exit(1 if bad else 0)

"""Per-kernel register / scratch usage of the gfx950 code objects inside libnmfx's object files (hipcc embeds them as clang offload bundles):

    python scripts/kernel_resources.py [objdir] [--all]

prints every kernel that spills (private_segment_fixed_size or *_spill_count > 0) -- a spilling register-stationary kernel is a wrong schedule, see
fused_kernel.h -- or, with --all, one line per kernel.  tests/test_abi_and_host.py runs it over the built tree."""
import glob
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def code_objects(path):
    f = open(path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    i = f.find(magic)
    while i >= 0:
        n = struct.unpack_from("<Q", f, i + 24)[0]
        o = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", f, o)
            o += 24
            trip = f[o:o + tl].decode()
            o += tl
            if "gfx950" in trip and size > 0:
                yield f[i + off:i + off + size]
        i = f.find(magic, i + 1)


def kernels(path):
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as t:
            t.write(co)
            t.flush()
            notes = subprocess.run([READELF, "--notes", t.name], capture_output=True, text=True).stdout
        for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
            g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "0"])[1]
            yield dict(name=g("name"), agpr=int(blk.split()[0]), vgpr=int(g("vgpr_count")), sgpr=int(g("sgpr_count")), scratch=int(g("private_segment_fixed_size")),
                       vspill=int(g("vgpr_spill_count")), sspill=int(g("sgpr_spill_count")), lds=int(g("group_segment_fixed_size")))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    objdir = args[0] if args else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nmf_toolbox_amd", "csrc", "_obj")
    bad = total = 0
    for o in sorted(glob.glob(os.path.join(objdir, "*.o"))):
        for k in kernels(o):
            total += 1
            spills = k["scratch"] or k["vspill"] or k["sspill"]
            bad += bool(spills)
            if spills or "--all" in sys.argv:
                print("%-28s %s  vgpr %d agpr %d sgpr %d scratch %d B vspill %d sspill %d" % (os.path.basename(o), k["name"], k["vgpr"], k["agpr"], k["sgpr"], k["scratch"], k["vspill"], k["sspill"]))
    print("%d kernels, %d with scratch / spills" % (total, bad))
    return bad


if __name__ == "__main__":
    main()

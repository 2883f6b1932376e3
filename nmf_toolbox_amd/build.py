"""Build libnmfx.so (hand-written HIP for gfx950 + the C ABI of include/nmfx.h), in-tree.

    python -m nmf_toolbox_amd.build        # or nmf_toolbox_amd.build.build()

hipcc cross-compiles without a GPU.  The library is linked against the HIP runtime that PyTorch
ships (torch/lib/libamdhip64.so, no SONAME) when torch is importable, so that a process which also
uses torch.distributed holds ONE HIP runtime and device pointers / streams can be shared; the
rpath falls back to /opt/rocm/lib for consumers without torch (the MEX gateway).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INC = os.path.join(os.path.dirname(HERE), "include")
OUT = os.path.join(HERE, "libnmfx.so")
SOURCES = ["fused_cnmf_a.hip", "fused_cnmf_b.hip", "fused_cnmf_c.hip", "fused_cnmf_e.hip", "fused_cnmf_d.hip", "fused_cnmf_g.hip", "fused_cnmf_f.hip", "gemm_pipe_edge.hip", "gemm_pipe.hip", "gemm.hip", "fused_k224_256.hip", "fused_rag_k224_256.hip", "fused_k128_192.hip", "fused_rag_k128_192.hip", "fused_k32_96.hip", "fused_rag_k32_96.hip", "fused.hip", "aux.hip", "projfunc.hip", "small_mm.hip", "gemm64.hip", "engine.hip", "host_io.hip", "blocking.hip", "sc.hip", "sc64.hip", "multi_sc.hip", "rccl_backend.hip"]
ARCH = "gfx950"


def _torch_lib_dir():
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec and spec.origin:
            d = os.path.join(os.path.dirname(spec.origin), "lib")
            if os.path.exists(os.path.join(d, "libamdhip64.so")):
                return d
    except Exception:
        pass
    return None


def _deps(src, incdirs, seen=None):
    """the file and every header it reaches through #include "..." (so that touching api_common.h does not recompile the 20 kernel-only translation units)"""
    import re
    seen = set() if seen is None else seen
    if src in seen or not os.path.exists(src):
        return seen
    seen.add(src)
    with open(src) as f:
        for name in re.findall(r'^\s*#\s*include\s*"([^"]+)"', f.read(), re.M):
            for d in [os.path.dirname(src)] + incdirs:
                cand = os.path.join(d, name)
                if os.path.exists(cand):
                    _deps(cand, incdirs, seen)
                    break
    return seen


def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


# A/B builds of the kernel switches in csrc/nmfx_internal.h (measurement only: `python -m nmf_toolbox_amd.build --variant r5` -> libnmfx_r5.so, which
# NMFX_LIB_VARIANT=r5 makes _lib.py load instead of libnmfx.so; nothing in the package refers to a variant)
VARIANTS = {
    "r5": ["-DNMFX_KL_MODE=0", "-DNMFX_G2_VEC=0", "-DNMFX_G1_ASM=0"],     # the round-5 kernels: scalar KL map, closed-form sum(S), ds_read_b32 second product
    "kl1": ["-DNMFX_KL_MODE=1", "-DNMFX_G2_VEC=0", "-DNMFX_G1_ASM=0"],    # consistent KL cost on scalar VALU instructions (6 per element)
    "kl2": ["-DNMFX_KL_MODE=2", "-DNMFX_G2_VEC=0", "-DNMFX_G1_ASM=0"],    # ... on packed instructions (8 per pair)
    "kl2g2": ["-DNMFX_KL_MODE=2", "-DNMFX_G2_VEC=1", "-DNMFX_G1_ASM=0"],  # + vector LDS reads of the second product
}


def build(force: bool = False, verbose: bool = False, sanitize: bool = False, variant: str = "") -> str:
    """sanitize: the HOST pass of every translation unit with -fsanitize=address,undefined (device code objects unchanged: GPU ASan is not available on
    this pool) into libnmfx_asan.so + the campaign driver tests/host_asan/fuzz_multi -- test infrastructure, never loaded by the package"""
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    objdir = os.path.join(CSRC, "_obj_asan" if sanitize else ("_obj_" + variant if variant else "_obj"))
    out = os.path.join(HERE, "libnmfx_asan.so") if sanitize else (os.path.join(HERE, "libnmfx_%s.so" % variant) if variant else OUT)
    vdefs = VARIANTS[variant] if variant else []
    # (-g / -fno-omit-frame-pointer for the HOST pass only: handed to the device pass as well they change the gfx950 code objects -- frame pointer, CFI spills --
    # and the register-stationary kernels then return garbage: measured, profiles/archive/r4_01_host_asan.md)
    # -fno-sanitize=function: UBSan's indirect-call check and HIP's kernel handles do not mix -- `auto kern = some_kernel<...>; hipLaunchKernelGGL(kern, ...)`
    # (every templated launch of this library) is then silently NOT launched (hipcc 7.2; reproduced in 40 lines, profiles/archive/r4_01_host_asan.md)
    san = ["-fsanitize=address,undefined", "-fno-sanitize=function", "-fno-gpu-sanitize", "-fno-sanitize-recover=undefined", "-Xarch_host", "-g", "-Xarch_host", "-fno-omit-frame-pointer"] if sanitize else []
    os.makedirs(objdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if force or _newer(sorted(_deps(src, [CSRC, INC])) + [os.path.abspath(__file__)], obj):
            cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Werror=uninitialized", "-Wno-pass-failed"] + san + vdefs + [
                   "-I", INC, "-I", CSRC, "-c", src, "-o", obj]
            if os.path.basename(src).startswith("fused_"):
                # the biggest tile bodies (K = Kh*T = 512: 512 MFMAs; the dual-map kernels at K = 96 / 128) are past clang's default
                # budget for `#pragma unroll`, and a partially unrolled loop indexes the accumulator arrays dynamically (wrong
                # schedule at best; the K = 96 dual H-step kernel came out with 192 of its 288 MFMAs and wrong results)
                cmd[3:3] = ["-mllvm", "-pragma-unroll-threshold=1000000"]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(srcs), 8)) as ex:
        objs = list(ex.map(compile_one, srcs))
    if force or _newer(objs, out):
        tl = _torch_lib_dir()
        libdirs = ([tl] if tl else []) + ["/opt/rocm/lib"]
        if sanitize:   # clang links the sanitizer runtimes into the EXECUTABLE; the shared object keeps its references to them undefined
            cmd = ["/opt/rocm/lib/llvm/bin/clang++", "-shared", "-fsanitize=address,undefined", "-fno-sanitize=function", "-o", out] + objs
        else:
            cmd = ["g++", "-shared", "-o", out] + objs
        for d in libdirs:
            cmd += ["-L" + d]
        cmd += ["-lamdhip64", "-Wl,-rpath," + ":".join(libdirs)] + ([] if sanitize else ["-Wl,--no-undefined"]) + ["-lstdc++", "-lm"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    if sanitize:
        drv_src = os.path.join(os.path.dirname(HERE), "tests", "host_asan", "fuzz_multi.cpp")
        drv = os.path.join(os.path.dirname(drv_src), "fuzz_multi")
        if force or _newer([drv_src, out], drv):
            cmd = ["/opt/rocm/lib/llvm/bin/clang++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize=function", "-fno-omit-frame-pointer", "-I", INC, drv_src, "-o", drv,
                   "-L" + HERE, "-lnmfx_asan", "-Wl,-rpath," + HERE + ":" + ":".join(([_torch_lib_dir()] if _torch_lib_dir() else []) + ["/opt/rocm/lib"]), "-lpthread"]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            # the same driver against the NORMAL library (glibc's MALLOC_CHECK_ instead of ASan), and the microscope tests/host_asan/probe.cpp against both
            rp = "-Wl,-rpath," + HERE + ":" + ":".join(([_torch_lib_dir()] if _torch_lib_dir() else []) + ["/opt/rocm/lib"])
            d = os.path.dirname(drv_src)
            have_plain = os.path.exists(OUT)   # (the normal library is built by build() without --sanitize)
            for c in (["g++", "-std=c++17", "-O1", "-g", "-I", INC, drv_src, "-o", drv + "_plain", "-L" + HERE, "-lnmfx", rp, "-lpthread"],
                      ["g++", "-std=c++17", "-O1", "-I", INC, os.path.join(d, "probe.cpp"), "-o", os.path.join(d, "probe_plain"), "-L" + HERE, "-lnmfx", rp],
                      ["/opt/rocm/lib/llvm/bin/clang++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize=function", "-I", INC,
                       os.path.join(d, "probe.cpp"), "-o", os.path.join(d, "probe_asan"), "-L" + HERE, "-lnmfx_asan", rp]):
                if "-lnmfx" in c and not have_plain:
                    continue
                if verbose:
                    print(" ".join(c), flush=True)
                subprocess.check_call(c)
    return out


if __name__ == "__main__":
    var = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else ""
    print(build(force="--force" in sys.argv, verbose=True, sanitize="--sanitize" in sys.argv, variant=var))

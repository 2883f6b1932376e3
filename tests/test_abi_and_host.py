"""CPU suite: the C-ABI library loads and exports every symbol include/nmfx.h declares; the host-side mirror validates
arguments exactly like the reference's local ValidateParameters; without a GPU every compute call fails LOUDLY."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

from conftest import ROOT, synth


def _lib():
    from nmf_toolbox_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from nmf_toolbox_amd import build
        build.build()
    return _lib


def test_header_symbols_are_exported():
    L = _lib()
    hdr = open(os.path.join(ROOT, "include", "nmfx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(nmfx_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    lib = L.load()
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(set(L.EXPORTS)) == declared          # the Python binding list tracks the header
    assert lib.nmfx_version() == 600


def test_struct_layouts_match_header():
    """ctypes mirrors of nmfx_problem / nmfx_result / nmfx_engine_desc have the sizes the C compiler gives."""
    L = _lib()
    import subprocess
    import tempfile
    src = '#include "nmfx.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu\\n", sizeof(nmfx_problem), sizeof(nmfx_result), sizeof(nmfx_engine_desc));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")])
        sizes = [int(x) for x in subprocess.check_output([os.path.join(d, "s")]).split()]
    assert sizes == [C.sizeof(L.Problem), C.sizeof(L.Result), C.sizeof(L.EngineDesc)]


def test_no_silent_cpu_fallback():
    L = _lib()
    import nmf_toolbox_amd as A
    if A.device_count() > 0:
        pytest.skip("a GPU is present: the loud-failure path is only observable without one")
    V, W0, H0 = synth(16, 24, 3)
    for call in (lambda: A.nmf(V, 3, dict(W_init=W0, H_init=H0)), lambda: A.cnmf(V, 3, 2), lambda: A.nmfsc(V, 3),
                 lambda: A.ReconstructFromDecomposition(W0, H0), lambda: A.projfunc(np.ones(8), 2.0, 1.0, True)):
        with pytest.raises(A.NmfxError) as ei:
            call()
        assert ei.value.status == L.NMFX_ERR_NO_DEVICE and "no CPU fallback" in str(ei.value)
    from nmf_toolbox_amd.engine import Engine
    import torch
    t = torch.zeros(4, 4)
    with pytest.raises(A.NmfxError):
        Engine(t, t, t)


def test_host_validation_mirrors_reference_errors():
    import nmf_toolbox_amd as A
    V, W0, H0 = synth(16, 24, 4)
    with pytest.raises(ValueError, match="No update equations defined for cost function with divergence type bogus"):   # nmf.m:166
        A.nmf(V, 4, dict(divergence="bogus"))
    with pytest.raises(ValueError, match="Requested 2 sources. Given 1 initial encoding matrices."):                    # nmf.m:280
        A.nmf(V, [2, 2], dict(H_init=[H0]))
    with pytest.raises(ValueError, match="Requested 2 sources. Given 3 initial basis matrices."):                       # nmf.m:302
        A.nmf(V, [2, 2], dict(W_init=[W0, W0, W0]))
    with pytest.raises(ValueError, match="Requested 2 sources. Given 3 sparsity levels."):                              # nmf.m:318
        A.nmf(V, [2, 2], dict(W_sparsity=[0.1, 0.2, 0.3]))
    with pytest.raises(ValueError, match="Requested 2 sources. Given 3 update switches."):                              # nmf.m:368
        A.nmf(V, [2, 2], dict(W_fixed=[True, False, True]))
    with pytest.raises(ValueError, match="alpha = 0 and beta = 0 is not supported at this time."):                      # nmf.m:121
        A.nmf(V, 4, dict(divergence="ab", alpha=0, beta=0))
    with pytest.raises(ValueError, match="alpha = 0 and beta = 0"):                                                      # cnmf.m:134
        A.cnmf(V, 4, 2, dict(divergence="ab_divergence", alpha=0, beta=0))
    with pytest.raises(ValueError, match="Negative values in data!"):                                                    # nmfsc.m:58
        A.nmfsc(-V, 4)


def test_validate_defaults():
    from nmf_toolbox_amd import toolbox as T
    V, W0, H0 = synth(16, 24, 4)
    cfg, W, H, wc, hc = T._validate(V, [4], 1, dict(maxiter=-1, tolerance=0, W_sparsity=-3, alpha=5, seed=1), False)
    assert cfg["maxiter"] == 100 and cfg["tolerance"] == 1e-3 and cfg["W_sparsity"] == [0.0] and cfg["alpha"] == 1.0 and not wc and not hc
    assert np.allclose(np.sqrt((W[0] ** 2).sum(0)), 1.0) and H[0].shape == (4, 24) and H[0].min() >= 2.0 ** -52          # nmf.m:277,298-299
    cfg, W, H, wc, hc = T._validate(V, [1, 3], 3, dict(divergence="ab", alpha=0.5, H_sparsity=[0.1], W_fixed=True, seed=1), True)
    assert cfg["alpha"] == 0.5 and cfg["H_sparsity"] == [0.1, 0.1] and cfg["W_fixed"] == [True, True] and wc and hc
    assert W[1].shape == (16, 3, 3) and np.allclose(np.sqrt((W[1] ** 2).sum((0, 2))), 3.0)                               # cnmf.m:331-335


def test_shard_columns():
    from nmf_toolbox_amd.engine import shard_columns
    for n in (8, 13, 65536):
        for w in (1, 2, 3, 8):
            parts = [shard_columns(n, w, r) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n and all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in parts) - min(b - a for a, b in parts) <= 1


def test_bench_self_launch_starts_ranks_without_a_launcher():
    """`python bench.py --gpus 2` outside torch.distributed.run starts the two ranks itself (bench.py::self_launch).  Without a GPU each rank stops at
    the "needs an MI355X" check -- reaching it under torchrun's two-rank report is the host-side logic under test."""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-side check of the launcher; the GPU box runs tests/test_gpu_sharded.py::test_bench_self_launches_two_ranks")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "tiny"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode != 0
    # torchrun SIGTERMs the other rank as soon as one has failed, so the message is there once or twice; its report names a rank > 0
    assert "bench.py needs an MI355X" in r.stderr and "local_rank: 1" in r.stderr, r.stderr[-1500:]


def test_pmc_traffic_stamp_guards_the_bench_line():
    """profiles/pmc_traffic.json feeds `roofline.traffic` of the bench line and is measured in separate rocprofv3 --pmc passes: it is stamped with the hash of
    the sources that decide a launch's HBM traffic (the kernels AND the files that set grid / split geometry: engine.hip, sc.hip).  A matching stamp: the figures
    are handed out, named as not measured in this run.  A stale stamp: bench.py must WITHHOLD them (`traffic: null`) and say why -- never print old bytes next to new
    kernels.  (scripts/pmc_passes.sh + profiles/pmc_stamp.py bring the file up to date; `python profiles/pmc_stamp.py --check` tells which state the tree is in.)"""
    import json
    sys.path.insert(0, ROOT)
    import bench
    assert {"engine.hip", "sc.hip", "fused_kernel.h", "fused_launch.h", "gemm_pipe.h"} <= set(bench.PMC_KERNEL_SOURCES)
    pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    got, note = bench.pmc_traffic_for("c3")
    if pm.get("_kernel_sources_sha16") == bench.kernel_sources_sha16():
        assert got and all(v > 4.4e9 for v in got.values()) and "not measured in this run" in note      # c3: at least the algorithmic 4.45e9 B per launch
    else:
        assert got == {} and "STALE" in note and "withheld" in note


def test_abi_version_and_sizes_are_checked_by_the_loaders():
    """nmfx_problem grew (multi_backend) -- a client built against another header must be turned away, not read past its struct"""
    L = _lib()
    lib = L.load()
    sz = (C.c_int32 * 3)()
    lib.nmfx_abi_sizes(C.byref(sz, 0), C.byref(sz, 4), C.byref(sz, 8))
    assert tuple(sz) == (C.sizeof(L.Problem), C.sizeof(L.Result), C.sizeof(L.EngineDesc))
    hdr = open(os.path.join(ROOT, "include", "nmfx.h")).read()
    assert int(re.search(r"#define NMFX_VERSION (\d+)", hdr).group(1)) == L.ABI_VERSION == lib.nmfx_version()
    # a binding written against another version refuses the library (fresh interpreter: load() caches)
    import subprocess
    code = ("from nmf_toolbox_amd import _lib\n_lib.ABI_VERSION = 200\n"
            "try:\n    _lib.load()\nexcept ImportError as e:\n    print('refused:', e)\n")
    out = subprocess.check_output([sys.executable, "-c", code], cwd=ROOT, text=True)
    assert "refused:" in out and "ABI version 600" in out
    # the MEX gateway makes the same check before anything else
    mex = open(os.path.join(ROOT, "matlab", "nmfx_mex.c")).read()
    assert "nmfx_version() != NMFX_VERSION" in mex and "nmfx_abi_sizes" in mex


def test_multi_backend_is_validated():
    """values outside {0, 1, 2} are an error at the C ABI (before any device is touched), and unknown names are one in the Python wrapper"""
    L = _lib()
    lib = L.load()
    from nmf_toolbox_amd import toolbox
    assert [toolbox._multi_backend(v) for v in (None, "auto", "peer", "rccl", 0, 1, 2, 2.0, np.int32(1))] == [0, 0, 1, 2, 0, 1, 2, 2, 1]
    for bad in ("nccl", 3, -1, 1.5, True):
        with pytest.raises(ValueError, match="nmfx_multi_backend"):
            toolbox._multi_backend(bad)
    V, W0, H0 = (np.ascontiguousarray(a, dtype=np.float32) for a in synth(16, 24, 4))
    Wo, Ho, cost = np.zeros_like(W0), np.zeros_like(H0), np.zeros(3)
    for mb, ng in ((7, 0), (-1, 1), (3, 2)):
        p, r = L.Problem(), L.Result()
        p.m, p.n, p.K_total, p.T, p.dtype = 16, 24, 4, 1, L.F32
        p.V, p.W_init, p.H_init = V.ctypes.data, W0.ctypes.data, H0.ctypes.data
        p.num_sources, p.maxiter, p.tolerance = 1, 3, 1e-3
        p.multi_backend, p.n_gpus = mb, ng
        r.W, r.H, r.cost = Wo.ctypes.data, Ho.ctypes.data, cost.ctypes.data
        assert lib.nmfx_nmf(C.byref(p), C.byref(r)) == L.NMFX_ERR_INVALID
        assert b"multi_backend" in lib.nmfx_last_error()


def test_rccl_lookup_survives_a_library_that_cannot_be_loaded():
    """dlerror() was called twice and the second call's NULL went into a std::string -- SIGSEGV on any dlopen failure.  A bogus
    NMFX_RCCL_LIB must leave the process alive, and the candidates after it must still be tried"""
    _lib()
    import subprocess
    code = ("import ctypes as C\nfrom nmf_toolbox_amd import _lib\nl = _lib.load()\nv = C.c_int32(0)\n"
            "p = l.nmfx_rccl_library(C.byref(v))\nprint('path=%r version=%d' % (p, v.value))\n")
    env = dict(os.environ, NMFX_RCCL_LIB="/nonexistent/librccl.so")
    res = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, text=True, capture_output=True)
    assert res.returncode == 0, (res.returncode, res.stderr[-500:])
    assert "path=" in res.stdout and "/nonexistent" not in res.stdout


def test_no_instruction_reads_an_asm_mfma_result_too_early():
    """The first product's MFMAs are inline asm (fused_kernel.h, NMFX_G1_ASM): hipcc's hazard recogniser does not see them, and a register copy it places right behind one
    reads a result that arrives 18 wait states later -- that is what broke every K > 256 test on the hardware in round 6 with code that read right.  scripts/mfma_asm_lint.py
    walks the BUILT objects; here: (a) it finds the pattern in a disassembly that has it, and lets a clean chain pass, (b) the shipped fused kernels have none."""
    _lib()
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import mfma_asm_lint as L
    bad = """0000000000001000 <k>:
	v_mfma_f32_32x32x2_f32 v[18:33], v36, a126, v[18:33]
	v_mov_b64_e32 v[48:49], v[32:33]
	v_mfma_f32_32x32x2_f32 v[34:49], v55, a127, v[34:49]
	s_endpgm
"""
    good = """0000000000001000 <k>:
	v_mfma_f32_32x32x2_f32 v[18:33], v36, a126, v[18:33]
	v_mfma_f32_32x32x2_f32 v[18:33], v37, a127, v[18:33]
	v_mfma_f32_32x32x2_f32 v[2:17], v56, a0, 0
	s_nop 3
	v_rcp_f32_e32 v60, v18
	s_branch 12
	v_mov_b32_e32 v18, v60
	s_endpgm
"""
    fb, nb = L.lint_text(bad)
    fg, ng = L.lint_text(good)
    assert nb == 2 and len(fb) == 1 and "v_mov_b64" in fb[0][2] and ng == 3 and not fg, (fb, fg)
    early = good.replace("s_nop 3\n", "").replace("v_mfma_f32_32x32x2_f32 v[2:17], v56, a0, 0\n", "")   # the reader right behind the chain's last MFMA
    assert len(L.lint_text(early)[0]) >= 1
    objdir = os.path.join(ROOT, "nmf_toolbox_amd", "csrc", "_obj")
    if not os.path.isdir(objdir) or not any(f.startswith("fused") for f in os.listdir(objdir)):
        pytest.skip("no built objects in this tree (only the shared library travelled)")
    import glob
    from kernel_resources import code_objects
    findings, n = [], 0
    for f in sorted(glob.glob(os.path.join(objdir, "fused*.o"))):
        for co in code_objects(f):
            fnd, k = L.lint_code_object(co)
            findings += fnd
            n += k
    assert n > 50000 and not findings, (n, findings[:3])

#!/usr/bin/env python3
"""bench.py -- NMF multiplicative-update iterations/s on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c4|c5|tiny] [--overlap 0|2|4]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
(a plain `python bench.py --gpus N` with N > 1 starts that second command itself: self_launch)

A "step" is one full multiplicative-update iteration (W step, H step, V_hat refresh, cost) on synthetic V
that is already resident in HBM.  Default workload = the configuration the metric is quoted on
(BASELINE.json configs[2]: nmf.m KL divergence, V = 16384 x 65536, K = 256); it fits one GPU (V = 4 GiB fp32).
For N > 1, V and H are column-sharded over the ranks (total work fixed => "strong"), W is replicated, and
each iteration has ONE all-reduce (RCCL) of the packed W-step partials.

One JSON line on rank 0: metric/value/..., plus
  roofline     dominant kernel: algorithmic flops per launch / mean launch duration (hipEvents on the stream the
               kernel runs on, recorded inside the timed region) against the fp32 MFMA peak (157.3 TFLOP/s)
  cpu_baseline the float64 literal restatement of nmf.m (oracle/, NumPy + OpenBLAS, all host cores) on a bounded
               column sample of the same workload, scaled linearly in n  (kind "port": MATLAB is unavailable)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_HBM_GBS = 8000.0
EPS = 2.0 ** -52

WORKLOADS = {
    # name: (algorithm, divergence, m, n, K, T, F_alg multiplier of m*n*K*T)   -- BASELINE.md section 3
    "c3": ("nmf", "kl", 16384, 65536, 256, 1, 8.0),
    "c2": ("nmf", "euclidean", 8192, 32768, 128, 1, 12.0),
    "c4": ("cnmf", "euclidean", 4096, 16384, 64, 8, 12.0),
    "c4kl": ("cnmf", "kl", 4096, 16384, 64, 8, 16.0),          # config 4's shape with the KL divergence (fused S / numerator passes; dev aid)
    "c2is": ("nmf", "is", 8192, 32768, 128, 1, 12.0),         # config 2's shape with the Itakura-Saito divergence (dual-map fused kernels; dev aid)
    "c2is256": ("nmf", "is", 8192, 32768, 256, 1, 12.0),      # ... with K = 256: above 192 the two element maps run as two single-map passes per half-iteration (dev aid)
    "c4is": ("cnmf", "is", 4096, 16384, 64, 8, 12.0),          # config 4's shape with the Itakura-Saito divergence (dev aid)
    "c2is512": ("nmf", "is", 8192, 32768, 512, 1, 12.0),      # ... and nmf with a factor wider than the register-stationary kernels hold (dev aid)
    "c5": ("nmfsc", "euclidean", 8192, 32768, 128, 1, 12.0),   # H_sparsity 0.5; F_alg = (5 + tries)*2mnK = 12 mnK at one try per line search
    "c4sc": ("cnmfsc", "euclidean", 4096, 16384, 64, 8, 14.0),  # config 4's shape through cnmfsc.m, H_sparsity 0.5; (12 + 2*tries)*mnKT per outer iteration
    "tiny": ("nmf", "kl", 512, 1024, 16, 1, 8.0),
    "c3_shard8": ("nmf", "kl", 16384, 8192, 256, 1, 8.0),     # what ONE of 8 ranks holds at c3 (dev aid for the small-kernel overheads)
    "c3_shard4": ("nmf", "kl", 16384, 16384, 256, 1, 8.0),
    "c3_shard2": ("nmf", "kl", 16384, 32768, 256, 1, 8.0),
}


CPU_SAMPLE_COLS = {"c3": 4096, "c2": 8192, "c2is": 8192, "c2is256": 4096, "c2is512": 2048, "c4": 4096, "c4kl": 4096, "c4is": 4096, "tiny": 1024}


# what decides a launch's HBM traffic: the kernels, and the files that set grid / column-split geometry (the workgroup order over the XCDs decides L2 reuse)
PMC_KERNEL_SOURCES = ("fused_kernel.h", "fused_launch.h", "fused.hip", "gemm_pipe.h", "gemm_common.h", "engine.hip", "sc.hip", "nmfx_internal.h")


def kernel_sources_sha16():
    """sha256 over the sources of the kernels profiles/pmc_traffic.json describes (the fused passes, the pipelined GEMM): the PMC passes are separate
    rocprofv3 runs, so their per-launch HBM bytes are only worth printing while the kernels are the ones that were measured"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "nmf_toolbox_amd", "csrc")
    for f in PMC_KERNEL_SOURCES:
        with open(os.path.join(d, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic_for(workload):
    """(dict tag -> HBM bytes per launch, source note) from profiles/pmc_traffic.json -- or ({}, why not) when the file is missing or was measured on other kernel sources"""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        pm = json.load(open(path))
    except Exception:
        return {}, None
    want = pm.get("_kernel_sources_sha16")
    try:
        have = kernel_sources_sha16()
    except Exception:
        have = None
    if want is None or have is None or want != have:
        return {}, ("profiles/pmc_traffic.json is STALE (measured on kernel sources %s, this tree has %s): traffic withheld -- re-run scripts/pmc_passes.sh and "
                    "profiles/pmc_stamp.py" % (want, have))
    return pm.get(workload, {}), ("profiles/pmc_traffic.json: HBM bytes per launch from a separate rocprofv3 --pmc pass of the same command (FETCH_SIZE x2 gfx950 correction "
                                  "+ WRITE_SIZE) on these kernel sources (sha16 %s), not measured in this run" % have)


def cpu_baseline(alg, div, m, n, K, T, name="c3"):
    """Reference CPU path: oracle (float64 literal restatement, same GEMM list as nmf.m) on a bounded column sample."""
    from oracle import nmf_oracle as O
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [os.cpu_count() or 1])
    except Exception:
        threads = os.cpu_count() or 1
    rs = np.random.RandomState
    W0 = np.fmax(rs(1).rand(m, K) if alg == "nmf" else rs(1).rand(m, K, T), EPS)

    def per_iter_seconds(ns, iters):
        V = np.fmax(rs(1000).rand(m, ns), EPS)
        H0 = np.fmax(rs(2).rand(K, ns), EPS)
        cfg = dict(divergence=div, W_init=W0, H_init=H0, tolerance=1e-300)
        run = (lambda it: O.nmf(V, K, dict(cfg, maxiter=it))) if alg == "nmf" else (lambda it: O.cnmf(V, K, T, dict(cfg, maxiter=it)))
        t0 = time.perf_counter(); run(1); t1 = time.perf_counter() - t0
        t0 = time.perf_counter(); run(1 + iters); tn = time.perf_counter() - t0
        d = (tn - t1) / iters                       # removes init / first-touch cost
        return d if d > 0 else tn / (1 + iters)

    # The WHOLE workload when the host can hold the oracle's float64 m x n temporaries (~14 of them: SURVEY 8(d) allows a sample only "if host RAM is short") -- one
    # timed iteration then (the difference of a 2- and a 1-iteration run; c3: ~20 s each on the GPU box).  Otherwise a FIXED column sample and iteration count per
    # workload, so the baseline is comparable between runs and rounds.  Either way the figure moves by ~10 % from run to run on a shared host (0.046 .. 0.051 it/s
    # for c3 across rounds): a reported baseline, not a measurement to three digits.
    try:
        import psutil
        avail = psutil.virtual_memory().available
    except Exception:
        avail = 0
    full = avail > 14 * 8 * m * n + (8 << 30)
    ns = n if full else min(n, CPU_SAMPLE_COLS.get(name, 4096))
    iters = 1 if (full and (float(m) * n * K * T) > 2.0 ** 34) else 5
    per_iter = per_iter_seconds(ns, iters)
    what = ("the whole V=%dx%d" % (m, n)) if ns == n else ("V=%dx%d (first %d of %d columns: host RAM %.0f GB is short of the ~%.0f GB the full problem needs)" % (m, ns, ns, n, avail / 2.0 ** 30, 14 * 8 * m * n / 2.0 ** 30))
    return dict(value=(1.0 / per_iter) * ns / n, unit="iterations/s", cores=int(threads), kind="port",
                sample="float64 NumPy/OpenBLAS literal restatement of %s.m (%s) on %s, K=%d%s: %.4f s per iteration (%d timed)%s"
                       % (alg, div, what, K, (", T=%d" % T) if T > 1 else "", per_iter, iters, "" if ns == n else ", scaled by %d/%d (cost is linear in n)" % (ns, n)))


def bench_nmfsc(args, torch, dist, dev, world, rank, force_dist):
    """BASELINE config 5: nmfsc.m with Hoyer projection on H.  A step = one outer iteration (H line search + W update + cost,
    nmfsc.m:141-245) on device-resident data; warm-up iterations run first and the timed call RESUMES from their state
    (same W, H, step sizes), so the timed region is exactly `steps` steady-state outer iterations."""
    from nmf_toolbox_amd import _lib
    from nmf_toolbox_amd.engine import nmfsc_sharded, shard_columns
    import ctypes as C
    alg, div, m, n, K, T, fmul = WORKLOADS[args.workload]
    lo, hi = shard_columns(n, world, rank)
    nl = hi - lo
    g = torch.Generator(device=dev)
    g.manual_seed(1000 + rank)
    V = torch.rand((nl, m), generator=g, device=dev, dtype=torch.float32)
    g.manual_seed(1)
    W = torch.rand((K, m), generator=g, device=dev, dtype=torch.float32)
    g.manual_seed(2 + rank)
    H = torch.rand((nl, K), generator=g, device=dev, dtype=torch.float32)
    lib = _lib.load()

    def sync():
        torch.cuda.synchronize()
        if world > 1 or force_dist:
            dist.barrier()
            torch.cuda.synchronize()

    kw = dict(H_sparsity=args.h_sparsity, tolerance=-1.0, path=args.path)
    c0, info = nmfsc_sharded(V, W, H, maxiter=max(args.warmup, 1), **kw)
    sync()
    _lib.check(lib.nmfx_nmfsc_profile(0 if args.no_profile else 1))
    t0 = time.perf_counter()
    c1, info1 = nmfsc_sharded(V, W, H, maxiter=args.steps, resume=info, **kw)
    sync()
    dt = time.perf_counter() - t0
    nt = lib.nmfx_nmfsc_profile_ntags()
    ms, cnt = (C.c_double * nt)(), (C.c_int32 * nt)()
    _lib.check(lib.nmfx_nmfsc_profile_read(ms, cnt))
    names = [lib.nmfx_nmfsc_profile_tag_name(t).decode() for t in range(nt)]
    _lib.check(lib.nmfx_nmfsc_profile(0))
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1 or force_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    if rank == 0:
        its = args.steps / dt
        tries = info1["triesH"]
        f_mnk = 2.0 * m * nl * K
        work = {names[0]: (f_mnk, 4.0 * (m * nl + m * K + K * nl)),                       # objective pass: S = W*H only
                names[2]: (2.0 * f_mnk, 4.0 * (m * nl + 2 * m * K + 2 * K * nl)),         # residual pass: S = W*H, then W'*(S - V); reads V, W (+ transposed copy), H, writes dH
                names[3]: (f_mnk + 2.0 * K * K * (m + nl), 4.0 * (m * nl + 3 * m * K + K * nl))}
        phases = {names[t]: round(ms[t] / args.steps, 4) for t in range(nt) if cnt[t] > 0}
        tags = {names[t]: (ms[t], cnt[t]) for t in range(nt) if cnt[t] > 0 and names[t] in work}
        roof = None
        if tags:
            name = max(tags, key=lambda k: tags[k][0])
            avg_ms = tags[name][0] / tags[name][1]
            ach = work[name][0] / (avg_ms * 1e-3) / 1e12
            pm, tsrc = pmc_traffic_for(args.workload)   # HBM bytes per launch of the dominant kernel from the separate --pmc passes (not measured in this run)
            hit = [v for k, v in pm.items() if name.startswith(k)]
            traffic = hit[0] if hit else None
            roof = dict(bound="mfma", kernel=name, achieved=round(ach, 3), peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s", frac=round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                        traffic=traffic, traffic_source=tsrc, avg_launch_ms=round(avg_ms, 4), launches=int(tags[name][1]), flops_per_launch=work[name][0],
                        algorithmic_bytes_per_launch=work[name][1], phases_ms_per_step=phases)
        pj = None
        if cnt[1] > 0:   # projfunc: HBM-class -- one read of H' and of the step direction, one write, per call (nmfsc.m:154-157)
            pms = ms[1] / cnt[1]
            pbytes = 4.0 * 3 * nl * K
            pj = dict(calls=int(cnt[1]), ms_per_call=round(pms, 4), algorithmic_bytes_per_call=pbytes, achieved_GBps=round(pbytes / (pms * 1e-3) / 1e9, 1),
                      frac_of_hbm_peak=round(pbytes / (pms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                      note="latency-bound: ~3 dependent block reductions per inner iteration of projfunc.m:28-55, one 1024-thread workgroup per row of H")
        out = {
            "metric": "NMF multiplicative-update iterations/s", "value": round(its, 4), "unit": "iterations/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "nmfsc.m (Hoyer projection on H, H_sparsity=%g) outer iterations, V=%dx%d K=%d fp32 on %d GPU(s)" % (args.h_sparsity, m, n, K, world),
                       "name": args.workload, "m": m, "n": n, "K": K, "T": 1, "divergence": "euclidean", "H_sparsity": args.h_sparsity,
                       "line_search_tries_H": tries, "path": "fused kernels (objective = fused cost pass; dH = W'*(W*H - V) and the objective of the iterate from one fused residual pass; MU W step in Gram form)"},
            "effective_tflops": round((5.0 + float(np.mean(tries))) * f_mnk * world * its / 1e12, 3),
            "cost_first_last": [float(c1[0]), float(c1[-1])], "cost_monotone": bool(np.all(np.diff(c1) <= 0)),
            "roofline": roof, "projfunc": pj,
            "world_size_seen": int(dist.get_world_size()) if (world > 1 or force_dist) else 1,
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline_nmfsc(m, n, K, args.h_sparsity)
            except Exception as ex:
                out["cpu_baseline"] = {"value": None, "unit": "iterations/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (ex,)}
        print(json.dumps(out), flush=True)
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline_nmfsc(m, n, K, sH, ns=2048):
    """float64 restatement of nmfsc.m on the first `ns` columns, outer iterations 3-5 (after the long first line searches), scaled by ns/n"""
    from oracle import nmf_oracle as O
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [os.cpu_count() or 1])
    except Exception:
        threads = os.cpu_count() or 1
    rs = np.random.RandomState
    V, W0, H0 = rs(1000).rand(m, ns), rs(1).rand(m, K), rs(2).rand(K, ns)
    cfg = dict(W_init=W0, H_init=H0, H_sparsity=sH, tolerance=1e-300)
    t0 = time.perf_counter(); O.nmfsc(V, K, dict(cfg, maxiter=2)); t2 = time.perf_counter() - t0
    t0 = time.perf_counter(); O.nmfsc(V, K, dict(cfg, maxiter=5)); t5 = time.perf_counter() - t0
    per = max((t5 - t2) / 3.0, 1e-9)
    return dict(value=(1.0 / per) * ns / n, unit="iterations/s", cores=int(threads), kind="port",
                sample="float64 NumPy restatement of nmfsc.m (H_sparsity=%g), V=%dx%d (first %d of %d columns), K=%d: %.3f s per outer iteration (iterations 3-5) "
                       "on the sample, scaled by %d/%d" % (sH, m, ns, ns, n, K, per, ns, n))


def bench_cnmfsc(args):
    """cnmfsc.m (cnmfsc.m:155-277) at config 4's shape with the Hoyer projection on H.  The algorithm exists behind the blocking call only (its line
    searches keep the control flow in the library), so the timed region is cut out of ONE call of warmup + steps outer iterations by the library's own
    per-iteration completion times: every outer iteration ends with an objective the host reads (cnmfsc.m:269-270), i.e. with the device drained."""
    import ctypes as C
    import nmf_toolbox_amd as A
    from nmf_toolbox_amd import _lib
    alg, div, m, n, K, T, fmul = WORKLOADS[args.workload]
    rs = np.random.RandomState
    V = np.asfortranarray(rs(1000).rand(m, n))
    W0, H0 = rs(1).rand(m, K, T), rs(2).rand(K, n)
    lib = _lib.load()
    A.cnmfsc(V[:256, :512], 8, 2, dict(maxiter=1, H_sparsity=0.5))     # first-call costs
    total = args.warmup + args.steps
    _lib.check(lib.nmfx_nmfsc_profile(0 if args.no_profile else 1))
    info = {}
    W, H, c = A.cnmfsc(V, K, T, dict(W_init=W0, H_init=H0, H_sparsity=args.h_sparsity, maxiter=total, nmfx_disable_stop=True), info=info)
    nt = lib.nmfx_nmfsc_profile_ntags()
    ms, cnt = (C.c_double * nt)(), (C.c_int32 * nt)()
    _lib.check(lib.nmfx_nmfsc_profile_read(ms, cnt))
    names = [lib.nmfx_nmfsc_profile_tag_name(t).decode() for t in range(nt)]
    _lib.check(lib.nmfx_nmfsc_profile(0))
    ts = (C.c_double * total)()
    got = lib.nmfx_sc_iteration_seconds(ts, total)
    assert got == total, (got, total)
    dt = ts[total - 1] - (ts[args.warmup - 1] if args.warmup > 0 else 0.0)
    its = args.steps / dt
    tries = info["triesH"]
    f = 2.0 * m * n * K * T
    label = {names[0]: "objective passes: S = sum_t W_t*rshift_t(H) in registers -> 0.5||V - S||^2 (fused_kernel<K*T, ..., TT=T>, cost-only form; S stored as V_hat only at cnmfsc.m:269, for the next H step)",
             names[2]: "dH = sum_t W_t'*lshift_t(V_hat - V) as Q = W_flat'*(V_hat - V) (two-operand GEMM) + shift-sum",
             names[3]: "W-step terms (cnmfsc.m:257-263) without V_hat: N = V*H_stack' for all t in ONE fused pass over V; G = Hs*Hs' from the T lag Grams of H; the slice loop "
                       "pos_t = sum_s Wcur_s*G[(s,.),(t,.)], W_t = W0_t.*N_t./max(pos_t, eps) in one launch over the rows of W (aux.hip::cnmfsc_w_slices)"}
    # flops / algorithmic bytes of a tag PER OUTER ITERATION.  The W-step tag: the pass over V (2*m*n*K*T), the lag Grams (2*K*KT*n) and the slice loop (2*m*KT*KT)
    # objective tag: ONE whole-matrix pass per outer iteration (cnmfsc.m:269, V_hat stored for the next H step) + the initial one (cnmfsc.m:152); every line-search
    # try is a quadratic-expansion evaluation, one KT x n x KT product on the stacked shifts (2*KT*KT*n flops, K*n-sized operands) -- not a pass over V
    ntry = float(sum(tries))
    per_it = {names[0]: ((f * (total + 1) + 2.0 * (K * T) ** 2 * n * ntry) / total, (4.0 * (2 * m * n + m * K * T + K * n) * (total + 1) + 4.0 * (2 * K * T * n + (K * T) ** 2) * ntry) / total), names[2]: (f * cnt[2] / total, 4.0 * (2 * m * n + m * K * T + K * n) * cnt[2] / total),
              names[3]: (f + 2.0 * K * K * T * n + 2.0 * m * (K * T) ** 2, 4.0 * (m * n + 3 * m * K * T + 2 * K * n))}
    tags = {names[t]: (ms[t], cnt[t]) for t in range(nt) if cnt[t] > 0 and names[t] in per_it}
    roof = None
    if tags:
        name = max(tags, key=lambda k: tags[k][0])
        ms_it = tags[name][0] / total
        ach = per_it[name][0] / (ms_it * 1e-3) / 1e12
        pm, tsrc = pmc_traffic_for(args.workload)   # HBM bytes per outer iteration of that tag's launches, from the separate --pmc passes (not measured in this run)
        roof = dict(bound="mfma", kernel=label[name], achieved=round(ach, 3), peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s", frac=round(ach / PEAK_F32_MFMA_TFLOPS, 4), traffic=([v for k, v in pm.items() if name.startswith(k)] or [None])[0],
                    traffic_source=tsrc, ms_per_outer_iteration=round(ms_it, 4), launch_groups=int(tags[name][1]), flops_per_outer_iteration=per_it[name][0], algorithmic_bytes_per_outer_iteration=per_it[name][1],
                    tag_frac_of_peak={k: round(per_it[k][0] / (tags[k][0] / total * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4) for k in tags},
                    phases_ms_per_iteration_whole_call={names[t]: round(ms[t] / total, 4) for t in range(nt) if cnt[t] > 0})
    out = {"metric": "NMF multiplicative-update iterations/s", "value": round(its, 4), "unit": "iterations/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "cnmfsc.m (Hoyer projection on H, H_sparsity=%g) outer iterations, V=%dx%d K=%d T=%d fp32 on 1 GPU" % (args.h_sparsity, m, n, K, T),
                      "name": args.workload, "m": m, "n": n, "K": K, "T": T, "divergence": "euclidean", "H_sparsity": args.h_sparsity, "line_search_tries_H": tries,
                      "timed_region": "outer iterations %d..%d of one blocking call, by the library's per-iteration completion times" % (args.warmup + 1, total)},
           "effective_tflops": round((12.0 + 2.0 * float(np.mean(tries[args.warmup:]))) * m * n * K * T * its / 1e12, 3),   # the REFERENCE's flop count per outer iteration (cnmfsc.m), not what the kernels issue
           "cost_first_last": [float(c[0]), float(c[-1])], "roofline": roof}
    if not args.no_cpu_baseline:
        try:
            from oracle import nmf_oracle as O
            from threadpoolctl import threadpool_info
            threads = max([p_.get("num_threads", 1) for p_ in threadpool_info()] or [os.cpu_count() or 1])
            ns = 2048
            cfg = dict(W_init=W0, H_init=H0[:, :ns], H_sparsity=args.h_sparsity, tolerance=1e-300)
            t0 = time.perf_counter(); O.cnmfsc(V[:, :ns], K, T, dict(cfg, maxiter=2)); t2 = time.perf_counter() - t0
            t0 = time.perf_counter(); O.cnmfsc(V[:, :ns], K, T, dict(cfg, maxiter=4)); t4 = time.perf_counter() - t0
            per = max((t4 - t2) / 2.0, 1e-9)
            out["cpu_baseline"] = dict(value=(1.0 / per) * ns / n, unit="iterations/s", cores=int(threads), kind="port",
                                       sample="float64 NumPy restatement of cnmfsc.m (H_sparsity=%g), V=%dx%d (first %d of %d columns), K=%d, T=%d: %.3f s per outer iteration "
                                              "(iterations 3-4) on the sample, scaled by %d/%d" % (args.h_sparsity, m, ns, ns, n, K, T, per, ns, n))
        except Exception as ex:
            out["cpu_baseline"] = {"value": None, "unit": "iterations/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (ex,)}
    print(json.dumps(out), flush=True)


def bench_blocking(args):
    """The drop-in path end to end: what `[W,H,cost] = nmf(V, K, config)` costs a host that holds V as a column-major float64 array
    (MATLAB's native layout), for `--steps` iterations with the stop rule disabled.  Not the BASELINE metric (that is HBM-resident)."""
    import torch
    import nmf_toolbox_amd as A
    from nmf_toolbox_amd import _lib
    alg, div, m, n, K, T, fmul = WORKLOADS[args.workload]
    if alg not in ("nmf", "cnmf"):
        sys.exit("--api blocking: nmf / cnmf workloads")
    dt_ = np.float64 if args.host_dtype == "f64" else np.float32
    tg = torch.Generator().manual_seed(1000)          # torch's CPU generator fills 8 GiB on all cores in a few seconds
    V = torch.rand((n, m), generator=tg, dtype=torch.float64 if dt_ == np.float64 else torch.float32).clamp_(min=EPS).numpy().T   # m x n, column-major
    rs = np.random.RandomState
    W0 = np.fmax(rs(1).rand(m, K) if T == 1 else rs(1).rand(m, K, T), EPS)
    H0 = np.fmax(rs(2).rand(K, n), EPS)
    cfg = dict(divergence=div, W_init=W0, H_init=H0, maxiter=args.steps, nmfx_disable_stop=True)
    A.nmf(np.asfortranarray(V[:256, :512]), 16, dict(divergence=div, maxiter=1))     # first-call costs (context, code objects, the pinned buffers) are not ingest
    # --gpus N: the single-process sharded driver behind the same call (nmfx_problem.n_gpus), ONE LINE PER EXCHANGE BACKEND -- RCCL (ncclAllReduce on every
    # device's stream) and the peer reduce-scatter + all-gather -- so that the first multi-GPU lease is an A/B.  N = 1 runs both with one shard.
    import ctypes as C
    lib = _lib.load()
    backends = [None] if args.gpus <= 1 and not args.backends else [b for b in (args.backends or "rccl,peer").split(",") if b]
    for be in backends:
        # NMFX_BENCH_ONE_DEVICE: the N shards all on device 0 (peer backend only: RCCL refuses one device twice) -- what a 1-GPU box can say about the peer
        # reduce kernel: N-1 of N slices of `packed` pulled per shard at HBM instead of xGMI speed, an upper bound of its bandwidth
        devs = [0] * max(args.gpus, 1) if os.environ.get("NMFX_BENCH_ONE_DEVICE") else list(range(max(args.gpus, 1)))
        c2 = dict(cfg) if be is None else dict(cfg, nmfx_gpus=devs, nmfx_multi_backend=be)
        run = (lambda: A.nmf(V, K, c2)) if alg == "nmf" else (lambda: A.cnmf(V, K, T, c2))
        t0 = time.perf_counter()
        try:
            W, H, c = run()
        except _lib.NmfxError as ex:
            print(json.dumps({"metric": "blocking host-buffer call, end to end (NOT the BASELINE metric)", "workload": args.workload, "n_gpus": args.gpus, "backend": be, "error": str(ex)}), flush=True)
            continue
        wall = time.perf_counter() - t0
        tm = _lib.last_call_timing()
        out = {"metric": "blocking host-buffer call, end to end (NOT the BASELINE metric)", "workload": args.workload, "m": m, "n": n, "K": K, "T": T, "divergence": div,
               "host_dtype": args.host_dtype, "iterations": args.steps, "call_wall_s": round(wall, 4), "ingest_s": round(tm["ingest_s"], 4),
               "iterate_s": round(tm["iterate_s"], 4), "egress_s": round(tm["egress_s"], 4), "python_wrapper_s": round(wall - tm["ingest_s"] - tm["iterate_s"] - tm["egress_s"], 4),
               "host_bytes_in": tm["host_bytes_in"], "GBps_h2d": round(tm["host_bytes_in"] / max(tm["ingest_s"], 1e-9) / 1e9, 2),
               "host_bytes_out": tm["host_bytes_out"], "GBps_d2h": round(tm["host_bytes_out"] / max(tm["egress_s"], 1e-9) / 1e9, 2),
               "ms_per_iteration_inside": round(1e3 * tm["iterate_s"] / args.steps, 4), "host_cores": os.cpu_count(),
               "cost_first_last": [float(c[0]), float(c[-1])]}
        if be is not None:
            ms, cnt, used, ver = C.c_double(0), C.c_int32(0), C.c_int32(0), C.c_int32(0)
            _lib.check(lib.nmfx_last_call_exchange(C.byref(ms), C.byref(cnt), C.byref(used)))
            mK = m * K * T
            payload = 4 * (mK + (K * T if div == "kl" else (K * T) ** 2))
            p_ = max(args.gpus, 1)
            out.update(n_gpus=p_, backend=be, backend_used={1: "peer", 2: "rccl"}.get(used.value, "?"), allreduce_ms_per_step=round(ms.value, 4), allreduces_timed=cnt.value,
                       allreduce_bytes=payload, allreduce_busbw_GBps=round(2.0 * (p_ - 1) / p_ * payload / max(ms.value * 1e-3, 1e-12) / 1e9, 2) if p_ > 1 else None,
                       rccl_library=(lib.nmfx_rccl_library(C.byref(ver)) or b"").decode(), rccl_version=ver.value,
                       note="one process, one host thread, one stream + engine per device (what a MEX caller of nmf() gets); ingest / iterate include the per-device uploads")
        print(json.dumps(out), flush=True)


def self_launch(n):
    """`python bench.py --gpus N` outside any launcher: re-run this very command line under torch.distributed.run with N ranks on
    127.0.0.1 and a free port, pass rank 0's JSON line through (the children inherit stdout), and return their exit status."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")                # torchrun would set 1 (and print a banner about it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="dev aid: no hipEvent pairs around the launch groups (the line then has no roofline object)")
    ap.add_argument("--path", type=int, default=0, help="0 auto, 1 generic (materialised V_hat), 2 fused")
    ap.add_argument("--overlap", type=int, default=int(os.environ.get("NMFX_W_CHUNKS", "0") or 0), choices=[0, 1, 2, 4, 8],
                    help="N > 1: row chunks of the W-step partial whose all-reduces overlap the next chunk's compute (0/1 = one blocking all-reduce)")
    ap.add_argument("--api", default="engine", choices=["engine", "blocking"],
                    help="engine: device-resident phase API (the metric).  blocking: ONE call of the host-buffer entry point a MATLAB user makes "
                         "(nmf.m:1 / cnmf.m:1) on float64 host arrays -- prints ingest_s / iterate_s / egress_s / GBps_h2d, not the metric")
    ap.add_argument("--host-dtype", default="f64", choices=["f64", "f32"], help="--api blocking: precision of the host arrays")
    ap.add_argument("--backends", default="", help="--api blocking: comma list of exchange backends to run, one JSON line each (rccl, peer); default with --gpus N > 1: both")
    ap.add_argument("--numpy-inputs", action="store_true",
                    help="V, W_init, H_init from numpy.random.RandomState seeds 1000 / 1 / 2 (tests/conftest.py::synth: SURVEY 8(d)'s inputs, the ones every parity test and "
                         "tests/golden/fullsize_*.npz use) generated on the host and uploaded once, instead of torch.rand on the device; the line then carries "
                         "`oracle_check`: the run's cost vector against the float64 oracle's fixture for the workload")
    ap.add_argument("--spinup-ms", type=float, default=300.0,
                    help="device spin-up before the W warm-up steps: the W-step partial of the engine (it changes neither W nor H) is launched untimed until this much "
                         "wall time has passed, so that the warm-up and the timed region run at the sustained clock, not on the ramp from idle (rocm-smi: 1.36 -> 2.39 GHz over "
                         "the first ~0.1 s; a 20-step run of a 1.2 ms iteration read 7 %% low without it).  0 = off")
    ap.add_argument("--h-sparsity", type=float, default=0.5, help="c5: Hoyer sparseness target of the rows of H (nmfsc.m:102-110)")
    args = ap.parse_args()

    if args.api == "blocking":
        return bench_blocking(args)
    import torch
    import torch.distributed as dist
    from nmf_toolbox_amd import _lib
    from nmf_toolbox_amd.engine import Engine, shard_columns

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "RANK" not in os.environ and "LOCAL_RANK" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves, exactly as the driver's torchrun command would
        return self_launch(args.gpus)
    if args.gpus != world:
        # a launcher's WORLD_SIZE is what actually runs; say so instead of dying on the first contact with a multi-GPU node
        print("bench.py: --gpus %d but WORLD_SIZE=%d: using the launcher's world size" % (args.gpus, world), file=sys.stderr, flush=True)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # dev aid for 1-GPU boxes: NMFX_BENCH_BACKEND=gloo NMFX_BENCH_ONE_DEVICE=1 runs all ranks on cuda:0 over gloo (RCCL refuses
    # two ranks on one device); the driver's multi-GPU runs use the defaults (one GPU per rank, nccl = RCCL)
    backend = os.environ.get("NMFX_BENCH_BACKEND", "nccl")
    if os.environ.get("NMFX_BENCH_ONE_DEVICE", "0") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_dist = os.environ.get("NMFX_BENCH_FORCE_DIST", "0") == "1"   # dev aid: run the N > 1 code path (RCCL all-reduce included) with one rank
    if force_dist and world == 1:
        os.environ.setdefault("MASTER_PORT", "29577")
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    alg, div, m, n, K, T, fmul = WORKLOADS[args.workload]
    if alg == "nmfsc":
        return bench_nmfsc(args, torch, dist, dev, world, rank, force_dist)
    if alg == "cnmfsc":
        if world > 1:
            sys.exit("c4sc: cnmfsc is single-GPU")
        return bench_cnmfsc(args)
    lo, hi = shard_columns(n, world, rank)
    nl = hi - lo
    # synthetic inputs generated in HBM: V = max(U(0,1), eps) per shard (seed 1000+rank), W seed 1, H seed 2+rank
    g = torch.Generator(device=dev)
    if args.numpy_inputs:
        # SURVEY 8(d) / tests/conftest.py::synth: the SAME inputs as the oracle fixtures (every rank forms the global arrays and keeps its columns)
        rs = np.random.RandomState
        Vn = np.fmax(rs(1000).rand(m, n), 2.0 ** -52)[:, lo:hi]
        V = torch.from_numpy(np.ascontiguousarray(Vn.T, dtype=np.float32)).to(dev)
        del Vn
        Wn = np.fmax(rs(1).rand(m, K) if T == 1 else rs(1).rand(m, K, T), 2.0 ** -52)
        W = torch.from_numpy(np.ascontiguousarray(Wn.reshape(m, K * T, order="F").T, dtype=np.float32)).to(dev)   # slice t in rows t*K .. t*K+K-1 of the (T*K, m) array
        H = torch.from_numpy(np.ascontiguousarray(np.fmax(rs(2).rand(K, n), 2.0 ** -52)[:, lo:hi].T, dtype=np.float32)).to(dev)
    else:
        g.manual_seed(1000 + rank)
        V = torch.rand((nl, m), generator=g, device=dev, dtype=torch.float32).clamp_(min=EPS)
        g.manual_seed(1)
        W = torch.rand((T * K, m), generator=g, device=dev, dtype=torch.float32).clamp_(min=EPS)      # identical on every rank
        g.manual_seed(2 + rank)
        H = torch.rand((nl, K), generator=g, device=dev, dtype=torch.float32).clamp_(min=EPS)
    halo = (0, 0)
    if alg == "cnmf" and world > 1 and T > 1:
        # column-sharded cnmf: H gets T-1 halo columns on each inner side, V on the right; the initial halo contents are the
        # neighbours' edge columns, moved once point-to-point (Engine.exchange_halos keeps H's halos fresh afterwards)
        h = T - 1
        hL, hR = (h if rank > 0 else 0), (h if rank < world - 1 else 0)
        Hx = torch.zeros((hL + nl + hR, K), device=dev, dtype=torch.float32)
        Hx[hL:hL + nl] = H
        Vx = torch.zeros((nl + hR, m), device=dev, dtype=torch.float32)
        Vx[:nl] = V
        torch.cuda.synchronize()
        ops = []
        if rank > 0:
            ops += [dist.P2POp(dist.isend, V[:h].contiguous(), rank - 1), dist.P2POp(dist.isend, H[:h].contiguous(), rank - 1),
                    dist.P2POp(dist.irecv, Hx[:hL], rank - 1)]
        if rank < world - 1:
            ops += [dist.P2POp(dist.irecv, Vx[nl:], rank + 1), dist.P2POp(dist.irecv, Hx[hL + nl:], rank + 1),
                    dist.P2POp(dist.isend, H[nl - h:].contiguous(), rank + 1)]
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        torch.cuda.synchronize()
        V, H, halo = Vx, Hx, (hL, hR)
    eng = Engine(V, W, H, divergence=div, T=T, algorithm=alg, path=args.path, halo=halo, use_dist=True if force_dist else None,
                 n_chunks=max(args.overlap, 1))
    eng.init()
    costs = torch.zeros(args.steps + args.warmup + 1, dtype=torch.float64, device=dev)

    def sync():
        torch.cuda.synchronize()
        if world > 1 or force_dist:
            dist.barrier()
            torch.cuda.synchronize()

    spun_ms, spun_calls = 0.0, 0
    if args.spinup_ms > 0:
        t_sp = time.perf_counter()
        while (time.perf_counter() - t_sp) * 1e3 < args.spinup_ms:
            for _ in range(4):
                eng.wstep_partial()          # reads V, W, H; writes only `packed` (and the engine's lagged-cost scratch): the factors do not move
            torch.cuda.synchronize()
            spun_calls += 4
        spun_ms = (time.perf_counter() - t_sp) * 1e3
    eng.iterate(args.warmup, costs)
    sync()
    eng.profile(0 if args.no_profile else 2)   # event pairs around the MFMA launch groups only: bracketing the small kernels too costs 0.4 % at N = 1, 4 % on an 8-GPU shard
    t0 = time.perf_counter()
    eng.iterate(args.steps, costs[args.warmup:])
    sync()
    dt = time.perf_counter() - t0
    prof = eng.profile_read()
    comm_total_ms, comm_calls = eng.comm_ms()
    eng.profile(False)
    path_kind, n_chunks_used, packed_bytes = eng.path_kind, eng.n_chunks, int(eng.packed.numel() * 4)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    rank_info = None
    if world > 1 or force_dist:
        # first contact with a multi-GPU node: every rank's own time and kernel path next to the MAX the metric is computed from
        mine = torch.tensor([dt, float(path_kind), float(eng.cost_lag)], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(allr, mine)
        rank_info = [[float(x) for x in t.tolist()] for t in allr]
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    c = costs[: args.warmup + args.steps].cpu().numpy()

    # N > 1: rank 0 also times the WHOLE problem on its own GPU (no collective) right after the timed region, so the line carries
    # its own strong-scaling reference; the driver computes efficiency itself from the per-N lines, this is a cross-check
    single_its = None
    if world > 1 and rank == 0 and alg == "nmf" and (m * n * 4 + 3 * m * K * 4 * 4 + 3 * n * K * 4) < 40e9:
        try:
            del eng
            torch.cuda.empty_cache()
            g.manual_seed(1000)
            Vf = torch.rand((n, m), generator=g, device=dev, dtype=torch.float32).clamp_(min=EPS)
            g.manual_seed(1)
            Wf = torch.rand((K, m), generator=g, device=dev, dtype=torch.float32).clamp_(min=EPS)
            g.manual_seed(2)
            Hf = torch.rand((n, K), generator=g, device=dev, dtype=torch.float32).clamp_(min=EPS)
            e1 = Engine(Vf, Wf, Hf, divergence=div, path=args.path, use_dist=False)
            e1.init()
            cs = torch.zeros(8, dtype=torch.float64, device=dev)
            e1.iterate(1, cs)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            e1.iterate(5, cs)
            torch.cuda.synchronize()
            single_its = 5.0 / (time.perf_counter() - t1)
            e1.close()
            del e1, Vf, Wf, Hf
            eng = None
        except Exception:
            single_its = None
    # RCCL prints its version banner through C stdio, which is flushed only at exit: push every rank's buffered output out
    # now so that rank 0's JSON line is the last thing on stdout
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    if world > 1 or force_dist:
        dist.barrier()
    if rank == 0:
        its = args.steps / dt
        f_alg = fmul * m * n * K * T
        # dominant kernel = the tag with the largest total time inside the timed region
        tags = {k: v for k, v in prof.items() if v["launches"] > 0 and v["flops"] > 0}
        roof = None
        if tags:
            name = max(tags, key=lambda k: tags[k]["ms_total"])
            d = tags[name]
            avg_ms = d["ms_total"] / d["launches"]
            ach = d["flops"] / (avg_ms * 1e-3) / 1e12
            pm, tsrc = pmc_traffic_for(args.workload) if world == 1 else ({}, None)   # the PMC passes were taken on the whole problem: they say nothing about a shard's launch
            traffic = pm.get(name)
            roof = dict(bound="mfma", kernel=name, achieved=round(ach, 3), peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s", frac=round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                        traffic=traffic, traffic_source=tsrc,
                        avg_launch_ms=round(avg_ms, 4), launches=d["launches"], flops_per_launch=d["flops"],
                        algorithmic_bytes_per_launch=d["bytes"],
                        phases_ms_per_step={k: round(v["ms_total"] / args.steps, 4) for k, v in prof.items() if v["launches"] > 0})
            # everything that is not bracketed: the small kernels (reductions, updates, transposes), launch gaps and, for N > 1, the exchange
            roof["phases_ms_per_step"]["small kernels + gaps (remainder)"] = round(1e3 * dt / args.steps - sum(roof["phases_ms_per_step"].values()), 4)
        out = {
            "metric": "NMF multiplicative-update iterations/s", "value": round(its, 4), "unit": "iterations/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s.m %s MU, V=%dx%d K=%d%s fp32, V column-sharded over %d GPU(s)" % (alg, div, m, n, K, (" T=%d" % T) if T > 1 else "", world),
                       "name": args.workload, "m": m, "n": n, "K": K, "T": T, "divergence": div, "cost_every_iteration": True,
                       "path": {1: "fused kernels (V_hat never materialised)", 2: "Gram form on the generic GEMM (V_hat never materialised)",
                                3: "fused cnmf passes, shift-sum in LDS + Gram denominators (V_hat never materialised)",
                                4: "fused cnmf passes, shift-sum in LDS; the element maps' values (KL: R = V./V_hat; IS / alpha-beta: both maps) in HBM, V_hat never",
                                5: "K > 256 (KL, IS, alpha-beta): S = W*H over column blocks on the stationary kernel, the element maps' values in HBM, V_hat never",
                                6: "euclidean with K > 256: numerators block by block on the stationary kernel, Gram-form cost, V_hat never",
                                0: "generic GEMM (materialised V_hat)"}[path_kind]},
            "effective_tflops": round(f_alg * its / 1e12, 3),
            "cost_first_last": [float(c[0]), float(c[-1])], "cost_monotone": bool(np.all(np.diff(c) <= 1e-7 * abs(c[0]))),
            "roofline": roof,
        }
        out["data"] = "synthetic (numpy RandomState 1000 / 1 / 2, SURVEY 8(d))" if args.numpy_inputs else "synthetic"
        if args.numpy_inputs:
            # the run's costs against the float64 oracle's at this geometry (tests/golden/fullsize_*.npz, made by tests/golden/make_fullsize_golden.py; read only --
            # nothing of oracle/ is executed here).  cost(i) is the cost after iteration i+1 of one uninterrupted run: warm-up and timed steps are one sequence
            fxname = {"c3": "c3_full", "c2": "c2_full", "c4": "c4_full_euclidean", "c4kl": "c4_full_kl", "c3_shard8": "c3_shard"}.get(args.workload)
            fxpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "fullsize_%s.npz" % fxname) if fxname else None
            if fxpath and os.path.exists(fxpath):
                fc = np.load(fxpath)["cost"]
                k = int(min(len(fc), len(c)))
                out["oracle_check"] = {"fixture": "tests/golden/fullsize_%s.npz" % fxname, "entries_compared": k,
                                       "max_rel_err": float(np.max(np.abs(c[:k] - fc[:k]) / np.abs(fc[:k]))) if k else None,
                                       "cost_bench": [float(x) for x in c[:k]], "cost_oracle_f64": [float(x) for x in fc[:k]], "contract": 1e-6}
            else:
                out["oracle_check"] = {"fixture": None, "note": "no full-size oracle fixture for this workload"}
        out["spinup"] = {"ms": round(spun_ms, 1), "wstep_partial_launches": spun_calls, "note": "untimed, before the warm-up steps; W and H untouched"}
        out["world_size_seen"] = int(dist.get_world_size()) if (world > 1 or force_dist) else 1
        if rank_info:
            ms = [1e3 * r[0] / args.steps for r in rank_info]
            out["per_rank_ms_per_step"] = {"min": round(min(ms), 4), "max": round(max(ms), 4), "all": [round(x, 4) for x in ms]}
            out["path_kind_per_rank"] = [int(r[1]) for r in rank_info]
            out["all_ranks_same_path"] = len({(int(r[1]), int(r[2])) for r in rank_info}) == 1
        out["overlap_chunks"] = int(n_chunks_used)
        if comm_calls:   # N > 1: how long the compute stream waited for the packed all-reduce (rank 0), per step
            ar_ms = comm_total_ms / args.steps
            S = packed_bytes
            out["allreduce_ms_per_step"] = round(ar_ms, 4)
            out["allreduce_bytes"] = S
            p_ = max(world, 1)
            if ar_ms > 0 and p_ > 1:   # bus bandwidth 2(p-1)/p * S / t against one xGMI link (ring bound) and all seven
                busbw = 2.0 * (p_ - 1) / p_ * S / (ar_ms * 1e-3) / 1e9
                out["allreduce_busbw_GBps"] = round(busbw, 2)
                out["allreduce_busbw_vs_link_153GBps"] = round(busbw / 153.0, 3)
                out["allreduce_busbw_vs_7links_1071GBps"] = round(busbw / 1071.0, 3)
        if single_its is not None:
            out["single_gpu_its_same_run"] = round(single_its, 4)
            out["strong_scaling_eff"] = round(its / (world * single_its), 4)
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(alg, div, m, n, K, T, args.workload)
            except Exception as ex:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "unit": "iterations/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (ex,)}
        print(json.dumps(out), flush=True)
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

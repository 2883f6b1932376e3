#!/usr/bin/env python3
"""bench.py -- NMF multiplicative-update iterations/s on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c4|tiny]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

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
    "tiny": ("nmf", "kl", 512, 1024, 16, 1, 8.0),
    "c3_shard8": ("nmf", "kl", 16384, 8192, 256, 1, 8.0),     # what ONE of 8 ranks holds at c3 (dev aid for the small-kernel overheads)
    "c3_shard4": ("nmf", "kl", 16384, 16384, 256, 1, 8.0),
    "c3_shard2": ("nmf", "kl", 16384, 32768, 256, 1, 8.0),
}


def cpu_baseline(alg, div, m, n, K, T, budget_s=40.0):
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

    pilot_n = min(n, 256)
    pilot = per_iter_seconds(pilot_n, 2)
    ns = pilot_n
    while ns * 2 <= n and pilot * (ns * 2 / pilot_n) * 4 <= budget_s:   # 1 + (1+2) iterations must fit the budget
        ns *= 2
    per_iter = per_iter_seconds(ns, 2) if ns > pilot_n else pilot
    k = min(10, int(12.0 / max(per_iter, 1e-3)))                        # spend ~10-20 s of CPU work on the final measurement
    if k > 2:
        per_iter = per_iter_seconds(ns, k)
    return dict(value=(1.0 / per_iter) * ns / n, unit="iterations/s", cores=int(threads), kind="port",
                sample="float64 NumPy/OpenBLAS literal restatement of %s.m (%s), V=%dx%d (first %d of %d columns), K=%d%s: %.4f s/iter on the sample, "
                       "scaled by %d/%d (cost is linear in n)" % (alg, div, m, ns, ns, n, K, (", T=%d" % T) if T > 1 else "", per_iter, ns, n))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--path", type=int, default=0, help="0 auto, 1 generic (materialised V_hat), 2 fused")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from nmf_toolbox_amd import _lib
    from nmf_toolbox_amd.engine import Engine, shard_columns

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
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
    lo, hi = shard_columns(n, world, rank)
    nl = hi - lo
    # synthetic inputs generated in HBM: V = max(U(0,1), eps) per shard (seed 1000+rank), W seed 1, H seed 2+rank
    g = torch.Generator(device=dev)
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
    eng = Engine(V, W, H, divergence=div, T=T, algorithm=alg, path=args.path, halo=halo, use_dist=True if force_dist else None)
    eng.init()
    costs = torch.zeros(args.steps + args.warmup + 1, dtype=torch.float64, device=dev)

    def sync():
        torch.cuda.synchronize()
        if world > 1 or force_dist:
            dist.barrier()
            torch.cuda.synchronize()

    eng.iterate(args.warmup, costs)
    sync()
    eng.profile(True)
    t0 = time.perf_counter()
    eng.iterate(args.steps, costs[args.warmup:])
    sync()
    dt = time.perf_counter() - t0
    prof = eng.profile_read()
    comm_total_ms, comm_calls = eng.comm_ms()
    eng.profile(False)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1 or force_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    c = costs[: args.warmup + args.steps].cpu().numpy()

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
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(pmc):
                try:
                    traffic = json.load(open(pmc)).get(args.workload, {}).get(name)
                except Exception:
                    traffic = None
            roof = dict(bound="mfma", kernel=name, achieved=round(ach, 3), peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s", frac=round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                        traffic=traffic, avg_launch_ms=round(avg_ms, 4), launches=d["launches"], flops_per_launch=d["flops"],
                        algorithmic_bytes_per_launch=d["bytes"],
                        phases_ms_per_step={k: round(v["ms_total"] / args.steps, 4) for k, v in prof.items() if v["launches"] > 0})
        out = {
            "metric": "NMF multiplicative-update iterations/s", "value": round(its, 4), "unit": "iterations/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s.m %s MU, V=%dx%d K=%d%s fp32, V column-sharded over %d GPU(s)" % (alg, div, m, n, K, (" T=%d" % T) if T > 1 else "", world),
                       "name": args.workload, "m": m, "n": n, "K": K, "T": T, "divergence": div, "cost_every_iteration": True,
                       "path": {1: "fused kernels (V_hat never materialised)", 2: "Gram form on the generic GEMM (V_hat never materialised)",
                                0: "generic GEMM (materialised V_hat)"}[eng.path_kind]},
            "effective_tflops": round(f_alg * its / 1e12, 3),
            "cost_first_last": [float(c[0]), float(c[-1])], "cost_monotone": bool(np.all(np.diff(c) <= 1e-7 * abs(c[0]))),
            "roofline": roof,
        }
        if comm_calls:   # N > 1: how long the compute stream waited for the packed all-reduce (rank 0), per step
            out["allreduce_ms_per_step"] = round(comm_total_ms / args.steps, 4)
            out["allreduce_bytes"] = int(eng.packed.numel() * 4)
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(alg, div, m, n, K, T)
            except Exception as ex:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "unit": "iterations/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (ex,)}
        print(json.dumps(out), flush=True)
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

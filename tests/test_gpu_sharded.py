"""-m gpu: the N > 1 path on real kernels.  A 1-GPU box cannot host two RCCL ranks, so (a) two engines on column halves
are driven in ONE process with the all-reduce replaced by an explicit sum of their packed buffers (linearity of the
W-step sums in the column index -- the size-independent property the sharding rests on), at a small size against the
oracle and at a BASELINE-sized shard against the unsharded HIP run; (b) the real torch.distributed loop is run with two
processes sharing cuda:0 over gloo."""
import os
import socket

import numpy as np
import pytest

from conftest import EPS, rel_fro, synth

pytestmark = pytest.mark.gpu


def _engines(torch, V, W0, H0, div, parts, path):
    from nmf_toolbox_amd.engine import Engine, colmajor_to_torch
    dev = "cuda:0"
    engs = []
    for r, (lo, hi) in enumerate(parts):
        e = Engine(colmajor_to_torch(V[:, lo:hi], dev), colmajor_to_torch(W0, dev), colmajor_to_torch(H0[:, lo:hi], dev), divergence=div, path=path, use_dist=False)
        from nmf_toolbox_amd import _lib
        _lib.check(e.lib.nmfx_engine_set_rank0(e.h, 1 if r == 0 else 0))
        e.init()
        engs.append(e)
    return engs


def _run_emulated(torch, engs, iters):
    """run_sharded_iterations with all_reduce == explicit sum over the engines of this process"""
    costs = []
    lag = engs[0].cost_lags

    def total_cost():
        c = 0.0
        for e in engs:
            e._copy_cost(e._cost_t)
            c += float(e._cost_t.item())
        return c

    for it in range(iters):
        for e in engs:
            e.wstep_partial()
        if lag and it > 0:
            costs.append(total_cost())
        s = engs[0].packed.clone()
        for e in engs[1:]:
            s += e.packed
        for e in engs:
            e.packed.copy_(s)
            e.wstep_finish()
            e.hstep()
        if not lag:
            costs.append(total_cost())
    if lag:
        for e in engs:
            e.cost_pass()
        costs.append(total_cost())
    return np.array(costs)


@pytest.mark.parametrize("div", ["kl", "euclidean"])
@pytest.mark.parametrize("path,m,n,K", [(2, 256, 1024, 64), (1, 192, 300, 10)])
def test_two_shards_equal_oracle(gpu_lib, div, path, m, n, K):
    import torch
    from oracle import nmf_oracle as O
    from nmf_toolbox_amd.engine import shard_columns, torch_to_colmajor
    V, W0, H0 = synth(m, n, K)
    parts = [shard_columns(n, 2, r) for r in range(2)]
    engs = _engines(torch, V, W0, H0, div, parts, path)
    cost = _run_emulated(torch, engs, 15)
    W, H, c0 = O.nmf(V, K, dict(divergence=div, W_init=W0, H_init=H0, maxiter=15, tolerance=1e-300))
    Wg = [torch_to_colmajor(e.W).reshape(m, K) for e in engs]
    Hg = np.concatenate([torch_to_colmajor(e.H) for e in engs], axis=1)
    assert np.array_equal(Wg[0], Wg[1])                      # replicated W stays bit-identical
    assert rel_fro(Wg[0], W) < 1e-5 and rel_fro(Hg, H) < 1e-5 and rel_fro(cost, c0) < 1e-6


def test_baseline_sized_shards_match_unsharded(gpu_lib):
    """BASELINE config-3 per-GPU shard geometry (m=16384, K=256, 8192 columns per rank), two ranks, 2 iterations:
    sharded == unsharded on the same 16384 columns, cost decreases, columns unit-norm."""
    import torch
    from nmf_toolbox_amd.engine import Engine, shard_columns, torch_to_colmajor
    m, n, K = 16384, 16384, 256
    g = torch.Generator(device="cuda:0")
    g.manual_seed(1000)
    V = torch.rand((n, m), generator=g, device="cuda:0").clamp_(min=EPS)
    g.manual_seed(1)
    W0 = torch.rand((K, m), generator=g, device="cuda:0").clamp_(min=EPS)
    g.manual_seed(2)
    H0 = torch.rand((n, K), generator=g, device="cuda:0").clamp_(min=EPS)
    ref = Engine(V.clone(), W0.clone(), H0.clone(), divergence="kl", path=2, use_dist=False)
    ref.init()
    cref = torch.zeros(2, dtype=torch.float64, device="cuda:0")
    ref.iterate(2, cref)
    parts = [shard_columns(n, 2, r) for r in range(2)]
    engs = []
    from nmf_toolbox_amd import _lib
    for r, (lo, hi) in enumerate(parts):
        e = Engine(V[lo:hi].clone(), W0.clone(), H0[lo:hi].clone(), divergence="kl", path=2, use_dist=False)
        _lib.check(e.lib.nmfx_engine_set_rank0(e.h, 1 if r == 0 else 0))
        e.init()
        engs.append(e)
    cost = _run_emulated(torch, engs, 2)
    torch.cuda.synchronize()
    Wr = ref.W.double()
    assert torch.equal(engs[0].W, engs[1].W)
    relW = float((engs[0].W.double() - Wr).norm() / Wr.norm())
    Hs = torch.cat([e.H for e in engs], dim=0).double()
    relH = float((Hs - ref.H.double()).norm() / ref.H.double().norm())
    assert relW < 2e-6 and relH < 2e-6, (relW, relH)            # only the summation order of N differs
    cr = cref.cpu().numpy()
    assert np.allclose(cost, cr, rtol=1e-9) and cr[1] < cr[0]
    nrm = (ref.W.double() ** 2).sum(dim=1).sqrt()
    assert float((nrm - 1).abs().max()) < 1e-5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dist_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nmf_toolbox_amd.engine import Engine, colmajor_to_torch, shard_columns, torch_to_colmajor
    m, n, K = 256, 1024, 64
    V, W0, H0 = synth(m, n, K)
    lo, hi = shard_columns(n, world, rank)
    dev = "cuda:0"
    e = Engine(colmajor_to_torch(V[:, lo:hi], dev), colmajor_to_torch(W0, dev), colmajor_to_torch(H0[:, lo:hi], dev), divergence="kl")
    assert e.dist is not None and e.rank == rank
    e.init()
    cost = torch.zeros(10, dtype=torch.float64, device=dev)
    e.iterate(10, cost)
    torch.cuda.synchronize()
    q.put((rank, torch_to_colmajor(e.W).reshape(m, K), torch_to_colmajor(e.H), cost.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_torch_distributed_loop_two_processes(gpu_lib):
    import torch.multiprocessing as mp
    from oracle import nmf_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dist_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    m, n, K = 256, 1024, 64
    V, W0, H0 = synth(m, n, K)
    W, H, c0 = O.nmf(V, K, dict(divergence="kl", W_init=W0, H_init=H0, maxiter=10, tolerance=1e-300))
    assert np.array_equal(res[0][1], res[1][1])
    assert rel_fro(res[0][1], W) < 1e-5 and rel_fro(np.concatenate([res[0][2], res[1][2]], axis=1), H) < 1e-5
    assert rel_fro(res[0][3], c0) < 1e-6 and rel_fro(res[1][3], c0) < 1e-6


# ---- cnmf on column shards: halo columns of H / V, exchanged between neighbours --------------------
def _cnmf_shard_engines(torch, V, W0, H0, div, T, parts, path=0):
    from nmf_toolbox_amd.engine import Engine, colmajor_to_torch
    from nmf_toolbox_amd import _lib
    dev = "cuda:0"
    m, n = V.shape
    K = H0.shape[0]
    h = T - 1
    engs = []
    for r, (lo, hi) in enumerate(parts):
        hL = h if r > 0 else 0
        hR = h if r < len(parts) - 1 else 0
        Vx = V[:, lo:hi + hR]
        Hx = H0[:, lo - hL:hi + hR]
        e = Engine(colmajor_to_torch(Vx, dev), colmajor_to_torch(W0, dev), colmajor_to_torch(Hx, dev),
                   divergence=div, T=T, algorithm="cnmf", halo=(hL, hR), use_dist=False, path=path)
        _lib.check(e.lib.nmfx_engine_set_rank0(e.h, 1 if r == 0 else 0))
        e.init()
        engs.append(e)
    return engs


def _emulated_halo_exchange(engs, T):
    h = T - 1
    for r in range(len(engs)):
        if r > 0:
            engs[r].H[0:h].copy_(engs[r - 1].H_local[engs[r - 1].n - h:engs[r - 1].n])
        if r < len(engs) - 1:
            engs[r].H[engs[r].hL + engs[r].n:].copy_(engs[r + 1].H_local[0:h])


@pytest.mark.parametrize("div", ["euclidean", "kl"])
@pytest.mark.parametrize("nshards,m,n,K,T", [(2, 96, 200, 6, 4), (3, 128, 333, 8, 5), (2, 128, 333, 8, 5)])
def test_cnmf_shards_with_halos_equal_oracle(gpu_lib, div, nshards, m, n, K, T):
    import torch
    from oracle import nmf_oracle as O
    from nmf_toolbox_amd.engine import shard_columns, torch_to_colmajor
    V, W0, H0 = synth(m, n, K, T=T)
    # the oracle's init rescales H by the slab norms of W (cnmf.m:157-166): shards are cut from the raw H_init, every rank applies the same factors
    parts = [shard_columns(n, nshards, r) for r in range(nshards)]
    engs = _cnmf_shard_engines(torch, V, W0, H0, div, T, parts)
    iters = 10
    costs = []
    for it in range(iters):
        for e in engs:
            e.wstep_partial()
        s = engs[0].packed.clone()
        for e in engs[1:]:
            s += e.packed
        for e in engs:
            e.packed.copy_(s)
            e.wstep_finish()
            e.hstep()
        _emulated_halo_exchange(engs, T)
        for e in engs:
            e.hstep_finish()
        c = 0.0
        for e in engs:
            e._copy_cost(e._cost_t)
            c += float(e._cost_t.item())
        costs.append(c)
    W, H, c0 = O.cnmf(V, K, T, dict(divergence=div, W_init=W0, H_init=H0, maxiter=iters, tolerance=1e-300))
    Wg = torch_to_colmajor(engs[0].W)
    Hg = np.concatenate([torch_to_colmajor(e.H_local) for e in engs], axis=1)
    assert torch.equal(engs[0].W, engs[-1].W)
    assert rel_fro(Wg.reshape(W.shape), W) < 1e-5 and rel_fro(Hg, H) < 1e-5 and rel_fro(np.array(costs), c0) < 1e-6


def _cnmf_dist_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nmf_toolbox_amd.engine import Engine, colmajor_to_torch, shard_columns, torch_to_colmajor
    m, n, K, T = 128, 333, 8, 5
    V, W0, H0 = synth(m, n, K, T=T)
    lo, hi = shard_columns(n, world, rank)
    h = T - 1
    hL, hR = (h if rank > 0 else 0), (h if rank < world - 1 else 0)
    dev = "cuda:0"
    e = Engine(colmajor_to_torch(V[:, lo:hi + hR], dev), colmajor_to_torch(W0, dev), colmajor_to_torch(H0[:, lo - hL:hi + hR], dev),
               divergence="kl", T=T, algorithm="cnmf", halo=(hL, hR))
    assert e.dist is not None and e.has_halos
    e.init()
    cost = torch.zeros(8, dtype=torch.float64, device=dev)
    e.iterate(8, cost)
    torch.cuda.synchronize()
    q.put((rank, torch_to_colmajor(e.W), torch_to_colmajor(e.H_local), cost.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_cnmf_distributed_halo_exchange_two_processes(gpu_lib):
    """the real point-to-point halo exchange (torch.distributed batch_isend_irecv), two processes on cuda:0 over gloo"""
    import torch.multiprocessing as mp
    from oracle import nmf_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cnmf_dist_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    m, n, K, T = 128, 333, 8, 5
    V, W0, H0 = synth(m, n, K, T=T)
    W, H, c0 = O.cnmf(V, K, T, dict(divergence="kl", W_init=W0, H_init=H0, maxiter=8, tolerance=1e-300))
    assert np.array_equal(res[0][1], res[1][1])
    assert rel_fro(res[0][1].reshape(W.shape), W) < 1e-5 and rel_fro(np.concatenate([res[0][2], res[1][2]], axis=1), H) < 1e-5
    assert rel_fro(res[0][3], c0) < 1e-6 and rel_fro(res[1][3], c0) < 1e-6

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
    # Gram-form cost of the euclidean fused path: its (rank-independent) mode decision wants the GLOBAL ||V||^2 -- what Engine.init all-reduces
    # under torch.distributed is summed by hand here
    vv = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in engs]
    for e, t in zip(engs, vv):
        _lib.check(e.lib.nmfx_engine_sumvv_local(e.h, t.data_ptr()))
    tot = sum(vv)
    for e in engs:
        _lib.check(e.lib.nmfx_engine_sumvv_set_global(e.h, tot.data_ptr()))
    torch.cuda.synchronize()
    return engs


def _run_emulated(torch, engs, iters, nch=1):
    """run_sharded_iterations with all_reduce == explicit sum over the engines of this process (nch > 1: row-chunked W step)"""
    costs = []
    lag = engs[0].cost_lags
    lagk = engs[0].cost_lag          # 1: cost(it-1) is ready after wstep_partial(it); 2: after wstep_finish(it) (Gram-form cost)

    def total_cost():
        c = 0.0
        for e in engs:
            e._copy_cost(e._cost_t)
            c += float(e._cost_t.item())
        return c

    for it in range(iters):
        if nch > 1:
            for c in range(nch):
                for e in engs:
                    e.wstep_partial_chunk(c, nch)
                s = engs[0].packed_chunk(c, nch).clone()
                for e in engs[1:]:
                    s += e.packed_chunk(c, nch)
                for e in engs:
                    e.packed_chunk(c, nch).copy_(s)
        else:
            for e in engs:
                e.wstep_partial()
            s = engs[0].packed.clone()
            for e in engs[1:]:
                s += e.packed
            for e in engs:
                e.packed.copy_(s)
        if lag and it > 0 and (lagk == 1 or nch > 1):
            costs.append(total_cost())
        for e in engs:
            e.wstep_finish()
        if lagk == 2 and nch == 1 and it > 0:
            costs.append(total_cost())
        for e in engs:
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


@pytest.mark.parametrize("div", ["kl", "euclidean"])
@pytest.mark.parametrize("nshards", [2, 3, 8])
def test_shards_run_to_run_determinism(gpu_lib, div, nshards):
    """the same column-sharded run twice: bit-identical W (on every shard), H and cost"""
    import torch
    from nmf_toolbox_amd.engine import shard_columns, torch_to_colmajor
    m, n, K = 256, 8 * 72 * 2, 64
    V, W0, H0 = synth(m, n, K)
    parts = [shard_columns(n, nshards, r) for r in range(nshards)]
    res = []
    for _ in range(2):
        engs = _engines(torch, V, W0, H0, div, parts, 2)
        cost = _run_emulated(torch, engs, 6)
        res.append(([e.W.clone() for e in engs], [e.H.clone() for e in engs], cost))
        for e in engs:
            e.close()
    for a, b in zip(res[0][0] + res[0][1], res[1][0] + res[1][1]):
        assert torch.equal(a, b)
    assert all(torch.equal(w, res[0][0][0]) for w in res[0][0])
    assert np.array_equal(res[0][2], res[1][2])


@pytest.mark.parametrize("div", ["kl", "euclidean"])
@pytest.mark.parametrize("nch", [2, 4])
def test_row_chunked_wstep_equals_oracle(gpu_lib, div, nch):
    """The W-step partial computed in row chunks (what lets the all-reduce of one chunk overlap the compute of the next) gives the
    same factors: chunked `packed` layout, chunk-aware W update, lagged cost assembled from all chunks' partials."""
    import torch
    from oracle import nmf_oracle as O
    from nmf_toolbox_amd.engine import torch_to_colmajor
    m, n, K = 512, 1024, 64
    V, W0, H0 = synth(m, n, K)
    engs = _engines(torch, V, W0, H0, div, [(0, 512), (512, 1024)], 2)
    costs = _run_emulated(torch, engs, 8, nch=nch)
    torch.cuda.synchronize()
    W, H, c0 = O.nmf(V, K, dict(divergence=div, W_init=W0, H_init=H0, maxiter=8, tolerance=1e-300))
    Wg = torch_to_colmajor(engs[0].W).reshape(m, K)
    Hg = np.concatenate([torch_to_colmajor(e.H) for e in engs], axis=1)
    assert np.array_equal(Wg, torch_to_colmajor(engs[1].W).reshape(m, K))
    assert rel_fro(Wg, W) < 1e-5 and rel_fro(Hg, H) < 1e-5 and rel_fro(costs, c0) < 1e-6
    with pytest.raises(Exception, match="multiples of 128"):
        engs[0].wstep_partial_chunk(0, 3)


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


def _switch_data(m, n, K, noise=0.33):
    """a rank-K product plus enough noise that 0.5||V - W*H||^2 starts above 5 % of 0.5||V||^2 and sinks below it after a few iterations: the Gram-form cost of the
    euclidean fused path switches to the explicit residual pass there"""
    rs = np.random.RandomState
    V = rs(1001).rand(m, K) @ rs(1002).rand(K, n) / K + noise * np.fmax(rs(1000).rand(m, n), EPS)
    _, W0, H0 = synth(m, n, K)
    return V, W0, H0


def _dist_worker(rank, world, port, q, K=64, div="kl", n_chunks=2, kind=1, lag=1, planted=False, iters=10):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nmf_toolbox_amd.engine import Engine, colmajor_to_torch, shard_columns, torch_to_colmajor
    m, n = 256, 1024
    V, W0, H0 = _switch_data(m, n, K) if planted else synth(m, n, K)
    lo, hi = shard_columns(n, world, rank)
    dev = "cuda:0"
    e = Engine(colmajor_to_torch(V[:, lo:hi], dev), colmajor_to_torch(W0, dev), colmajor_to_torch(H0[:, lo:hi], dev), divergence=div,
               n_chunks=n_chunks)                   # (2: also exercises the row-chunked, async all-reduce form of the W step)
    assert e.dist is not None and e.rank == rank and e.n_chunks == n_chunks and e.path_kind == kind and e.cost_lag == lag
    e.init()
    cost = torch.zeros(iters, dtype=torch.float64, device=dev)
    e.iterate(iters, cost)
    torch.cuda.synchronize()
    q.put((rank, torch_to_colmajor(e.W).reshape(m, K), torch_to_colmajor(e.H), cost.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("K,div,n_chunks,kind,lag", [(64, "kl", 2, 1, 1), (64, "euclidean", 1, 1, 2), (320, "kl", 1, 5, 1), (320, "euclidean", 1, 6, 2)])
def test_torch_distributed_loop_two_processes(gpu_lib, K, div, n_chunks, kind, lag):
    """run_sharded_iterations with real processes (gloo, both on cuda:0): the fused KL kernels with row chunks, the euclidean fused path with its Gram-form cost
    (lag 2, the global ||V||^2 all-reduced at init), and K > 256 in column blocks for both divergences (paths 5 and 6)"""
    import torch.multiprocessing as mp
    from oracle import nmf_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dist_worker, args=(r, 2, port, q, K, div, n_chunks, kind, lag)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    m, n = 256, 1024
    V, W0, H0 = synth(m, n, K)
    W, H, c0 = O.nmf(V, K, dict(divergence=div, W_init=W0, H_init=H0, maxiter=10, tolerance=1e-300))
    assert np.array_equal(res[0][1], res[1][1])
    assert rel_fro(res[0][1], W) < 1e-5 and rel_fro(np.concatenate([res[0][2], res[1][2]], axis=1), H) < 1e-5
    assert rel_fro(res[0][3], c0) < 1e-6 and rel_fro(res[1][3], c0) < 1e-6


def test_torch_distributed_loop_through_the_gram_cost_switch(gpu_lib):
    """Two processes, euclidean fused path (cost lag 2), PLANTED data: the residual falls below 5 % of ||V||^2 within a few iterations, the device-side flag turns the
    explicit residual pass on and, two W updates later, both ranks go back to the one-pass kernel -- from where on the cost of an iteration turns up one phase
    EARLIER.  The merged loop of round 3 read it after the next wstep_partial had already overwritten it (every entry from the switch on shifted by one; advisor,
    r3): the cost vector must equal the unsharded oracle's entry for entry, on both ranks, through the switch."""
    import torch.multiprocessing as mp
    from oracle import nmf_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    K, iters = 64, 40
    procs = [ctx.Process(target=_dist_worker, args=(r, 2, port, q, K, "euclidean", 1, 1, 2, True, iters)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    m, n = 256, 1024
    V, W0, H0 = _switch_data(m, n, K)
    W, H, c0 = O.nmf(V, K, dict(divergence="euclidean", W_init=W0, H_init=H0, maxiter=iters, tolerance=1e-300))
    cross = int(np.argmax(c0 < 0.05 * 0.5 * np.sum(V * V)))
    assert 3 <= cross <= iters - 6, cross                                               # the run really crosses the switch, with iterations to spare on both sides
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][3], res[1][3])   # W and the cost vector bit-identical on both ranks
    assert rel_fro(res[0][1], W) < 1e-5 and rel_fro(np.concatenate([res[0][2], res[1][2]], axis=1), H) < 1e-5
    worst = float(np.max(np.abs(res[0][3] - c0) / c0))
    assert worst < 1e-6, (worst, res[0][3][:8], c0[:8])                                  # entry for entry: a shifted vector is off by the iteration-to-iteration decrease


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
    # Gram-form cost of the euclidean fused path: its (rank-independent) mode decision wants the GLOBAL ||V||^2 -- what Engine.init all-reduces
    # under torch.distributed is summed by hand here
    vv = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in engs]
    for e, t in zip(engs, vv):
        _lib.check(e.lib.nmfx_engine_sumvv_local(e.h, t.data_ptr()))
    tot = sum(vv)
    for e in engs:
        _lib.check(e.lib.nmfx_engine_sumvv_set_global(e.h, tot.data_ptr()))
    torch.cuda.synchronize()
    return engs


def _emulated_halo_exchange(engs, T):
    h = T - 1
    for r in range(len(engs)):
        if r > 0:
            engs[r].H[0:h].copy_(engs[r - 1].H_local[engs[r - 1].n - h:engs[r - 1].n])
        if r < len(engs) - 1:
            engs[r].H[engs[r].hL + engs[r].n:].copy_(engs[r + 1].H_local[0:h])


@pytest.mark.parametrize("div", ["euclidean", "kl"])
@pytest.mark.parametrize("nshards,m,n,K,T", [(2, 96, 200, 6, 4), (3, 128, 333, 8, 5), (2, 128, 333, 8, 5),
                                             (2, 256, 520, 64, 4), (3, 192, 777, 32, 8), (4, 129, 1024, 64, 2)])   # the last three: fused shift-sum passes, halos as the left context
def test_cnmf_shards_with_halos_equal_oracle(gpu_lib, div, nshards, m, n, K, T):
    import torch
    from oracle import nmf_oracle as O
    from nmf_toolbox_amd.engine import shard_columns, torch_to_colmajor
    V, W0, H0 = synth(m, n, K, T=T)
    # the oracle's init rescales H by the slab norms of W (cnmf.m:157-166): shards are cut from the raw H_init, every rank applies the same factors
    parts = [shard_columns(n, nshards, r) for r in range(nshards)]
    engs = _cnmf_shard_engines(torch, V, W0, H0, div, T, parts)
    if K >= 32:       # instantiated (K, T) pairs: the fused shift-sum passes on every shard, for KL too (R = V./V_hat also on the right-halo columns)
        assert all(e.path_kind == (4 if div == "kl" else 3) for e in engs)
    iters = 10
    costs = []
    lag = engs[0].cost_lag          # where the cost of an iteration turns up: 0 after its H step, 1 after the next W-step partial (KL on the fused passes: out of the S pass)
    assert lag in (0, 1) and all(e.cost_lag == lag for e in engs)

    def read_cost():
        c = 0.0
        for e in engs:
            e._copy_cost(e._cost_t)
            c += float(e._cost_t.item())
        costs.append(c)

    for it in range(iters):
        for e in engs:
            e.wstep_partial()
        if lag == 1 and it > 0:
            read_cost()
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
        if lag == 0:
            read_cost()
    if lag == 1:
        for e in engs:
            e.cost_pass()
        read_cost()
    W, H, c0 = O.cnmf(V, K, T, dict(divergence=div, W_init=W0, H_init=H0, maxiter=iters, tolerance=1e-300))
    Wg = torch_to_colmajor(engs[0].W)
    Hg = np.concatenate([torch_to_colmajor(e.H_local) for e in engs], axis=1)
    assert torch.equal(engs[0].W, engs[-1].W)
    assert rel_fro(Wg.reshape(W.shape), W) < 1e-5 and rel_fro(Hg, H) < 1e-5 and rel_fro(np.array(costs), c0) < 1e-6


def _cnmf_dist_worker(rank, world, port, q, K=8, T=5, div="kl"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nmf_toolbox_amd.engine import Engine, colmajor_to_torch, shard_columns, torch_to_colmajor
    m, n = 128, 333
    V, W0, H0 = synth(m, n, K, T=T)
    lo, hi = shard_columns(n, world, rank)
    h = T - 1
    hL, hR = (h if rank > 0 else 0), (h if rank < world - 1 else 0)
    dev = "cuda:0"
    e = Engine(colmajor_to_torch(V[:, lo:hi + hR], dev), colmajor_to_torch(W0, dev), colmajor_to_torch(H0[:, lo - hL:hi + hR], dev),
               divergence=div, T=T, algorithm="cnmf", halo=(hL, hR))
    assert e.dist is not None and e.has_halos
    if K % 32 == 0:       # an instantiated pair: the fused passes on both shards; with KL the cost lags one pass (out of the next S pass)
        assert e.path_kind == (4 if div == "kl" else 3) and e.cost_lag == (1 if div == "kl" else 0)
    e.init()
    cost = torch.zeros(8, dtype=torch.float64, device=dev)
    e.iterate(8, cost)
    torch.cuda.synchronize()
    q.put((rank, torch_to_colmajor(e.W), torch_to_colmajor(e.H_local), cost.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("K,T,div", [(8, 5, "kl"), (32, 4, "kl"), (64, 2, "euclidean")])
def test_cnmf_distributed_halo_exchange_two_processes(gpu_lib, K, T, div):
    """the real point-to-point halo exchange (torch.distributed batch_isend_irecv), two processes on cuda:0 over gloo -- on the general kernels (K = 8) and on
    the fused shift-sum passes, where KL brings a lagged cost through run_sharded_iterations together with the halos"""
    import torch.multiprocessing as mp
    from oracle import nmf_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cnmf_dist_worker, args=(r, 2, port, q, K, T, div)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    m, n = 128, 333
    V, W0, H0 = synth(m, n, K, T=T)
    W, H, c0 = O.cnmf(V, K, T, dict(divergence=div, W_init=W0, H_init=H0, maxiter=8, tolerance=1e-300))
    assert np.array_equal(res[0][1], res[1][1])
    assert rel_fro(res[0][1].reshape(W.shape), W) < 1e-5 and rel_fro(np.concatenate([res[0][2], res[1][2]], axis=1), H) < 1e-5
    assert rel_fro(res[0][3], c0) < 1e-6 and rel_fro(res[1][3], c0) < 1e-6


# ---- nmfsc on column shards: distributed projfunc reductions, all-reduce through the callback ------
class _ThreadRendezvous:
    """all-reduce among `world` threads of ONE process that each drive a shard on the same GPU (stand-in for RCCL ranks)"""

    def __init__(self, world):
        import threading
        self.world, self.slots, self.result = world, [None] * world, None
        self.barrier = threading.Barrier(world)

    def allreduce(self, rank, t, op):
        import torch
        self.slots[rank] = t
        self.barrier.wait()
        if rank == 0:
            st = torch.stack(self.slots)                    # fixed rank order: every "rank" gets bit-identical sums
            self.result = st.sum(0) if op == 0 else st.max(0).values
        self.barrier.wait()
        t.copy_(self.result)
        self.barrier.wait()


def _nmfsc_threads(V, W0, H0, world, **kw):
    import threading
    import torch
    from nmf_toolbox_amd.engine import colmajor_to_torch, nmfsc_sharded, shard_columns, torch_to_colmajor
    rv = _ThreadRendezvous(world)
    out, errs = [None] * world, []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            lo, hi = shard_columns(V.shape[1], world, rank)
            Vt, Wt, Ht = colmajor_to_torch(V[:, lo:hi], "cuda:0"), colmajor_to_torch(W0, "cuda:0"), colmajor_to_torch(H0[:, lo:hi], "cuda:0")
            cost, info = nmfsc_sharded(Vt, Wt, Ht, allreduce=lambda t, op: rv.allreduce(rank, t, op), **kw)
            torch.cuda.synchronize()
            out[rank] = (torch_to_colmajor(Wt).reshape(W0.shape), torch_to_colmajor(Ht), cost, info)
        except Exception as ex:   # a dead thread would leave the others in the barrier forever
            errs.append(ex)
            rv.barrier.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(timeout=600) for t in th]
    if errs:
        raise errs[0]
    return out


@pytest.mark.parametrize("sW,sH", [(0.0, 0.5), (0.4, 0.6), (0.0, 0.0), (0.3, 0.0)])
@pytest.mark.parametrize("world,path", [(2, 2), (4, 2), (2, 0)])   # path 2: the fused MFMA kernels by name; 0: what a problem this small gets by default (float64 gradients)
def test_nmfsc_column_shards_equal_oracle(gpu_lib, sW, sH, world, path):
    from oracle import nmf_oracle as O
    m, n, K = 256, 1024, 64
    V, W0, H0 = synth(m, n, K)
    V = 2.5 * V                                                # the global max(V) rescale (nmfsc.m:62) spans the shards
    cfg = dict(W_init=W0, H_init=H0, maxiter=12, tolerance=1e-12)
    if sW:
        cfg["W_sparsity"] = sW
    if sH:
        cfg["H_sparsity"] = sH
    i0 = {}
    W, H, cost = O.nmfsc(V, K, cfg, info=i0)
    res = _nmfsc_threads(V, W0, H0, world, W_sparsity=sW, H_sparsity=sH, maxiter=12, tolerance=1e-12, path=path)
    for r in res[1:]:
        assert np.array_equal(r[0], res[0][0]) and np.array_equal(r[2], res[0][2])      # W and cost replicated bit-for-bit
    Hs = np.concatenate([r[1] for r in res], axis=1)
    assert res[0][3]["triesH"] == i0["triesH"] and res[0][3]["triesW"] == i0["triesW"]
    assert rel_fro(res[0][0], W) <= 1e-5 and rel_fro(Hs, H) <= 1e-5, (rel_fro(res[0][0], W), rel_fro(Hs, H))
    assert len(res[0][2]) == len(cost) and rel_fro(res[0][2], cost) <= 1e-6
    if sH:                                                     # Hoyer postconditions hold on WHOLE rows of H (projfunc.m:3-7)
        L1s = np.sqrt(n) - (np.sqrt(n) - 1) * sH
        assert np.allclose(Hs.sum(1), L1s, rtol=1e-5) and np.allclose((Hs ** 2).sum(1), 1.0, rtol=1e-5) and Hs.min() >= 0


def test_nmfsc_sharded_any_K_and_negative_data(gpu_lib):
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(256, 600, 20)                            # K = 20: padded to 32 inside the library, invisible outside
    for sW, sH in ((0.0, 0.5), (0.3, 0.0), (0.4, 0.6)):
        i0 = {}
        cfg = dict(W_init=W0, H_init=H0, maxiter=8, tolerance=1e-12, W_sparsity=sW, H_sparsity=sH)
        W, H, cost = O.nmfsc(V, 20, cfg, info=i0)
        res = _nmfsc_threads(V, W0, H0, 2, W_sparsity=sW, H_sparsity=sH, maxiter=8, tolerance=1e-12)
        Hs = np.concatenate([r[1] for r in res], axis=1)
        assert res[0][3]["triesH"] == i0["triesH"] and res[0][3]["triesW"] == i0["triesW"]
        assert rel_fro(res[0][0], W) <= 1e-5 and rel_fro(Hs, H) <= 1e-5 and rel_fro(res[0][2], cost) <= 1e-6
    with pytest.raises(Exception, match="fused kernels only"):
        _nmfsc_threads(*synth(256, 600, 300), 2, H_sparsity=0.5, maxiter=2)   # K > 256 has no sharded path
    Vn = synth(256, 512, 64)[0]
    Vn[3, 400] = -1.0                                          # only the second shard sees it: the check is global
    W0, H0 = synth(256, 512, 64)[1:]
    with pytest.raises(ValueError, match="Negative values in data!"):
        _nmfsc_threads(Vn, W0, H0, 2, maxiter=2)


@pytest.mark.parametrize("K", [3, 8])
def test_nmfsc_small_K_column_shards_equal_oracle(gpu_lib, K):
    """K <= 8 on column shards: the float64 gradient kernels (aux.hip::smallk_grad), dW summed over the shards as doubles, the
    distributed projfunc stepping along a float64 direction"""
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(200, 640, K)
    for sW, sH in ((0.0, 0.7), (0.3, 0.4), (0.5, 0.0)):
        i0 = {}
        cfg = dict(W_init=W0, H_init=H0, maxiter=6, tolerance=1e-300, W_sparsity=sW, H_sparsity=sH)
        W, H, cost = O.nmfsc(V, K, cfg, info=i0)
        res = _nmfsc_threads(V, W0, H0, 2, W_sparsity=sW, H_sparsity=sH, maxiter=6, tolerance=1e-300)
        Hs = np.concatenate([r[1] for r in res], axis=1)
        assert res[0][3]["triesH"] == i0["triesH"] and res[0][3]["triesW"] == i0["triesW"]
        assert np.array_equal(res[1][0], res[0][0])
        assert rel_fro(res[0][0], W) <= 1e-5 and rel_fro(Hs, H) <= 1e-5 and rel_fro(res[0][2], cost) <= 1e-6, (K, sW, sH, rel_fro(res[0][0], W), rel_fro(Hs, H))


def test_nmfsc_ragged_column_shards_equal_oracle(gpu_lib):
    """300 + 300 columns (not multiples of 128) and m = 257: the masked-edge kernels under the sharded nmfsc"""
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(257, 600, 32)
    cfg = dict(W_init=W0, H_init=H0, W_sparsity=0.3, H_sparsity=0.5, maxiter=8, tolerance=1e-12)
    i0 = {}
    W, H, cost = O.nmfsc(V, 32, cfg, info=i0)
    res = _nmfsc_threads(V, W0, H0, 2, W_sparsity=0.3, H_sparsity=0.5, maxiter=8, tolerance=1e-12, path=2)
    Hs = np.concatenate([r[1] for r in res], axis=1)
    assert res[0][3]["triesH"] == i0["triesH"] and res[0][3]["triesW"] == i0["triesW"]
    assert rel_fro(res[0][0], W) <= 1e-5 and rel_fro(Hs, H) <= 1e-5 and rel_fro(res[0][2], cost) <= 1e-6, (rel_fro(res[0][0], W), rel_fro(Hs, H))


def _nmfsc_dist_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nmf_toolbox_amd.engine import colmajor_to_torch, nmfsc_sharded, shard_columns, torch_to_colmajor
    m, n, K = 256, 1024, 64
    V, W0, H0 = synth(m, n, K)
    lo, hi = shard_columns(n, world, rank)
    Vt, Wt, Ht = colmajor_to_torch(V[:, lo:hi], "cuda:0"), colmajor_to_torch(W0, "cuda:0"), colmajor_to_torch(H0[:, lo:hi], "cuda:0")
    cost, info = nmfsc_sharded(Vt, Wt, Ht, W_sparsity=0.3, H_sparsity=0.5, maxiter=8, tolerance=1e-12)
    torch.cuda.synchronize()
    q.put((rank, torch_to_colmajor(Wt).reshape(m, K), torch_to_colmajor(Ht), cost, info["triesH"], info["triesW"]))
    dist.barrier()
    dist.destroy_process_group()


def test_nmfsc_torch_distributed_two_processes(gpu_lib):
    import torch.multiprocessing as mp
    from oracle import nmf_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nmfsc_dist_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    V, W0, H0 = synth(256, 1024, 64)
    i0 = {}
    W, H, cost = O.nmfsc(V, 64, dict(W_init=W0, H_init=H0, W_sparsity=0.3, H_sparsity=0.5, maxiter=8, tolerance=1e-12), info=i0)
    assert np.array_equal(res[0][1], res[1][1]) and res[0][4] == i0["triesH"] and res[0][5] == i0["triesW"]
    assert rel_fro(res[0][1], W) <= 1e-5 and rel_fro(np.concatenate([res[0][2], res[1][2]], axis=1), H) <= 1e-5
    assert rel_fro(res[0][3], cost) <= 1e-6


# ---- n_gpus behind the blocking C ABI (include/nmfx.h: nmfx_problem.n_gpus / device_ids): one process, one stream + engine per shard,
# peer reduce-scatter + all-gather of `packed`.  The test box has ONE GPU: device_ids = [0, 0, ...] puts every shard on it, which runs
# the same sharding, exchange and event code the 8-GPU node runs (minus the xGMI hop).
@pytest.mark.parametrize("div,m,n,K,path", [("kl", 256, 1024, 64, 2), ("euclidean", 256, 1024, 64, 2), ("kl", 192, 333, 12, 0), ("is", 160, 300, 8, 1),
                                            ("kl", 513, 1000, 100, 2)])
@pytest.mark.parametrize("ndev", [2, 3, 8])
def test_blocking_api_n_gpus_matches_oracle(gpu_lib, div, m, n, K, path, ndev):
    from oracle import nmf_oracle as O
    from conftest import record_err
    V, W0, H0 = synth(m, n, K)
    cfg = dict(divergence=div, W_init=W0, H_init=H0, maxiter=12, tolerance=1e-12, W_sparsity=0.01, H_sparsity=0.02)
    ref = O.nmf(V, K, cfg)
    got = gpu_lib.nmf(V, K, dict(cfg, nmfx_path=path, nmfx_gpus=[0] * ndev))
    one = gpu_lib.nmf(V, K, dict(cfg, nmfx_path=path))
    assert len(got[2]) == len(ref[2])
    e = dict(W=rel_fro(got[0], ref[0]), H=rel_fro(got[1], ref[1]), cost=rel_fro(got[2], ref[2]))
    record_err(**e)
    assert e["W"] <= 1e-5 and e["H"] <= 1e-5 and e["cost"] <= (1e-5 if div == "is" else 1e-6), e
    assert rel_fro(got[0], one[0]) <= 2e-6 and rel_fro(got[1], one[1]) <= 2e-6      # vs one shard: only the summation order of N differs


def test_blocking_api_n_gpus_stop_rule_multi_source_and_lnmf(gpu_lib):
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(256, 512, 64, planted=True)
    cfg = dict(divergence="euclidean", W_init=W0, H_init=H0, maxiter=300, tolerance=5e-2, nmfx_path=2)
    ref = O.nmf(V, 64, cfg)
    got = gpu_lib.nmf(V, 64, dict(cfg, nmfx_gpus=[0, 0, 0, 0]))
    assert len(ref[2]) < 300 and len(got[2]) == len(ref[2])                        # the stop rule fires at the same iteration on 4 shards
    Ks = [24, 40]
    cfg = dict(divergence="kl", W_init=[W0[:, :24], W0[:, 24:]], H_init=[H0[:24], H0[24:]], W_sparsity=[0.05, 0.0], H_sparsity=[0.0, 0.1],
               W_fixed=[False, True], maxiter=20, tolerance=1e-12)
    ref = O.nmf(V, Ks, cfg)
    got = gpu_lib.nmf(V, Ks, dict(cfg, nmfx_gpus=2 * [0]))
    assert rel_fro(np.hstack(got[0]), np.hstack(ref[0])) <= 1e-5 and rel_fro(np.vstack(got[1]), np.vstack(ref[1])) <= 1e-5 and rel_fro(got[2], ref[2]) <= 1e-6
    cfg = dict(W_init=W0 / W0.sum(0), H_init=H0, maxiter=10, tolerance=1e-12)
    ref = O.lnmf(V, 64, cfg)
    got = gpu_lib.lnmf(V, 64, dict(cfg, nmfx_gpus=[0, 0, 0]))
    # planted (well-fitting) V: the KL cost (53) is 1/600 of sum(V) (32768), so the -5e-9 relative bias of v_rcp_f32 / v_log_f32 in
    # sum(V .* log(V ./ V_hat)) shows as 2.1e-6 of the cost (measured; identical on one shard) -- conditioning of the cost value, not
    # of the factors: W 3e-7, H 6e-8.  On non-planted data the same path holds 1e-6 (test_lnmf_matches_oracle and below).
    assert rel_fro(got[0], ref[0]) <= 1e-5 and rel_fro(got[1], ref[1]) <= 1e-5 and rel_fro(got[2], ref[2]) <= 5e-6
    V2, W2, H2 = synth(256, 512, 64)
    cfg = dict(W_init=W2 / W2.sum(0), H_init=H2, maxiter=10, tolerance=1e-12)
    ref = O.lnmf(V2, 64, cfg)
    got = gpu_lib.lnmf(V2, 64, dict(cfg, nmfx_gpus=[0, 0, 0]))
    assert rel_fro(got[0], ref[0]) <= 1e-5 and rel_fro(got[1], ref[1]) <= 1e-5 and rel_fro(got[2], ref[2]) <= 1e-6
    with pytest.raises(Exception, match="n_gpus > 1 is not implemented"):
        gpu_lib.cnmfsc(V, 8, 2, dict(maxiter=1, nmfx_gpus=[0, 0]))
    with pytest.raises(Exception, match="every shard needs at least T-1"):
        gpu_lib.cnmf(V[:, :20], 8, 6, dict(maxiter=1, nmfx_gpus=8 * [0]))
    with pytest.raises(Exception):
        gpu_lib.nmf(V, 64, dict(maxiter=1, nmfx_gpus=[0, 7]))                      # no such device on a 1-GPU box


@pytest.mark.parametrize("sW,sH", [(0.0, 0.5), (0.3, 0.5)])
def test_nmfsc_dev_resume_equals_one_run(gpu_lib, sW, sH):
    """nmfx_problem.sc_resume / sc_stepsize_*0 (what bench.py --workload c5 times): a + b outer iterations in two calls against a + b in one -- same
    line-search tries, same factors.  With both line searches active the objective that closes the first call comes from another kernel than
    inside one long run (a cost-only pass instead of the speculative residual pass: another summation order), so W / H / cost are compared at
    the contract, not bit for bit."""
    import torch
    from nmf_toolbox_amd.engine import nmfsc_sharded, colmajor_to_torch
    m, n, K, a, b = 256, 1024, 64, 3, 3
    V, W0, H0 = synth(m, n, K)
    dev = "cuda:0"
    mk = lambda: (colmajor_to_torch(V, dev), colmajor_to_torch(W0, dev), colmajor_to_torch(H0, dev))
    kw = dict(W_sparsity=sW, H_sparsity=sH, tolerance=-1.0, path=2)
    V1, W1, H1 = mk()
    c_one, i_one = nmfsc_sharded(V1, W1, H1, maxiter=a + b, **kw)
    V2, W2, H2 = mk()
    c_a, i_a = nmfsc_sharded(V2, W2, H2, maxiter=a, **kw)
    c_b, i_b = nmfsc_sharded(V2, W2, H2, maxiter=b, resume=i_a, **kw)
    assert i_a["triesH"] + i_b["triesH"] == i_one["triesH"] and i_a["triesW"] + i_b["triesW"] == i_one["triesW"]
    assert abs(i_b["stepsizeH"] - i_one["stepsizeH"]) <= 1e-12 * i_one["stepsizeH"]
    rel = lambda x, y: float((x.double() - y.double()).norm() / y.double().norm())
    assert rel(W2, W1) < 1e-5 and rel(H2, H1) < 1e-5
    both = np.concatenate([c_a, c_b[1:]])
    assert len(both) == len(c_one) and np.max(np.abs(both - c_one) / c_one) < 1e-6 and abs(c_b[0] - c_a[-1]) <= 1e-6 * c_a[-1]


def test_projfunc_dev_equals_host_entry_point(gpu_lib):
    """nmfx_projfunc_dev (device buffers, asynchronous, optional fused step src + mu*dir) against nmfx_projfunc on the same fp32 vectors"""
    import ctypes as C
    import torch
    from nmf_toolbox_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(4)
    N, count = 3000, 5
    S = rs.rand(N, count).astype(np.float32)
    D = rs.randn(N, count).astype(np.float32)
    k1 = np.sqrt(N) - (np.sqrt(N) - 1) * 0.6
    X = torch.from_numpy(np.ascontiguousarray(S.T)).cuda()
    it = torch.zeros(count, dtype=torch.int32, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.nmfx_projfunc_dev(st, X.data_ptr(), N, count, float(k1), 1.0, 1, None, None, 0.0, it.data_ptr()))
    torch.cuda.synchronize()
    want, wit = zip(*[gpu_lib.projfunc(S[:, c].astype(np.float32), k1, 1.0, True) for c in range(count)])
    got = X.cpu().numpy().T
    for c in range(count):
        assert rel_fro(got[:, c], np.asarray(want[c]).ravel()) < 2e-6 and int(it[c]) == int(wit[c])
    # fused step: projection of src + mu*dir, formed in fp64 while loading
    Xs, Dd, Out = torch.from_numpy(np.ascontiguousarray(S.T)).cuda(), torch.from_numpy(np.ascontiguousarray(D.T)).cuda(), torch.zeros(count, N, device="cuda")
    _lib.check(lib.nmfx_projfunc_dev(st, Out.data_ptr(), N, count, float(k1), 1.0, 1, Xs.data_ptr(), Dd.data_ptr(), -0.05, None))
    torch.cuda.synchronize()
    for c in range(count):
        w, _ = gpu_lib.projfunc(S[:, c].astype(np.float64) - 0.05 * D[:, c].astype(np.float64), k1, 1.0, True)
        assert rel_fro(Out[c].cpu().numpy(), np.asarray(w).ravel()) < 2e-6


@pytest.mark.parametrize("div,path", [("euclidean", 2), ("kl", 2), ("euclidean", 1)])     # cost_lag 2 (Gram-form cost), 1 (fused KL), 0 (materialised V_hat)
def test_engine_loop_stop_rule_equals_blocking_call(gpu_lib, div, path):
    """nmf.m:221-224 inside the device-level loop (Engine.iterate(..., tolerance)): same number of iterations, same W / H / cost as the blocking call,
    whose stop logic is a separate implementation (csrc/blocking.hip) -- and as the oracle"""
    import torch
    from oracle import nmf_oracle as O
    from nmf_toolbox_amd.engine import Engine, colmajor_to_torch, torch_to_colmajor
    m, n, K = 256, 1024, 64
    V, W0, H0 = synth(m, n, K, planted=True)
    probe = O.nmf(V, K, dict(divergence=div, W_init=W0, H_init=H0, maxiter=30, tolerance=1e-300))[2]
    dec = -np.diff(probe)
    tol = float(0.5 * (dec[11] + dec[12]))
    assert np.all(dec[:14] > 0) and dec[11] > dec[12]
    cfg = dict(divergence=div, W_init=W0, H_init=H0, maxiter=30, tolerance=tol)
    Wr, Hr, cr = O.nmf(V, K, cfg)
    # float32 host arrays: the blocking call then starts its float64 master copies from the same fp32 values the device-level Engine is handed (with float64
    # host arrays it starts them from the doubles themselves, nmfx_engine_init_f64 -- closer to the oracle, but no longer the same run bit for bit)
    f32 = np.float32
    Wb, Hb, cb = gpu_lib.nmf(V.astype(f32), K, dict(cfg, W_init=W0.astype(f32), H_init=H0.astype(f32), nmfx_path=path))
    e = Engine(colmajor_to_torch(V, "cuda:0"), colmajor_to_torch(W0, "cuda:0"), colmajor_to_torch(H0, "cuda:0"), divergence=div, path=path, use_dist=False)
    e.init()
    cost = torch.zeros(30, dtype=torch.float64, device="cuda:0")
    ran = e.iterate(30, cost, tolerance=tol)
    assert ran == len(cb) == len(cr) and 5 < ran < 30
    We, He = torch_to_colmajor(e.W).reshape(m, K), torch_to_colmajor(e.H)
    assert np.array_equal(We, Wb) and np.array_equal(He, Hb)                 # the two stop implementations hand back the same state, bit for bit
    # KL on planted (well-fitting) data: the cost is a small difference of sums of the size of sum(V), and the v_rcp / v_log element map carries a
    # systematic -5e-9 * sum(V) (constant over the iterations, so it cancels in the differences the stop rule looks at): 1.6e-6 of the cost here
    assert rel_fro(We, Wr) < 1e-5 and rel_fro(He, Hr) < 1e-5 and rel_fro(cost[:ran].cpu().numpy(), cr) < (3e-6 if div == "kl" else 1e-6)
    assert np.max(np.abs(np.diff(cost[:ran].cpu().numpy()) - np.diff(cr))) < 1e-4 * tol
    e.close()


@pytest.mark.parametrize("div", ["euclidean", "kl"])
@pytest.mark.parametrize("ndev,m,n,K,T", [(2, 96, 200, 6, 4), (3, 128, 333, 8, 5), (8, 192, 1030, 32, 8), (2, 256, 520, 64, 4), (4, 129, 1024, 64, 2)])
def test_blocking_api_n_gpus_cnmf_matches_oracle(gpu_lib, div, ndev, m, n, K, T):
    """cnmf.m behind the blocking call on N column shards of one process (nmfx_problem.n_gpus; what a MEX caller of cnmf() gets): T-1 halo columns
    of H copied between neighbouring devices after every H update (cnmf.m:188,219 shift across the shard edges), the packed W-step sums through the
    peer reduce-scatter / all-gather.  device_ids names the one GPU of the box N times."""
    from oracle import nmf_oracle as O
    from conftest import record_err
    V, W0, H0 = synth(m, n, K, T=T)
    cfg = dict(divergence=div, W_init=W0, H_init=H0, maxiter=10, tolerance=1e-12, W_sparsity=0.01, H_sparsity=0.02)
    ref = O.cnmf(V, K, T, cfg)
    got = gpu_lib.cnmf(V, K, T, dict(cfg, nmfx_gpus=[0] * ndev))
    one = gpu_lib.cnmf(V, K, T, cfg)
    assert len(got[2]) == len(ref[2])
    e = dict(W=rel_fro(got[0], ref[0]), H=rel_fro(got[1], ref[1]), cost=rel_fro(got[2], ref[2]))
    record_err(**e)
    assert e["W"] <= 1e-5 and e["H"] <= 1e-5 and e["cost"] <= 1e-6, e
    assert rel_fro(got[0], one[0]) <= 3e-6 and rel_fro(got[1], one[1]) <= 3e-6      # vs one shard: summation order only


@pytest.mark.parametrize("div,ab", [("is", None), ("ab", (0.5, 1.5))])
@pytest.mark.parametrize("ndev,m,n,K,T", [(8, 192, 1030, 32, 8), (2, 256, 520, 64, 4), (3, 132, 1024, 64, 2), (2, 96, 200, 6, 4)])
def test_blocking_api_n_gpus_cnmf_is_and_alpha_beta(gpu_lib, div, ab, ndev, m, n, K, T):
    """IS / alpha-beta cnmf on column shards: with the fused passes (the eight pairs, m a multiple of 4) both element maps' values are also formed on a shard's T-1
    right-halo columns (as KL's R is); the last case (K = 6) runs the materialised path on its shards.  Against the oracle and the one-shard call."""
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(m, n, K, T=T)
    cfg = dict(divergence=div, W_init=W0, H_init=H0, maxiter=6, tolerance=1e-12, W_sparsity=0.01, H_sparsity=0.02)
    if ab:
        cfg["alpha"], cfg["beta"] = ab
    ref = O.cnmf(V, K, T, cfg)
    got = gpu_lib.cnmf(V, K, T, dict(cfg, nmfx_gpus=[0] * ndev))
    one = gpu_lib.cnmf(V, K, T, cfg)
    assert len(got[2]) == len(ref[2])
    assert rel_fro(got[0], ref[0]) <= 1e-5 and rel_fro(got[1], ref[1]) <= 1e-5 and rel_fro(got[2], ref[2]) <= 1e-6
    assert rel_fro(got[0], one[0]) <= 3e-6 and rel_fro(got[1], one[1]) <= 3e-6      # vs one shard: summation order only


def test_blocking_api_n_gpus_cnmf_stop_rule_and_sources(gpu_lib):
    from oracle import nmf_oracle as O
    m, n, K, T = 128, 600, 16, 4
    V, W0, H0 = synth(m, n, K, T=T)
    probe = O.cnmf(V, K, T, dict(W_init=W0, H_init=H0, maxiter=30, tolerance=1e-300))[2]
    dec = -np.diff(probe)
    assert np.all(dec[:14] > 0) and dec[9] > dec[10]
    cfg = dict(W_init=W0, H_init=H0, maxiter=30, tolerance=float(0.5 * (dec[9] + dec[10])))
    ref = O.cnmf(V, K, T, cfg)
    got = gpu_lib.cnmf(V, K, T, dict(cfg, nmfx_gpus=[0, 0, 0]))
    assert 5 < len(ref[2]) < 30 and len(got[2]) == len(ref[2])                      # cnmf.m:254-257 fires at the same iteration on 3 shards
    assert rel_fro(got[0], ref[0]) <= 1e-5 and rel_fro(got[1], ref[1]) <= 1e-5 and rel_fro(got[2], ref[2]) <= 1e-6
    Ks = [6, 10]
    cfg = dict(divergence="kl", W_init=[W0[:, :6], W0[:, 6:]], H_init=[H0[:6], H0[6:]], W_sparsity=[0.05, 0.0], H_sparsity=[0.0, 0.1],
               W_fixed=[False, True], maxiter=8, tolerance=1e-12)
    ref = O.cnmf(V, Ks, T, cfg)
    got = gpu_lib.cnmf(V, Ks, T, dict(cfg, nmfx_gpus=[0, 0]))
    assert rel_fro(np.concatenate(got[0], 1), np.concatenate(ref[0], 1)) <= 1e-5 and rel_fro(np.vstack(got[1]), np.vstack(ref[1])) <= 1e-5
    assert rel_fro(got[2], ref[2]) <= 1e-6


@pytest.mark.parametrize("sW,sH", [(0.0, 0.5), (0.4, 0.6), (0.0, 0.0), (0.3, 0.0)])
@pytest.mark.parametrize("ndev,K", [(2, 64), (3, 20), (8, 128)])
def test_blocking_api_n_gpus_nmfsc_matches_oracle(gpu_lib, sW, sH, ndev, K):
    """nmfsc.m behind the blocking call on N column shards of one process (csrc/multi_sc.hip): one host thread per shard over nmfx_nmfsc_dev, the
    all-reduce callback served by the peer reduce (fp32 for [V*H' | H*H'], fp64 for objectives and the distributed projfunc, projfunc.m:22-53).
    Identical line-search tries (nmfsc.m:152-175, 203-226) are the point: every shard must take the oracle's branches."""
    from oracle import nmf_oracle as O
    m, n = 256, 1024 + 8 * 9
    V, W0, H0 = synth(m, n, K)
    V = 2.5 * V                                                # the global max(V) rescale (nmfsc.m:62) spans the shards
    cfg = dict(W_init=W0, H_init=H0, maxiter=10, tolerance=1e-12, nmfx_path=2)
    if sW:
        cfg["W_sparsity"] = sW
    if sH:
        cfg["H_sparsity"] = sH
    i0, i1, i2 = {}, {}, {}
    W, H, cost = O.nmfsc(V, K, cfg, info=i0)
    Wg, Hg, cg = gpu_lib.nmfsc(V, K, dict(cfg, nmfx_gpus=[0] * ndev), info=i1)
    assert i1["triesH"] == i0["triesH"] and i1["triesW"] == i0["triesW"]
    assert rel_fro(Wg, W) <= 1e-5 and rel_fro(Hg, H) <= 1e-5, (rel_fro(Wg, W), rel_fro(Hg, H))
    assert len(cg) == len(cost) and rel_fro(cg, cost) <= 1e-6
    W1, H1, c1 = gpu_lib.nmfsc(V, K, cfg, info=i2)             # one shard, same kernels: summation order only
    assert i2["triesH"] == i1["triesH"] and rel_fro(Wg, W1) <= 3e-6 and rel_fro(Hg, H1) <= 3e-6
    if sH:                                                     # Hoyer postconditions hold on WHOLE rows of H (projfunc.m:3-7)
        L1s = np.sqrt(n) - (np.sqrt(n) - 1) * sH
        assert np.allclose(Hg.sum(1), L1s, rtol=1e-5) and np.allclose((Hg ** 2).sum(1), 1.0, rtol=1e-5) and Hg.min() >= 0


def test_blocking_api_n_gpus_nmfsc_stop_rule_errors_and_determinism(gpu_lib):
    from oracle import nmf_oracle as O
    m, n, K = 256, 1024, 32
    V, W0, H0 = synth(m, n, K)
    probe = O.nmfsc(V, K, dict(W_init=W0, H_init=H0, H_sparsity=0.5, maxiter=25, tolerance=1e-300))[2]
    dec = -np.diff(probe)
    j = 8
    assert np.all(dec[:j + 2] > 0) and dec[j] > dec[j + 1]
    cfg = dict(W_init=W0, H_init=H0, H_sparsity=0.5, maxiter=25, tolerance=float(0.5 * (dec[j] + dec[j + 1])), nmfx_path=2)
    ref = O.nmfsc(V, K, cfg)
    a = gpu_lib.nmfsc(V, K, dict(cfg, nmfx_gpus=4 * [0]))
    b = gpu_lib.nmfsc(V, K, dict(cfg, nmfx_gpus=4 * [0]))
    assert 3 < len(ref[2]) < 26 and len(a[2]) == len(ref[2])                        # nmfsc.m:241-244 fires at the same iteration on 4 shards
    assert rel_fro(a[0], ref[0]) <= 1e-5 and rel_fro(a[1], ref[1]) <= 1e-5 and rel_fro(a[2], ref[2]) <= 1e-6
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])   # run-to-run: bit-identical (fixed summation order)
    Vn = V.copy()
    Vn[3, 900] = -1.0
    with pytest.raises(Exception, match="Negative values in data!"):
        gpu_lib.nmfsc(Vn, K, dict(cfg, nmfx_gpus=[0, 0]))
    with pytest.raises(Exception, match="fused kernels only"):                       # a worker thread's error text reaches the caller
        gpu_lib.nmfsc(*([synth(256, 600, 300)[0]] + [300]), dict(H_sparsity=0.5, maxiter=2, nmfx_gpus=[0, 0]))
    with pytest.raises(Exception):
        gpu_lib.nmfsc(V, K, dict(cfg, nmfx_gpus=[0, 7]))                             # no such device on a 1-GPU box


@pytest.mark.parametrize("alg,div", [("cnmf", "kl"), ("cnmf", "euclidean"), ("nmf", "kl"), ("nmf", "euclidean")])
def test_blocking_api_n_gpus_one_short_shard(gpu_lib, alg, div):
    """n = 319 on 5 shards: 64, 64, 64, 64 and 63 columns -- the last one is below what the fused kernels take.  Every shard must then run the same (general)
    kernels: the packed layout and the W update's summation order depend on the path (found by scripts/fuzz_campaign_r3.py: 'shards picked different kernel
    paths')."""
    from oracle import nmf_oracle as O
    m, n, K, T = 128, 319, 32, 4
    V, W0, H0 = synth(m, n, K, T=T if alg == "cnmf" else None)
    cfg = dict(divergence=div, W_init=W0, H_init=H0, maxiter=6, tolerance=1e-12)
    if alg == "cnmf":
        ref, got = O.cnmf(V, K, T, cfg), gpu_lib.cnmf(V, K, T, dict(cfg, nmfx_gpus=[0] * 5))
    else:
        ref, got = O.nmf(V, K, cfg), gpu_lib.nmf(V, K, dict(cfg, nmfx_gpus=[0] * 5))
    assert rel_fro(got[0], ref[0]) <= 1e-5 and rel_fro(got[1], ref[1]) <= 1e-5 and rel_fro(got[2], ref[2]) <= 1e-6


@pytest.mark.parametrize("workload", ["tiny", "c4"])
def test_bench_self_launches_two_ranks(gpu_lib, workload):
    """`python3 bench.py --gpus 2` as a PLAIN command (how the driver starts the 1-GPU run): bench.py starts its own two ranks under torch.distributed.run.
    On the 1-GPU box both ranks share cuda:0 over gloo (RCCL refuses two ranks on one device); the line must say world_size_seen 2."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NMFX_BENCH_BACKEND="gloo", NMFX_BENCH_ONE_DEVICE="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", workload, "--no-cpu-baseline"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["world_size_seen"] == 2 and out["all_ranks_same_path"] and out["value"] > 0
    assert out["cost_monotone"] and len(out["per_rank_ms_per_step"]["all"]) == 2


@pytest.mark.parametrize("div,K", [("euclidean", 64), ("is", 256)])
def test_engine_without_the_transposed_copy_of_V(gpu_lib, div, K):
    """nmfx_engine_desc.flags bit 0 / Engine(no_vt=True): the euclidean fused path -- and IS above K = 192, whose H step otherwise runs as 4 + 2 m*n*K on that copy --
    without V' (what every rank falls back to TOGETHER when one workspace does not fit): smaller workspace, same results to rounding, cost vector within the contract
    of the run with the copy."""
    import torch
    from nmf_toolbox_amd.engine import Engine, colmajor_to_torch
    m, n = 512, 2048
    V, W0, H0 = synth(m, n, K)
    out = []
    for no_vt in (False, True):
        e = Engine(colmajor_to_torch(V, "cuda:0"), colmajor_to_torch(W0, "cuda:0"), colmajor_to_torch(H0, "cuda:0"), divergence=div, use_dist=False, no_vt=no_vt)
        assert e.path_kind == 1 and (int(e.desc.flags) & 1) == (1 if no_vt else 0)
        e.init()
        cost = torch.zeros(6, dtype=torch.float64, device="cuda:0")
        e.iterate(6, cost)
        torch.cuda.synchronize()
        out.append((e.W.double().cpu().numpy(), e.H.double().cpu().numpy(), cost.cpu().numpy(), e.workspace.numel()))
        e.close()
    assert out[1][3] < out[0][3] - 4 * m * n + 4096                     # the copy (m*n floats) is what the flag gives back
    assert rel_fro(out[1][0], out[0][0]) < 2e-6 and rel_fro(out[1][1], out[0][1]) < 2e-6 and rel_fro(out[1][2], out[0][2]) < 1e-7


def _background_load(stop_after_s, started=None, stop=None):
    """a second process keeping the GPU busy with its own factorisations (different kernels, different timing) until told to stop (or for stop_after_s at most)"""
    import time
    import nmf_toolbox_amd as A
    V, W0, H0 = synth(512, 2048, 96)
    t0 = time.time()
    while time.time() - t0 < stop_after_s and not (stop is not None and stop.is_set()):
        if started is not None:
            started.set()
        A.nmf(V, 96, dict(divergence="kl", W_init=W0, H_init=H0, maxiter=20, tolerance=1e-300))
        A.nmf(V, 96, dict(divergence="euclidean", W_init=W0, H_init=H0, maxiter=20, tolerance=1e-300))


def test_run_to_run_determinism_under_concurrent_load(gpu_lib):
    """Same inputs => bit-identical outputs, also while ANOTHER process shares the GPU and shifts every timing: three shards on one device (ragged 427 / 427 / 426
    columns) for the paths whose tiles are filled by LDS-DMA behind counted waits -- KL, euclidean (Gram-form cost), IS with the dual-map kernel (K = 96) and as two
    single-map passes (K = 256), KL cnmf with halos.  This is the condition under which round 4 found the functor-12 / 14 race (a tile read before its DMA rows had
    landed: right most of the time when nothing else ran)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    started, stop = ctx.Event(), ctx.Event()
    bg = ctx.Process(target=_background_load, args=(120.0, started, stop))
    bg.start()
    try:
        assert started.wait(timeout=300)                     # the load is on the GPU (its process has imported the library and entered the loop)
        m, n = 384, 1280
        cases = [("nmf", "kl", 256, 1), ("nmf", "euclidean", 128, 1), ("nmf", "is", 96, 1), ("nmf", "is", 256, 1), ("cnmf", "kl", 64, 4)]
        for alg, div, K, T in cases:
            V, W0, H0 = synth(m, n, K, T=T if alg == "cnmf" else None)
            cfg = dict(divergence=div, W_init=W0, H_init=H0, maxiter=6, tolerance=1e-300, W_sparsity=0.01, nmfx_gpus=[0, 0, 0])
            run = (lambda: gpu_lib.cnmf(V, K, T, cfg)) if alg == "cnmf" else (lambda: gpu_lib.nmf(V, K, cfg))
            first = run()
            for rep in range(7):
                again = run()
                assert np.array_equal(again[0], first[0]) and np.array_equal(again[1], first[1]) and np.array_equal(again[2], first[2]), (alg, div, K, rep)
    finally:
        stop.set()
        bg.join(timeout=120)
        if bg.is_alive():
            bg.terminate()
    assert bg.exitcode == 0


# ---- the RCCL backend of the blocking multi-GPU call (rccl_backend.hip; north_star: ONE RCCL all-reduce of the packed W-step sums per iteration) ----------------
def test_rccl_library_loads(gpu_lib):
    import ctypes as C
    from nmf_toolbox_amd import _lib
    v = C.c_int32(0)
    path = _lib.load().nmfx_rccl_library(C.byref(v))
    assert path and b"rccl" in path and v.value > 0, (path, v.value)


@pytest.mark.parametrize("alg,div,m,n,K,T", [("nmf", "kl", 256, 1024, 64, 1), ("nmf", "euclidean", 256, 1024, 64, 1), ("nmf", "is", 200, 600, 40, 1), ("nmf", "euclidean", 130, 700, 300, 1),
                                             ("cnmf", "euclidean", 128, 512, 32, 4), ("cnmf", "kl", 128, 512, 32, 4), ("lnmf", "kl", 256, 1024, 64, 1)])
def test_blocking_api_rccl_backend_one_shard(gpu_lib, alg, div, m, n, K, T):
    """the 1-GPU box runs the RCCL branch of the sharded driver with ONE shard (nmfx_gpus = [0], nmfx_multi_backend = "rccl": ncclCommInitAll over one device,
    one ncclAllReduce per iteration on the engine's stream): bit-identical to the peer branch and to the unsharded call, the exchange timed and named"""
    import ctypes as C
    from oracle import nmf_oracle as O
    from nmf_toolbox_amd import _lib
    V, W0, H0 = synth(m, n, K, T=(T if alg == "cnmf" else None))
    if alg == "lnmf":
        cfg = dict(W_init=W0 / W0.sum(0), H_init=H0, maxiter=6, tolerance=1e-300)
        run = lambda c: gpu_lib.lnmf(V, K, c)
        ref = O.lnmf(V, K, cfg)
    elif alg == "cnmf":
        cfg = dict(divergence=div, W_init=W0, H_init=H0, maxiter=6, tolerance=1e-300)
        run = lambda c: gpu_lib.cnmf(V, K, T, c)
        ref = O.cnmf(V, K, T, cfg)
    else:
        cfg = dict(divergence=div, W_init=W0, H_init=H0, maxiter=6, tolerance=1e-300, W_sparsity=0.01)
        run = lambda c: gpu_lib.nmf(V, K, c)
        ref = O.nmf(V, K, cfg)
    out = {}
    for be in ("rccl", "peer"):
        out[be] = run(dict(cfg, nmfx_gpus=[0], nmfx_multi_backend=be))
        ms, cnt, used = C.c_double(0), C.c_int32(0), C.c_int32(0)
        _lib.check(_lib.load().nmfx_last_call_exchange(C.byref(ms), C.byref(cnt), C.byref(used)))
        assert used.value == (2 if be == "rccl" else 1) and cnt.value == 6 and ms.value > 0
    plain = run(cfg)
    for a, b, c in zip(out["rccl"], out["peer"], plain):
        assert np.array_equal(np.asarray(a), np.asarray(b)) and np.array_equal(np.asarray(a), np.asarray(c))
    assert rel_fro(out["rccl"][0], ref[0]) <= 1e-5 and rel_fro(out["rccl"][1], ref[1]) <= 1e-5 and rel_fro(out["rccl"][2], ref[2]) <= (1e-5 if div == "is" else 1e-6)


def test_blocking_api_rccl_backend_refuses_duplicate_devices(gpu_lib):
    """RCCL takes distinct GPUs: asked for by name on [0, 0] it is an error with the reason; "auto" falls back to the peer exchange"""
    import ctypes as C
    from nmf_toolbox_amd import _lib
    V, W0, H0 = synth(128, 512, 32)
    cfg = dict(divergence="kl", W_init=W0, H_init=H0, maxiter=3, tolerance=1e-300, nmfx_gpus=[0, 0])
    with pytest.raises(_lib.NmfxError) as ei:
        gpu_lib.nmf(V, 32, dict(cfg, nmfx_multi_backend="rccl"))
    assert ei.value.status == _lib.NMFX_ERR_UNSUPPORTED and "more than once" in str(ei.value)
    gpu_lib.nmf(V, 32, cfg)
    used = C.c_int32(0)
    _lib.check(_lib.load().nmfx_last_call_exchange(None, None, C.byref(used)))
    assert used.value == 1


def test_blocking_multi_gpu_call_reports_its_own_timing(gpu_lib):
    """run_mu_multi left nmfx_last_call_timing at whatever an EARLIER call on the thread had written (a warm-up's 4 ms for ten C3 iterations).
    The three spans now belong to the call: a long sharded call after a short unsharded one reports more iterate time than the short call took in total, at
    least the time its fused passes need (2 x 2*m*n*K flop per iteration at the fp32 MFMA peak is a floor no run can beat), and they add up to no more than the
    wall time of the call"""
    import time
    from nmf_toolbox_amd import _lib
    m, n, K, iters = 2048, 8192, 256, 20
    V, W0, H0 = synth(m, n, K)
    small = synth(128, 256, 32)
    gpu_lib.nmf(small[0], 32, dict(divergence="kl", W_init=small[1], H_init=small[2], maxiter=2, tolerance=1e-300))   # what used to be reported afterwards
    short = _lib.last_call_timing()
    cfg = dict(divergence="kl", W_init=W0, H_init=H0, maxiter=iters, tolerance=1e-300, nmfx_gpus=[0, 0])
    t0 = time.perf_counter()
    gpu_lib.nmf(V, K, cfg)
    wall = time.perf_counter() - t0
    tm = _lib.last_call_timing()
    floor = iters * 8.0 * m * n * K / 157.3e12        # nmf.m:152-153,183-184 on the fused passes: 8*m*n*K flop per iteration
    assert tm["iterate_s"] >= floor, (tm, floor)
    assert tm["iterate_s"] > short["iterate_s"] and tm["ingest_s"] > 0 and tm["egress_s"] > 0, (tm, short)
    assert tm["ingest_s"] + tm["iterate_s"] + tm["egress_s"] <= wall * 1.001, (tm, wall)
    assert tm["host_bytes_in"] >= 8.0 * (m * n + m * K + K * n), tm

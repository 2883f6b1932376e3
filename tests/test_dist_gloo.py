"""CPU suite: the N > 1 iteration loop (nmf_toolbox_amd.engine.run_sharded_iterations) under torch.distributed/gloo,
world_size 2.  The phases are computed by a float64 NumPy stand-in that follows the engine's packed all-reduce layouts
([N | P] generic, [N | rowsum(H)] fused KL, [N | H*H'] fused euclidean); the test proves that column-sharding V/H,
all-reducing only the packed W-step sums and finishing redundantly on every rank reproduces the unsharded oracle, with
bit-identical W on all ranks, for both cost schedules (in-step and lagged)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import EPS, synth


class NumpyShard:
    """one rank's phases in float64 (mirrors nmf_toolbox_amd/csrc/api.hip's engine, same packed layouts)"""

    def __init__(self, V, W, H, div, layout, lamW, lamH, fixW, fixH, rank0, n_chunks=1):
        self.n_chunks = n_chunks                           # > 1: row-chunked W-step partial, packed = [chunk 0 | chunk 1 | ... | tail]
        self.V, self.W, self.H, self.div, self.layout = V, W.copy(), H.copy(), div, layout
        self.m, self.n = V.shape
        self.K = W.shape[1]
        self.lamW, self.lamH, self.fixW, self.fixH, self.rank0 = lamW, lamH, fixW.astype(bool), fixH.astype(bool), rank0
        # where the cost of iteration i turns up (nmfx_engine_cost_lag): 0 after hstep(i), 1 after wstep_partial(i+1), 2 after wstep_finish(i+1)
        # ("gram": the euclidean fused path's cost in Gram form, out of the column sums of the W update)
        self.cost_lag = 2 if layout == "gram" else (1 if layout == "fused" else 0)
        self.cost_lags = self.cost_lag != 0
        mk = self.m * self.K
        tail = mk if layout == "generic" and div != "kl" else (self.K * self.K if (layout in ("fused", "gram") and div == "euclidean") else self.K)
        self.packed = torch.zeros(mk + tail, dtype=torch.float64)
        self.cost_local = 0.0
        self.latch_at = None      # "gram" layout: from W step number latch_at on the engine is back on the one-pass kernel -- the cost of the previous iteration then
        self.wsteps = 0           # comes out of wstep_partial (point 1) while cost_lag keeps saying 2 (include/nmfx.h, nmfx_engine_cost_lag)
        self.W *= 1.0 / np.sqrt((self.W ** 2).sum(0))[None, :]                       # nmf.m:130-134

    def _cost(self):
        S = self.W @ self.H
        c = 0.5 * np.sum((self.V - S) ** 2) if self.div == "euclidean" else np.sum(self.V * np.log(self.V / S) - self.V + S)
        c += float(np.sum(self.lamH * np.abs(self.H).sum(1)))
        if self.rank0:
            c += float(np.sum(self.lamW * np.abs(self.W).sum(0)))
        self.cost_local = c

    def wstep_partial(self):
        mk = self.m * self.K
        S = self.W @ self.H
        A = self.V / S if self.div == "kl" else self.V
        N = A @ self.H.T
        self.classic = self.cost_lag == 2 and self.latch_at is not None and self.wsteps >= self.latch_at
        self.wsteps += 1
        if self.cost_lag == 1 or self.classic:
            self._cost()
        p = self.packed.numpy()
        p[:mk] = N.ravel(order="F")
        if self.div == "kl":
            p[mk:] = self.H.sum(1)
        elif self.layout in ("fused", "gram"):
            p[mk:] = (self.H @ self.H.T).ravel(order="F")
        else:
            p[mk:] = (S @ self.H.T).ravel(order="F")

    def wstep_partial_chunk(self, c, nch):
        """rows [c*m/nch, (c+1)*m/nch) of N as a contiguous (m/nch x K) block; tail and lagged cost with the last chunk"""
        cr = self.m // nch
        if c == 0:
            S = self.W @ self.H
            A = self.V / S if self.div == "kl" else self.V
            self._N = A @ self.H.T
        p = self.packed.numpy()
        p[c * cr * self.K:(c + 1) * cr * self.K] = self._N[c * cr:(c + 1) * cr].ravel(order="F")
        if c == nch - 1:
            mk = self.m * self.K
            if self.cost_lags:
                self._cost()
            p[mk:] = self.H.sum(1) if self.div == "kl" else (self.H @ self.H.T).ravel(order="F")

    def packed_chunk(self, c, nch):
        cr = self.m // nch
        hi = (c + 1) * cr * self.K if c < nch - 1 else self.packed.numel()
        return self.packed[c * cr * self.K:hi]

    def wstep_finish(self):
        mk = self.m * self.K
        p = self.packed.numpy()
        if self.n_chunks > 1:
            cr = self.m // self.n_chunks
            N = np.concatenate([p[c * cr * self.K:(c + 1) * cr * self.K].reshape(cr, self.K, order="F") for c in range(self.n_chunks)], axis=0)
        else:
            N = p[:mk].reshape(self.m, self.K, order="F")
        if self.div == "kl":
            P = np.broadcast_to(p[mk:][None, :], N.shape)
        elif self.layout in ("fused", "gram"):
            P = self.W @ p[mk:].reshape(self.K, self.K, order="F")
        else:
            P = p[mk:].reshape(self.m, self.K, order="F")
        W = self.W
        dn, dp = (W * P).sum(0), (W * N).sum(0)
        if self.cost_lag == 2 and not getattr(self, "classic", False):
            # 0.5*||V - W*H||^2 = 0.5*||V||^2 - <W, V*H'> + 0.5*<W, W*(H*H')> of the state this step started from: the all-reduced sums make the
            # cross terms global, so rank 0 alone carries them; every rank adds its own 0.5*||V_local||^2 and lambda_H*|H_local|
            c = 0.5 * np.sum(self.V ** 2) + float(np.sum(self.lamH * np.abs(self.H).sum(1)))
            if self.rank0:
                c += 0.5 * dn.sum() - dp.sum() + float(np.sum(self.lamW * np.abs(W).sum(0)))
            self.cost_local = c
        Wn = W * ((N + W * dn) / np.fmax(P + W * dp + self.lamW[None, :], EPS))
        Wn *= 1.0 / np.sqrt((Wn ** 2).sum(0))[None, :]
        self.W = np.where(self.fixW[None, :], W, Wn)

    def hstep(self):
        S = self.W @ self.H
        if self.div == "kl":
            neg, pos = self.W.T @ (self.V / S), np.broadcast_to(self.W.sum(0)[:, None], self.H.shape)
        else:
            neg, pos = self.W.T @ self.V, self.W.T @ S
        Hn = self.H * (neg / np.fmax(pos + self.lamH[:, None], EPS))
        self.H = np.where(self.fixH[:, None], self.H, Hn)
        if not self.cost_lags:
            self._cost()

    def cost_pass(self):
        self._cost()

    def backup_W(self):
        self._Wbak = self.W.copy()

    def restore_W(self):
        self.W = self._Wbak.copy()

    def _copy_cost(self, dst):
        dst[0] = self.cost_local


class NumpyShardMerged(NumpyShard):
    """the same phases behind the one-call-per-iteration entry point (nmfx_engine_between_allreduces)"""

    def between_allreduces(self, last, lag2_cost_dst=None):
        self.wstep_finish()
        if lag2_cost_dst is not None:
            self._copy_cost(lag2_cost_dst)
        self.hstep()
        if not last:
            if not self.cost_lags:
                saved = self.cost_local          # the generic path's cost belongs to the H step just done, not to the next partial
            self.wstep_partial()
            if not self.cost_lags:
                self.cost_local = saved


def _worker(rank, world, port, div, layout, iters, q, n_chunks=1, tolerance=None, latch_at=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nmf_toolbox_amd.engine import run_sharded_iterations, shard_columns
    m, n, K = 48, 90, 6
    V, W0, H0 = synth(m, n, K)
    lamW = np.array([0.05] * 2 + [0.0] * 4)
    lamH = np.array([0.0] * 2 + [0.1] * 4)
    fixW = np.array([0, 0, 0, 0, 1, 1])
    fixH = np.array([1, 1, 0, 0, 0, 0])
    lo, hi = shard_columns(n, world, rank)
    cls = NumpyShardMerged if n_chunks == 0 else NumpyShard          # n_chunks == 0 selects the merged-call loop
    n_chunks = max(n_chunks, 1)
    be = cls(V[:, lo:hi], W0, H0[:, lo:hi], div, layout, lamW, lamH, fixW, fixH, rank == 0, n_chunks)
    be.latch_at = latch_at
    cost = torch.zeros(iters, dtype=torch.float64)
    ran = run_sharded_iterations(be, iters, dist, None, cost, tolerance)
    q.put((rank, lo, hi, be.W, be.H, cost.numpy()[:ran].copy()))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cases():
    for div in ("euclidean", "kl"):
        for layout, n_chunks in (("generic", 1), ("fused", 1), ("fused", 3), ("generic", 0), ("fused", 0)):
            yield div, layout, n_chunks, None
    yield "euclidean", "gram", 1, None          # Gram-form cost: lagged, delivered by wstep_finish
    yield "euclidean", "gram", 0, None
    for div, layout in (("euclidean", "generic"), ("euclidean", "fused"), ("kl", "fused"), ("euclidean", "gram")):
        yield div, layout, 1, 0.05              # the stop rule of nmf.m:221-224 inside the sharded loop
    # a lag-2 engine that goes back to the one-pass kernel at W step 5 (the Gram-form cost's switch): from there on the cost turns up one phase earlier, and the
    # merged loop's next wstep_partial would overwrite it before the round-3 loop read it (advisor, round 3) -- both loops, and with the stop rule
    yield "euclidean", "gram", 0, None, 5
    yield "euclidean", "gram", 1, None, 5
    yield "euclidean", "gram", 1, 0.05, 5


@pytest.mark.parametrize("case", list(_cases()))
def test_sharded_loop_matches_unsharded_oracle(case):
    div, layout, n_chunks, tolerance = case[:4]
    latch_at = case[4] if len(case) > 4 else None
    from oracle import nmf_oracle as O
    world, iters = 2, 12 if tolerance is None else 40
    if tolerance is not None:       # a tolerance that makes nmf.m:221 fire around iteration 10 of this problem: between two consecutive decreases of the cost
        Vt, Wt, Ht = synth(48, 90, 6)
        ct = O.nmf(Vt, [2, 2, 2], dict(divergence=div, W_init=[Wt[:, :2], Wt[:, 2:4], Wt[:, 4:]], H_init=[Ht[:2], Ht[2:4], Ht[4:]], W_sparsity=[0.05, 0.0, 0.0],
                                       H_sparsity=[0.0, 0.1, 0.1], W_fixed=[False, False, True], H_fixed=[True, False, False], maxiter=iters, tolerance=1e-300))[2]
        dec = -np.diff(ct)
        assert np.all(dec[:12] > 0) and dec[9] > dec[10]
        tolerance = 0.5 * (dec[9] + dec[10])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, div, layout, iters, q, n_chunks, tolerance, latch_at)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    m, n, K = 48, 90, 6
    V, W0, H0 = synth(m, n, K)
    cfg = dict(divergence=div, W_init=[W0[:, :2], W0[:, 2:4], W0[:, 4:]], H_init=[H0[:2], H0[2:4], H0[4:]], W_sparsity=[0.05, 0.0, 0.0],
               H_sparsity=[0.0, 0.1, 0.1], W_fixed=[False, False, True], H_fixed=[True, False, False], maxiter=iters, tolerance=1e-300 if tolerance is None else tolerance)
    W, H, cost = O.nmf(V, [2, 2, 2], cfg)
    if tolerance is not None:
        assert 2 < len(cost) < iters                                # the rule really fired, and not at once
        assert all(len(r[5]) == len(cost) for r in res)             # ... at the same iteration on every rank as in the unsharded oracle
    W, H = np.hstack(W), np.vstack(H)
    assert np.array_equal(res[0][3], res[1][3])                     # W bit-identical on both ranks
    Hs = np.concatenate([r[4] for r in res], axis=1)
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    assert rel(res[0][3], W) < 1e-11 and rel(Hs, H) < 1e-11
    for r in res:
        assert rel(r[5], cost) < 1e-12                              # every rank holds the global cost vector


# ---- the distributed projfunc protocol of nmfsc on column shards ----------------------------------
def _projfunc_phases(S_local, N, k1, k2, allreduce):
    """float64 NumPy mirror of nmf_toolbox_amd/csrc/projfunc.hip::projfunc_cols_dist: K vectors at once, each split over the
    ranks; `allreduce(array)` sums a (K, 4) array over ranks in place.  Phases: init -> [shift+sums -> step -> zero]*."""
    K = S_local.shape[0]
    v = S_local.astype(np.float64).copy()
    Z = np.zeros_like(v, dtype=bool)
    done = np.zeros(K, dtype=bool)
    nz = np.zeros(K)
    red = np.zeros((K, 4))
    red[:, 0] = v.sum(1)
    allreduce(red)
    iters = np.ones(K, dtype=int)
    while True:
        tot, cnt = red[:, 0].copy(), red[:, 1].copy()
        red[:] = 0
        act = ~done
        nz[act] = cnt[act]
        shift, mid = (k1 - tot) / (N - nz), k1 / (N - nz)
        for k in np.nonzero(act)[0]:
            v[k, ~Z[k]] += shift[k]
            w = np.where(Z[k], 0.0, v[k] - mid[k])
            red[k, :3] = [(w * w).sum(), (w * v[k]).sum(), (v[k] * v[k]).sum()]
        allreduce(red)
        a, b, c = red[:, 0].copy(), 2 * red[:, 1], red[:, 2] - k2
        red[:] = 0
        for k in np.nonzero(act)[0]:
            disc = b[k] * b[k] - 4 * a[k] * c[k]
            alphap = (-b[k] + (np.sqrt(disc) if disc > 0 else 0.0)) / (2 * a[k])
            w = np.where(Z[k], 0.0, v[k] - mid[k])
            v[k] = alphap * w + v[k]
            red[k, 0] = np.count_nonzero(~(v[k] >= 0))
        allreduce(red)
        if np.all(red[:, 0] == 0):
            return v, iters
        neg = red[:, 0].copy()
        red[:] = 0
        done |= neg == 0
        for k in np.nonzero(~done)[0]:
            iters[k] += 1
            Z[k] = v[k] <= 0
            v[k, Z[k]] = 0.0
            red[k, :2] = [v[k].sum(), np.count_nonzero(Z[k])]
        allreduce(red)


def _pf_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nmf_toolbox_amd.engine import shard_columns
    rs = np.random.RandomState(5)
    S = np.abs(rs.randn(7, 301)) + 1e-3
    S[3] = np.abs(rs.randn(301)) ** 4                                  # a peaky row: more zeroing rounds than the others
    lo, hi = shard_columns(301, world, rank)
    N = 301
    k1 = np.sqrt(N) - (np.sqrt(N) - 1) * 0.7

    def allreduce(a):
        t = torch.from_numpy(a)
        dist.all_reduce(t)

    v, iters = _projfunc_phases(S[:, lo:hi], N, k1, 1.0, allreduce)
    q.put((rank, v, iters))
    dist.barrier()
    dist.destroy_process_group()


def test_distributed_projfunc_protocol_matches_projfunc_m():
    from oracle import nmf_oracle as O
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pf_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rs = np.random.RandomState(5)
    S = np.abs(rs.randn(7, 301)) + 1e-3
    S[3] = np.abs(rs.randn(301)) ** 4
    N = 301
    k1 = np.sqrt(N) - (np.sqrt(N) - 1) * 0.7
    V = np.concatenate([r[1] for r in res], axis=1)
    assert np.array_equal(res[0][2], res[1][2])                           # same branch sequence on both ranks
    its = []
    for k in range(7):
        v, it = O.projfunc(S[k], k1, 1.0, True)
        its.append(it)
        assert np.allclose(V[k], v, rtol=1e-12, atol=1e-14) and res[0][2][k] == it
    assert len(set(its)) > 1                                              # rows really finished at different rounds


def _ws_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nmf_toolbox_amd import _lib
    from nmf_toolbox_amd.engine import agree_on_workspace

    def any_rank(failed):
        t = torch.tensor([1.0 if failed else 0.0], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return bool(t.item())

    def run(capacity, flags0=0, err=None):
        asked, released = [], []

        def alloc(nb):
            asked.append(nb)
            if err is not None and rank == 1:
                raise err
            if nb > capacity[rank]:
                raise RuntimeError("HIP out of memory. Tried to allocate %d bytes" % nb)
            return bytearray(8)

        try:
            buf, flags = agree_on_workspace(flags0[rank] if isinstance(flags0, list) else flags0, lambda f: 600 if f & 1 else 1000, alloc, any_rank, lambda: released.append(1))
            return ("ok", flags, asked, len(released), buf is not None)
        except _lib.NmfxError as ex:
            return ("nomem", ex.status, asked, len(released), False)
        except RuntimeError as ex:
            return ("raised", str(ex), asked, len(released), False)

    out = [run([2000, 2000]),                 # fits everywhere: flags untouched, one attempt
           run([2000, 700]),                  # rank 1 is short: BOTH ranks drop the transposed copy and retry
           run([2000, 500]),                  # rank 1 cannot hold it either way: NOMEM on both ranks
           run([2000, 700], flags0=1),        # already without the copy and it fits
           run([2000, 500], flags0=1),        # already without the copy: no second attempt, NOMEM on both
           run([2000, 2000], flags0=[0, 1]),  # the flags differ at entry (NMFX_NO_VT on rank 1 only): both run without the copy
           run([2000, 500], flags0=[1, 0])]   # ... and a failure then is NOMEM on both ranks together (no rank is left waiting in a second all-reduce)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_workspace_retry_is_collective():
    """a rank that cannot hold the transposed copy of V must take every other rank with it -- the kernel path and the summation order of the
    replicated W update follow from the descriptor flags.  Engine's allocation loop (engine.py::agree_on_workspace) with a fake allocator on two gloo ranks."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ws_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from nmf_toolbox_amd import _lib
    for rank in (0, 1):
        fits, short, never, pre, pre_never, mixed, mixed_never = res[rank]
        assert mixed == ("ok", 1, [600], 0, True)
        assert mixed_never[:2] == ("nomem", _lib.NMFX_ERR_NOMEM) and mixed_never[2] == [600]
        assert fits == ("ok", 0, [1000], 0, True)
        assert short == ("ok", 1, [1000, 600], 1, True)                 # same flags, same two attempts on the rank that had room as on the one that had not
        assert never[:2] == ("nomem", _lib.NMFX_ERR_NOMEM) and never[2] == [1000, 600]
        assert pre == ("ok", 1, [600], 0, True)
        assert pre_never[:2] == ("nomem", _lib.NMFX_ERR_NOMEM) and pre_never[2] == [600]


def test_workspace_retry_single_process_and_foreign_errors():
    from nmf_toolbox_amd import _lib
    from nmf_toolbox_amd.engine import agree_on_workspace

    def alloc_limit(limit):
        def alloc(nb):
            if nb > limit:
                raise MemoryError()
            return nb
        return alloc

    same = lambda failed: failed
    assert agree_on_workspace(0, lambda f: 600 if f & 1 else 1000, alloc_limit(2000), same) == (1000, 0)
    assert agree_on_workspace(0, lambda f: 600 if f & 1 else 1000, alloc_limit(700), same) == (600, 1)
    with pytest.raises(_lib.NmfxError) as ei:
        agree_on_workspace(0, lambda f: 600 if f & 1 else 1000, alloc_limit(100), same)
    assert ei.value.status == _lib.NMFX_ERR_NOMEM

    def broken(nb):
        raise RuntimeError("invalid device ordinal")

    with pytest.raises(RuntimeError, match="invalid device ordinal"):      # not an out-of-memory condition: never swallowed, never retried
        agree_on_workspace(0, lambda f: 1000, broken, same)

"""Device-resident engine: the phase API of include/nmfx.h on buffers that already live in HBM.

PyTorch is plumbing here -- it owns the device allocations, the HIP stream and (for N > 1)
`torch.distributed` over RCCL; every numeric step is a libnmfx kernel.  Column-major (MATLAB) buffers
are carried as torch tensors of the *reversed* shape, i.e. V (m x n, column-major) is a contiguous
torch tensor of shape (n, m).

Multi-GPU: V and H are column-sharded, W is replicated; per iteration ONE all-reduce
of the packed W-step partials [N | P] (KL: [N | rowsum(H)]).  Everything after the all-reduce is
computed redundantly and identically on every rank, so W stays bit-identical across ranks.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib

DIV_CODES = {"euclidean": _lib.DIV_EUCLIDEAN, "kl": _lib.DIV_KL, "kl_divergence": _lib.DIV_KL, "is": _lib.DIV_IS,
             "is_divergence": _lib.DIV_IS, "ab": _lib.DIV_AB, "ab_divergence": _lib.DIV_AB, "frobenius": _lib.DIV_EUCLIDEAN_NOCOST}


def colmajor_to_torch(a, device):
    """NumPy array (MATLAB shape) -> fp32 torch tensor holding its column-major image (reversed shape)."""
    import torch
    a = np.asarray(a)
    t = torch.from_numpy(np.ascontiguousarray(a.transpose(), dtype=np.float32))
    return t.to(device)


def torch_to_colmajor(t):
    """inverse of colmajor_to_torch -> float64 NumPy array of MATLAB shape"""
    return np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float64).transpose())


def shard_columns(n, world, rank):
    """contiguous column block of rank `rank` (first n % world ranks get one extra column)"""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def run_sharded_iterations(backend, iters, dist, group=None, cost_out=None, tolerance=None):
    """The N > 1 iteration loop, independent of what computes the phases.  Returns the number of iterations run.

    `backend` provides wstep_partial(), wstep_finish(), hstep(), the tensor `packed` (this rank's W-step sums, in place
    all-reducible) and _copy_cost(dst) (this rank's cost partial into a 1-element fp64 tensor).  One all-reduce of
    `packed` per iteration is the only data-path collective; a backend with n_chunks > 1 computes `packed` in row chunks
    (wstep_partial_chunk / packed_chunk) and the same bytes travel as n_chunks pipelined all-reduces.  nmf.m returns the cost
    vector (nmf.m:206-218): without a stop rule the ranks' partials are collected per iteration and summed over ranks ONCE at
    the end (8*iters bytes).  With `tolerance` > 0 the stop rule of nmf.m:221-224 is applied: the cost of every iteration is
    summed over the ranks as soon as it exists (one 8-byte all-reduce) and read by the host; the loop stops with W and H at
    the state of the iteration the rule fired on, like the reference.
    Where the cost of iteration i turns up is the backend's `cost_lag`: 0 after hstep(i); 1 after wstep_partial(i+1) (fused KL
    passes); 2 after wstep_finish(i+1) (Gram-form cost of the euclidean fused path: W has moved by then, so the loop keeps the
    previous W -- backend.backup_W() / restore_W() -- to hand back on a stop).  The last one needs backend.cost_pass().
    """
    lag = bool(getattr(backend, "cost_lags", False))
    lagk = int(getattr(backend, "cost_lag", 1 if lag else 0))   # 2: the lagged cost comes out of wstep_finish (Gram-form cost), not of wstep_partial
    stop_on = tolerance is not None and tolerance > 0
    stop_le = bool(getattr(backend, "stop_le", False))          # lnmf.m:84 compares with <=
    if stop_on and cost_out is None:
        raise ValueError("the stop rule needs cost_out")
    reduced = 0                                                   # cost_out[:reduced] already hold GLOBAL costs

    def emit(idx):
        backend._copy_cost(cost_out[idx:idx + 1])

    def fired(idx):
        """global cost(idx) now; True when nmf.m:221 says stop"""
        nonlocal reduced
        if dist is not None:
            dist.all_reduce(cost_out[idx:idx + 1], group=group)
        reduced = idx + 1
        if idx == 0:
            return False
        c, prev = float(cost_out[idx]), float(cost_out[idx - 1])
        return (c <= prev and prev - c <= tolerance) if stop_le else (c < prev and prev - c < tolerance)

    def allreduce_packed():
        ev = getattr(backend, "comm_events", None)                # measurement hook: how long the compute stream stalls on the exchange
        if ev is not None:
            a, b = backend.torch.cuda.Event(enable_timing=True), backend.torch.cuda.Event(enable_timing=True)
            a.record()
        if dist is not None:
            dist.all_reduce(backend.packed, group=group)          # the ONE exchange step of an iteration
        if ev is not None:
            b.record()
            ev.append((a, b))

    nch = int(getattr(backend, "n_chunks", 1))
    merged = nch == 1 and hasattr(backend, "between_allreduces") and not getattr(backend, "has_halos", False) and not stop_on
    ran, stopped = iters, False
    for it in range(iters):
        if merged:
            # one host call per iteration next to the collective: [wstep_finish, hstep, next wstep_partial] is a single C entry point
            if it == 0:
                backend.wstep_partial()
            if lagk == 1 and it > 0 and cost_out is not None:
                emit(it - 1)
            allreduce_packed()
            # lag 2: cost(it-1) is complete right after the wstep_finish inside this call and the wstep_partial that follows it there may overwrite it (an
            # engine that has gone back to the one-pass kernel produces cost(it) in that pass): the copy into cost_out happens inside the call, between the two
            backend.between_allreduces(it == iters - 1, cost_out[it - 1:it] if (lagk == 2 and it > 0 and cost_out is not None) else None)
            if not lag and cost_out is not None:
                emit(it)
            continue
        if nch > 1:
            # row-chunked W step: the all-reduce of chunk c (async, on the collective's own stream) overlaps the compute of
            # chunk c+1; the lagged cost is complete after the last chunk
            works = []
            for c in range(nch):
                backend.wstep_partial_chunk(c, nch)
                works.append(dist.all_reduce(backend.packed_chunk(c, nch), group=group, async_op=True))
            for w in works:
                w.wait()
            if lag and it > 0 and cost_out is not None:
                emit(it - 1)                                  # (row chunks carry their cost inside the pass: always available here)
                if stop_on and fired(it - 1):                 # W and H are untouched until wstep_finish: the state of iteration it-1
                    ran, stopped = it, True
                    break
        else:
            backend.wstep_partial()
            if lagk == 1 and it > 0 and cost_out is not None:
                emit(it - 1)
                if stop_on and fired(it - 1):
                    ran, stopped = it, True
                    break
            allreduce_packed()
        keep_w = stop_on and lagk == 2 and nch == 1 and it > 0
        if keep_w:
            backend.backup_W()
        backend.wstep_finish()
        if lagk == 2 and nch == 1 and it > 0 and cost_out is not None:
            emit(it - 1)
            if stop_on and fired(it - 1):
                backend.restore_W()                           # the update that produced cost(it-1) has already moved W: hand back the one before it
                ran, stopped = it, True
                break
        backend.hstep()
        if getattr(backend, "has_halos", False):
            backend.exchange_halos()                          # cnmf only: T-1 columns of H to each neighbour ...
            backend.hstep_finish()                            # ... then V_hat / cost with the new H
        if not lag and cost_out is not None:
            emit(it)
            if stop_on and fired(it):
                ran, stopped = it + 1, True
                break
    if lag and iters > 0 and cost_out is not None and not stopped:
        backend.cost_pass()
        emit(iters - 1)
    if ran > reduced and cost_out is not None and dist is not None:
        dist.all_reduce(cost_out[reduced:ran], group=group)   # local partials -> global costs, one small collective
    return ran


def agree_on_workspace(flags, size_of, alloc, any_rank, release=lambda: None):
    """Allocate the engine workspace so that EVERY rank ends with the same descriptor flags.  size_of(flags) -> bytes, alloc(bytes) -> buffer (raises an
    out-of-memory error when it does not fit), any_rank(failed) -> True when the allocation failed on at least one rank (a MAX all-reduce; identity without a
    process group).  Bit 0 set on ANY rank at entry is set on all of them first.  Then try with those flags; if any rank fails, all ranks drop theirs and retry without the transposed copy of V (flags bit 0).  A second
    failure anywhere, or a first one with bit 0 already set, is NMFX_ERR_NOMEM on all ranks.  Errors other than out-of-memory propagate.  -> (buffer, flags)"""
    if any_rank(bool(flags & 1)):        # flags that differ at the start (no_vt / NMFX_NO_VT on one rank only) are reconciled first: one rank without the copy -> all without
        flags |= 1
    for attempt in range(2):
        nbytes = size_of(flags)
        buf, failed = None, False
        try:
            buf = alloc(nbytes)
        except (MemoryError, RuntimeError) as ex:            # torch.OutOfMemoryError is a RuntimeError
            if not isinstance(ex, MemoryError) and "out of memory" not in str(ex).lower():
                raise
            failed = True
        if not any_rank(failed):
            return buf, flags
        buf = None
        if attempt == 1 or flags & 1:
            raise _lib.NmfxError(_lib.NMFX_ERR_NOMEM, "Engine: the workspace (%d bytes) does not fit on a rank, even without the transposed copy of V" % nbytes)
        release()
        flags |= 1


class Engine:
    """One rank's multiplicative-update engine on HBM-resident V (local column shard), W, H."""

    def __init__(self, V, W, H, divergence="euclidean", T=1, algorithm="nmf", lamW=None, lamH=None, fixW=None, fixH=None,
                 group=None, use_dist=None, path=0, halo=(0, 0), n_valid=None, n_chunks=None, alpha=1.0, beta=1.0, no_vt=False):
        import torch
        self.torch = torch
        if not (V.is_cuda and W.is_cuda and H.is_cuda):
            raise _lib.NmfxError(_lib.NMFX_ERR_NO_DEVICE, "Engine needs CUDA/HIP tensors: there is no CPU fallback")
        self.lib = _lib.load()
        self.V, self.W, self.H = V.contiguous(), W.contiguous(), H.contiguous()
        self.hL, self.hR = int(halo[0]), int(halo[1])          # cnmf shards: H = [left halo | local | right halo], V = [local | right halo]
        self.m = self.V.shape[1]
        self.n = self.H.shape[0] - self.hL - self.hR
        self.K = self.H.shape[1]
        self.T = int(T)
        assert self.V.shape[0] == self.n + self.hR and self.W.numel() == self.m * self.K * self.T
        self.H_local = self.H[self.hL:self.hL + self.n]        # this rank's own columns (a view)
        dist = torch.distributed
        self.dist = dist if (use_dist if use_dist is not None else (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1)) else None
        self.group = group
        self.rank = dist.get_rank(group) if self.dist else 0
        d = _lib.EngineDesc()
        d.m, d.n_local, d.K_total, d.T = self.m, self.n, self.K, self.T
        d.divergence = DIV_CODES[divergence] if isinstance(divergence, str) else int(divergence)
        is_ab = d.divergence == _lib.DIV_AB                      # nmf.m:255-266: alpha / beta are used for 'ab' only, every other divergence forces (1, 1)
        d.alpha, d.beta = (float(alpha), float(beta)) if is_ab else (1.0, 1.0)
        self._keep = []
        for name, val, dt in (("lamW_col", lamW, np.float32), ("lamH_row", lamH, np.float32), ("fixW_col", fixW, np.uint8), ("fixH_row", fixH, np.uint8)):
            if val is not None:
                arr = np.ascontiguousarray(np.broadcast_to(np.asarray(val, dtype=dt), (self.K,)))
                self._keep.append(arr)
                setattr(d, name, arr.ctypes.data_as(C.c_void_p))
        d.device = self.V.device.index or 0
        d.stream = C.c_void_p(torch.cuda.current_stream(self.V.device).cuda_stream)
        d.algorithm = {"nmf": 0, "cnmf": 1, "lnmf": 2}[algorithm]
        self.stop_le = algorithm == "lnmf"                 # lnmf.m:84
        d.path = int(path)
        if self.dist is not None and d.path == 0:
            # every rank must run the same kernels (the packed layout and the summation order of the replicated W update depend on them): the fused paths
            # want at least 64 local columns, so one short shard sends all ranks to the general kernels (as the blocking multi-GPU call does)
            nmin = torch.tensor([float(self.n)], dtype=torch.float64, device=self.V.device)
            self.dist.all_reduce(nmin, op=self.dist.ReduceOp.MIN, group=group)
            if float(nmin.item()) < 64:
                d.path = 1
        d.halo_left, d.halo_right = self.hL, self.hR
        d.n_valid = int(n_valid) if n_valid is not None else self.n + self.hR
        d.flags = 1 if (no_vt or os.environ.get("NMFX_NO_VT")) else 0
        self.desc = d
        nbytes, count = C.c_size_t(0), C.c_size_t(0)
        _lib.check(self.lib.nmfx_engine_packed_count(C.byref(d), C.byref(count)))
        self.packed = torch.zeros(count.value, dtype=torch.float32, device=self.V.device)
        # the workspace may hold a transposed copy of V (euclidean paths; nmfx_engine_desc.flags bit 0 = without).  If it does not fit HERE, every rank gives it
        # up together: the kernel path -- and the summation order of the replicated W update -- follows from the descriptor, which must be the same everywhere
        def size_of(flags):
            d.flags = flags
            _lib.check(self.lib.nmfx_engine_workspace_bytes(C.byref(d), C.byref(nbytes)))
            return nbytes.value

        def any_rank(failed):
            if self.dist is None:
                return failed
            ft = torch.tensor([1.0 if failed else 0.0], dtype=torch.float64, device=self.V.device)
            self.dist.all_reduce(ft, op=self.dist.ReduceOp.MAX, group=group)
            return bool(ft.item())

        self.workspace, d.flags = agree_on_workspace(d.flags, size_of, lambda nb: torch.empty(nb, dtype=torch.uint8, device=self.V.device), any_rank,
                                                     torch.cuda.empty_cache)
        size_of(d.flags)
        self.cost_buf = None
        h = C.c_void_p()
        _lib.check(self.lib.nmfx_engine_create(C.byref(d), self.V.data_ptr(), self.W.data_ptr(), self.H.data_ptr(), self.workspace.data_ptr(),
                                               nbytes.value, self.packed.data_ptr(), C.byref(h)))
        self.h = h
        _lib.check(self.lib.nmfx_engine_set_rank0(self.h, 1 if self.rank == 0 else 0))
        self._cost_t = torch.zeros(1, dtype=torch.float64, device=self.V.device)
        self.path_kind = int(self.lib.nmfx_engine_is_fused(self.h))
        self.cost_lag = int(self.lib.nmfx_engine_cost_lag(self.h))   # cost of iteration i: 0 after hstep(i), 1 after wstep_partial(i+1), 2 after wstep_finish(i+1)
        self.cost_lags = self.cost_lag != 0
        # row chunks of the W-step partial on column shards (all-reduce of chunk c overlapping the compute of chunk c+1): fused
        # path, m a multiple of 128*n_chunks.  Off (1) unless asked for: the K = 256 kernels fill every CU (one 512-VGPR wave per
        # SIMD, 256 workgroups), so a concurrent RCCL kernel can only run by displacing compute workgroups -- whether the overlap
        # wins has to be measured on an xGMI node first; NMFX_W_CHUNKS / n_chunks turn it on.
        self.n_chunks = int(n_chunks) if n_chunks is not None else int(os.environ.get("NMFX_W_CHUNKS", "1"))
        dual = d.divergence in (_lib.DIV_IS, _lib.DIV_AB)   # fused IS / alpha-beta passes produce [N | P] in one piece: no row chunks
        if self.path_kind != 1 or dual or self.m % (128 * max(self.n_chunks, 1)) != 0:
            self.n_chunks = 1
        self.has_halos = bool(self.hL or self.hR)
        if self.has_halos:   # V_hat / cost are refreshed only after the neighbours' new H columns have arrived
            _lib.check(self.lib.nmfx_engine_defer_hstep_finish(self.h, 1))

    def close(self):
        if getattr(self, "h", None):
            self.lib.nmfx_engine_destroy(self.h)
            self.h = None

    __del__ = close

    def init(self):
        _lib.check(self.lib.nmfx_engine_init(self.h))
        # euclidean fused path: the cost in Gram form needs the GLOBAL ||V||^2 for its (rank-independent) mode decision: one 8-byte all-reduce, once
        vv = self.torch.zeros(1, dtype=self.torch.float64, device=self.V.device)
        _lib.check(self.lib.nmfx_engine_sumvv_local(self.h, vv.data_ptr()))
        if self.dist is not None:
            self.dist.all_reduce(vv, group=self.group)
        _lib.check(self.lib.nmfx_engine_sumvv_set_global(self.h, vv.data_ptr()))
        self.torch.cuda.current_stream(self.V.device).synchronize()    # vv goes out of scope

    # ---- the four phases (HIP kernels on this rank's shard) -------------------------------------
    def wstep_partial(self):
        _lib.check(self.lib.nmfx_engine_wstep_partial(self.h))

    def wstep_partial_chunk(self, chunk, nchunks):
        _lib.check(self.lib.nmfx_engine_wstep_partial_chunk(self.h, int(chunk), int(nchunks)))

    def packed_chunk(self, chunk, nchunks):
        """the slice of `packed` that is final after chunk `chunk` (the last one carries the tail)"""
        off, cnt = C.c_size_t(0), C.c_size_t(0)
        _lib.check(self.lib.nmfx_engine_packed_chunk(self.h, int(chunk), int(nchunks), C.byref(off), C.byref(cnt)))
        return self.packed[off.value:off.value + cnt.value]

    def wstep_finish(self):
        _lib.check(self.lib.nmfx_engine_wstep_finish(self.h))

    def hstep(self):
        _lib.check(self.lib.nmfx_engine_hstep(self.h))

    def hstep_finish(self):
        _lib.check(self.lib.nmfx_engine_hstep_finish(self.h))

    def between_allreduces(self, last, lag2_cost_dst=None):
        _lib.check(self.lib.nmfx_engine_between_allreduces_cost(self.h, 1 if last else 0, lag2_cost_dst.data_ptr() if lag2_cost_dst is not None else None))

    def cost_pass(self):
        _lib.check(self.lib.nmfx_engine_cost_pass(self.h))

    def exchange_halos(self):
        """cnmf on column shards: refresh H's halo columns from the neighbouring ranks (T-1 columns each way, point-to-point).
        Rank r sends its first hR' columns to r-1 (their right halo) and its last hL' columns to r+1 (their left halo)."""
        if self.dist is None or (self.hL == 0 and self.hR == 0):
            return
        dist, torch = self.dist, self.torch
        world, rank = dist.get_world_size(self.group), self.rank
        ops, keep = [], []
        h = self.T - 1
        if h == 0:
            return
        if rank > 0:                       # left neighbour exists: receive my left halo, send my first h columns
            ops.append(dist.P2POp(dist.irecv, self.H[0:self.hL], rank - 1, self.group))
            buf = self.H_local[0:h].contiguous(); keep.append(buf)
            ops.append(dist.P2POp(dist.isend, buf, rank - 1, self.group))
        if rank < world - 1:               # right neighbour exists
            ops.append(dist.P2POp(dist.irecv, self.H[self.hL + self.n:self.hL + self.n + self.hR], rank + 1, self.group))
            buf = self.H_local[self.n - h:self.n].contiguous(); keep.append(buf)
            ops.append(dist.P2POp(dist.isend, buf, rank + 1, self.group))
        if ops:
            stream_ordered = dist.get_backend(self.group) == "nccl"   # RCCL point-to-point is ordered on the current stream
            if not stream_ordered:
                torch.cuda.current_stream(self.V.device).synchronize()   # gloo copies through the host: the H update must have finished
            for req in dist.batch_isend_irecv(ops):
                req.wait()
            if not stream_ordered:
                torch.cuda.synchronize(self.V.device)

    def iterate(self, iters, cost_out=None, tolerance=None):
        """At most `iters` full iterations; cost_out: optional fp64 device tensor (>= iters) receiving the GLOBAL cost per iteration.
        tolerance > 0 applies the stop rule of nmf.m:221-224 (needs cost_out).  Returns the number of iterations run."""
        if self.dist is None and not (tolerance is not None and tolerance > 0):
            ptr = cost_out.data_ptr() if cost_out is not None else None
            _lib.check(self.lib.nmfx_engine_iterate(self.h, int(iters), ptr))
            return int(iters)
        return run_sharded_iterations(self, iters, self.dist, self.group, cost_out, tolerance)

    def backup_W(self):
        if getattr(self, "_Wbak", None) is None:
            self._Wbak = self.torch.empty_like(self.W)
        self._Wbak.copy_(self.W)

    def restore_W(self):
        self.W.copy_(self._Wbak)

    def _copy_cost(self, dst):
        _lib.check(self.lib.nmfx_engine_copy_cost(self.h, dst.data_ptr()))

    def cost(self):
        """global cost of the current (W, H) as a Python float (synchronises)"""
        self.cost_pass()
        self._copy_cost(self._cost_t)
        if self.dist is not None:
            self.dist.all_reduce(self._cost_t, group=self.group)
        return float(self._cost_t.item())

    # ---- measurement hooks -------------------------------------------------------------------
    def profile(self, enable=True):
        """hipEvent pairs around the launch groups: True / 1 all of them, 2 the MFMA launch groups only (bench.py), False / 0 off"""
        _lib.check(self.lib.nmfx_engine_profile(self.h, int(enable)))
        self.comm_events = [] if (enable and self.dist is not None and self.V.is_cuda) else None

    def comm_ms(self):
        """after torch.cuda.synchronize(): total ms the compute stream spent in the packed all-reduce while profiling was on"""
        ev = getattr(self, "comm_events", None) or []
        return sum(a.elapsed_time(b) for a, b in ev), len(ev)

    def profile_read(self):
        """after torch.cuda.synchronize(): {tag_name: dict(ms_total, launches, flops, bytes)}"""
        nt = self.lib.nmfx_engine_profile_ntags()
        ms = (C.c_double * nt)()
        cnt = (C.c_int32 * nt)()
        _lib.check(self.lib.nmfx_engine_profile_read(self.h, ms, cnt))
        out = {}
        for t in range(nt):
            f, b = C.c_double(0), C.c_double(0)
            _lib.check(self.lib.nmfx_engine_tag_work(self.h, t, C.byref(f), C.byref(b)))
            out[self.lib.nmfx_engine_profile_tag_name(t).decode()] = dict(ms_total=ms[t], launches=cnt[t], flops=f.value, bytes=b.value)
        return out


# ---- nmfsc on column shards -------------------------------------------------------------------------
class _DevPtr:
    """a raw device pointer as a __cuda_array_interface__ object, so torch can wrap the library's buffers without copying"""

    def __init__(self, ptr, count, typestr):
        self.__cuda_array_interface__ = dict(shape=(int(count),), typestr=typestr, data=(int(ptr), False), version=2, strides=None)


def nmfsc_sharded(V, W, H, n_total=None, W_sparsity=0.0, H_sparsity=0.0, W_fixed=False, H_fixed=False, maxiter=100, tolerance=1e-3,
                  group=None, path=0, allreduce=None, resume=None):
    """nmfsc.m:57-245 with V / H column-sharded over the ranks of `group` and W replicated.

    V: (n_local, m) fp32 CUDA tensor (= column-major m x n_local), W: (K, m), H: (n_local, K); W and H are updated in place
    (W ends identical on every rank).  Every cross-rank sum is requested by libnmfx through a callback and served here by
    torch.distributed (RCCL with the nccl backend): one [V*H' | H*H'] all-reduce per outer iteration, 8 bytes per objective
    evaluation, 4*K doubles per projfunc reduction.  `allreduce(tensor, op)` replaces torch.distributed (tests).
    Returns (cost ndarray, info dict).  tolerance < 0 disables the stop rule.
    resume = the info dict of a previous call on the same buffers: V is taken as already rescaled (info["Vs"]), W / H as that
    call's final state and the line searches continue from its step sizes -- two calls of a + b iterations equal one of a + b."""
    import torch
    dist = torch.distributed
    lib = _lib.load()
    if not (V.is_cuda and W.is_cuda and H.is_cuda):
        raise _lib.NmfxError(_lib.NMFX_ERR_NO_DEVICE, "nmfsc_sharded needs CUDA/HIP tensors: there is no CPU fallback")
    V, dev = V.contiguous(), V.device
    assert W.is_contiguous() and H.is_contiguous() and W.dtype == H.dtype == V.dtype == torch.float32
    n_local, m = V.shape
    K = H.shape[1]
    assert H.shape[0] == n_local and W.shape == (K, m)
    use_dist = allreduce is None and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    if allreduce is None and use_dist:
        ops = {_lib.REDUCE_SUM: dist.ReduceOp.SUM, _lib.REDUCE_MAX: dist.ReduceOp.MAX}

        def allreduce(t, op):
            dist.all_reduce(t, op=ops[op], group=group)
    if resume is not None:
        Vs, vmax = resume["Vs"], resume["vmax"]
    else:
        # nmfsc.m:57-62: data must be non-negative; V = V / max(V(:)) with the GLOBAL max -- both steps are libnmfx kernels (nmfx_minmax_dev /
        # nmfx_scale_dev: what a C host driving nmfx_nmfsc_dev calls too); torch only allocates and carries the MAX all-reduce of [max, -min]
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        mm = torch.empty(2, dtype=torch.float64, device=dev)
        _lib.check(lib.nmfx_minmax_dev(stream, V.data_ptr(), V.numel(), mm.data_ptr()))
        if allreduce is not None:
            allreduce(mm, _lib.REDUCE_MAX)
        vmax, vmin = float(mm[0]), -float(mm[1])
        if vmin < 0:
            raise ValueError("Negative values in data!")
        Vs = torch.empty_like(V)
        _lib.check(lib.nmfx_scale_dev(stream, V.data_ptr(), V.numel(), vmax, Vs.data_ptr()))
    nt = torch.tensor([float(n_local)], dtype=torch.float64, device=dev)
    if allreduce is not None:
        allreduce(nt, _lib.REDUCE_SUM)
    n_sum = int(round(float(nt[0])))
    if n_total is None:
        n_total = n_sum
    assert int(n_total) == n_sum, "n_total does not match the sum of the shards' columns"
    errors = []

    def _cb(ctx, ptr, count, dtype, op, stream):
        try:
            t = torch.as_tensor(_DevPtr(ptr, count, "<f4" if dtype == _lib.F32 else "<f8"), device=dev)
            allreduce(t, op)
            return 0
        except Exception as ex:   # never let an exception cross the C frame
            errors.append(ex)
            return 1

    cb = _lib.ALLREDUCE_FN(_cb) if allreduce is not None else C.cast(None, _lib.ALLREDUCE_FN)
    cost = np.zeros(int(maxiter) + 1)
    tH, tW = np.zeros(int(maxiter), dtype=np.int32), np.zeros(int(maxiter), dtype=np.int32)
    fw, fh = np.asarray([bool(W_fixed)], dtype=np.uint8), np.asarray([bool(H_fixed)], dtype=np.uint8)
    p = _lib.Problem()
    p.m, p.n, p.K_total, p.T, p.dtype = m, n_local, K, 1, _lib.F32
    p.num_sources = 1
    p.W_fixed, p.H_fixed = fw.ctypes.data_as(C.c_void_p), fh.ctypes.data_as(C.c_void_p)
    p.maxiter, p.tolerance = int(maxiter), float(tolerance)
    p.device = dev.index or 0
    p.sc_W_sparsity, p.sc_H_sparsity, p.path = float(W_sparsity), float(H_sparsity), int(path)
    if resume is not None:
        p.sc_resume, p.sc_stepsize_H0, p.sc_stepsize_W0 = 1, float(resume["stepsizeH"]), float(resume["stepsizeW"])
    r = _lib.Result()
    r.cost, r.tries_H, r.tries_W = cost.ctypes.data_as(C.c_void_p), tH.ctypes.data_as(C.c_void_p), tW.ctypes.data_as(C.c_void_p)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    rc = lib.nmfx_nmfsc_dev(C.byref(p), Vs.data_ptr(), W.data_ptr(), H.data_ptr(), int(n_total), stream, cb, None, C.byref(r))
    if errors:
        raise errors[0]
    _lib.check(rc)
    info = dict(triesH=[int(t) for t in tH if t > 0], triesW=[int(t) for t in tW if t > 0], stepsizeH=r.stepsize_H, stepsizeW=r.stepsize_W,
                converged_early=bool(r.converged_early), vmax=vmax, Vs=Vs)
    return cost[: r.cost_len].copy(), info

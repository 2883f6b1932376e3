// nmfsc (nmfsc.m:57-245) behind the blocking call on N GPUs of the calling process: nmfx_problem.n_gpus > 1.
// V and H are column-sharded, W is replicated.  The line searches are host-driven and branch on every objective, so each
// shard gets its own host thread running the device-level entry point (nmfx_nmfsc_dev) on its own stream; every cross-shard sum that entry point
// asks for -- the [V*H' | H*H'] terms of an outer iteration, 8 bytes per objective, 4*K doubles per reduction of the distributed projfunc
// (projfunc.m:22-53) -- is served here by a peer all-reduce over xGMI mappings: the threads meet at a spin barrier to publish their buffer
// addresses, stream order across devices is carried by events (no stream is ever drained), sums run in the fixed order 0 .. N-1 so that every
// device holds bit-identical results and therefore takes the same branches.  device_ids may name one device several times (N shards on one GPU:
// how the 1-GPU test box runs this path).
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "api_common.h"

using namespace nmfx;

namespace {

struct RawPtrs { const void *p[NMFX_MAX_GPUS]; };

// dst[i] = sum_h src[h][off + i], h = 0 .. ndev-1 in that order (dst may be src[self] + off: element-wise in place)
template <typename T>
__global__ void peer_sum_kernel(RawPtrs bufs, int ndev, long off, long count, T *dst) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    T s = static_cast<const T *>(bufs.p[0])[off + i];
    for (int h = 1; h < ndev; ++h) s += static_cast<const T *>(bufs.p[h])[off + i];
    dst[i] = s;
}
template <typename T>
nmfx_status peer_sum(hipStream_t st, const RawPtrs &bufs, int ndev, long off, long count, T *dst) {
    if (count <= 0) return NMFX_OK;
    hipLaunchKernelGGL(peer_sum_kernel<T>, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, bufs, ndev, off, count, dst);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

struct SpinBarrier {
    std::atomic<int> arrived{0}, generation{0};
    int n = 0;
    std::atomic<int> *abort = nullptr;
    bool wait() {   // false: a peer thread has failed, nobody waits for it any more
        const int g = generation.load(std::memory_order_acquire);
        if (arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
            arrived.store(0, std::memory_order_relaxed);
            generation.fetch_add(1, std::memory_order_release);
            return abort->load(std::memory_order_acquire) == 0;
        }
        for (unsigned spin = 0; generation.load(std::memory_order_acquire) == g; ++spin) {
            if (abort->load(std::memory_order_acquire)) return false;
            if (spin > 4096) std::this_thread::yield();
        }
        return abort->load(std::memory_order_acquire) == 0;
    }
};

constexpr size_t SMALL_BYTES = 64 * 1024;   // up to here every device sums the whole buffer itself (two barriers); above, slices + gather (three)

struct ThreadComm {
    int N = 0;
    int dev[NMFX_MAX_GPUS];
    hipStream_t st[NMFX_MAX_GPUS] = {};
    hipEvent_t evA[NMFX_MAX_GPUS] = {}, evB[NMFX_MAX_GPUS] = {}, evC[NMFX_MAX_GPUS] = {};
    void *ptr[NMFX_MAX_GPUS] = {};
    void *tmp[NMFX_MAX_GPUS] = {};
    std::atomic<int> abort{0};
    SpinBarrier bar;
};
struct RankCtx { ThreadComm *tc; int g; };

nmfx_status allreduce_body(ThreadComm &C, int g, void *dev_ptr, long count, int dtype, hipStream_t st) {
    const int N = C.N;
    const size_t esz = dtype == NMFX_F64 ? 8 : 4, bytes = (size_t)count * esz;
    C.ptr[g] = dev_ptr;
    NMFX_HIP(hipEventRecord(C.evA[g], st));
    if (!C.bar.wait()) { set_error("nmfsc on %d devices: a peer shard failed", N); return NMFX_ERR_INVALID; }
    RawPtrs bufs{};
    for (int h = 0; h < N; ++h) bufs.p[h] = C.ptr[h];
    for (int h = 0; h < N; ++h) if (h != g) NMFX_HIP(hipStreamWaitEvent(st, C.evA[h], 0));   // every shard's contribution is complete
    if (bytes <= SMALL_BYTES) {
        if (dtype == NMFX_F64) TRY(peer_sum<double>(st, bufs, N, 0, count, static_cast<double *>(C.tmp[g])));
        else TRY(peer_sum<float>(st, bufs, N, 0, count, static_cast<float *>(C.tmp[g])));
        NMFX_HIP(hipEventRecord(C.evB[g], st));
        if (!C.bar.wait()) { set_error("nmfsc on %d devices: a peer shard failed", N); return NMFX_ERR_INVALID; }
        for (int h = 0; h < N; ++h) if (h != g) NMFX_HIP(hipStreamWaitEvent(st, C.evB[h], 0));   // ... and nobody is still reading mine
        NMFX_HIP(hipMemcpyAsync(dev_ptr, C.tmp[g], bytes, hipMemcpyDeviceToDevice, st));
        return NMFX_OK;
    }
    // reduce-scatter (device g owns slice g, summed in place out of the peers' HBM) + all-gather (peer copies), as the packed exchange of nmf
    const long per = (long)((((size_t)count + N - 1) / N + 3) & ~(size_t)3);
    auto slice = [&](int h, long *off, long *cnt) { *off = std::min(count, per * h); *cnt = std::min(count, per * (h + 1)) - *off; };
    long off, cnt;
    slice(g, &off, &cnt);
    if (dtype == NMFX_F64) TRY(peer_sum<double>(st, bufs, N, off, cnt, static_cast<double *>(dev_ptr) + off));
    else TRY(peer_sum<float>(st, bufs, N, off, cnt, static_cast<float *>(dev_ptr) + off));
    NMFX_HIP(hipEventRecord(C.evB[g], st));
    if (!C.bar.wait()) { set_error("nmfsc on %d devices: a peer shard failed", N); return NMFX_ERR_INVALID; }
    for (int h = 0; h < N; ++h) {
        if (h == g) continue;
        slice(h, &off, &cnt);
        NMFX_HIP(hipStreamWaitEvent(st, C.evB[h], 0));
        if (cnt > 0) NMFX_HIP(hipMemcpyPeerAsync(static_cast<char *>(dev_ptr) + (size_t)off * esz, C.dev[g], static_cast<const char *>(C.ptr[h]) + (size_t)off * esz, C.dev[h], (size_t)cnt * esz, st));
    }
    NMFX_HIP(hipEventRecord(C.evC[g], st));
    if (!C.bar.wait()) { set_error("nmfsc on %d devices: a peer shard failed", N); return NMFX_ERR_INVALID; }
    for (int h = 0; h < N; ++h) if (h != g) NMFX_HIP(hipStreamWaitEvent(st, C.evC[h], 0));   // nobody overwrites its buffer while a peer still copies out of it
    return NMFX_OK;
}

int32_t thread_allreduce(void *ctx, void *dev_ptr, int64_t count, int32_t dtype, int32_t op, void *stream) {
    RankCtx *rc = static_cast<RankCtx *>(ctx);
    ThreadComm &C = *rc->tc;
    if (op != NMFX_REDUCE_SUM || (dtype != NMFX_F32 && dtype != NMFX_F64) || count <= 0) { C.abort.store(1); return 1; }
    if (allreduce_body(C, rc->g, dev_ptr, (long)count, dtype, static_cast<hipStream_t>(stream)) != NMFX_OK) { C.abort.store(1); return 1; }
    return 0;
}

struct ScMulti {
    ThreadComm C;
    DevBuf V[NMFX_MAX_GPUS], W[NMFX_MAX_GPUS], H[NMFX_MAX_GPUS], tmp[NMFX_MAX_GPUS];
    ~ScMulti() {
        for (int g = 0; g < C.N; ++g) {
            (void)hipSetDevice(C.dev[g]);
            if (C.st[g]) (void)hipStreamSynchronize(C.st[g]);
        }
        staging_quiesce();   // (the ingest left its DMA-done events recorded on these streams)
        for (int g = 0; g < C.N; ++g) {
            (void)hipSetDevice(C.dev[g]);
            unpool_event(C.dev[g], C.evA[g]); unpool_event(C.dev[g], C.evB[g]); unpool_event(C.dev[g], C.evC[g]);
            unpool_stream(C.dev[g], C.st[g]);   // (drained above)
        }
    }
};

}  // namespace

namespace nmfx {

nmfx_status run_nmfsc_multi(const nmfx_problem *p, nmfx_result *r) {
    TRY(validate_problem(p, r, true));
    if (p->T != 1) { set_error("nmfsc: T must be 1"); return NMFX_ERR_INVALID; }
    const int N = p->n_gpus, K = p->K_total;
    const long m = p->m, n = p->n;
    if (N > NMFX_MAX_GPUS || N > n) { set_error("n_gpus = %d: at most %d devices and one column per device", N, NMFX_MAX_GPUS); return NMFX_ERR_INVALID; }
    if (p->sc_resume) { set_error("nmfsc: sc_resume belongs to the device-level entry point"); return NMFX_ERR_INVALID; }
    // column shards run nmfx_nmfsc_dev, which takes the fused kernels only: say so HERE, before anything is uploaded or a thread is started (nmf / cnmf fall back
    // to the general kernels for short shards; nmfsc has no sharded general path)
    if (p->path == 1 || K > 256 || ((m < 64 || n / N < 64) && p->path != 2)) {
        set_error("nmfsc on %d devices: fused kernels only -- K <= 256 (got %d), m >= 64 (got %ld) and at least 64 columns per device (got %ld); use fewer devices or n_gpus = 1",
                  N, K, m, n / N);
        return NMFX_ERR_UNSUPPORTED;
    }
    double vmin = INFINITY, vmax = -INFINITY;   // nmfsc.m:57-62 on the whole matrix
    host_minmax(p->V, p->dtype, (size_t)m * n, &vmin, &vmax);
    if (vmin < 0) { set_error("Negative values in data!"); return NMFX_ERR_NEGATIVE; }
    DeviceGuard dg_;
    ScMulti M;
    ThreadComm &C = M.C;
    for (int g = 0; g < N; ++g) {
        C.dev[g] = p->device_ids ? p->device_ids[g] : g;
        TRY(check_device(C.dev[g]));
    }
    for (int g = 0; g < N; ++g)
        for (int h = 0; h < N; ++h) {
            if (C.dev[g] == C.dev[h]) continue;
            int can = 0;
            NMFX_HIP(hipDeviceCanAccessPeer(&can, C.dev[g], C.dev[h]));
            if (!can) { set_error("device %d cannot access device %d as a peer", C.dev[g], C.dev[h]); return NMFX_ERR_UNSUPPORTED; }
            NMFX_HIP(hipSetDevice(C.dev[g]));
            hipError_t pe = hipDeviceEnablePeerAccess(C.dev[h], 0);
            if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) { set_error("hipDeviceEnablePeerAccess(%d -> %d): %s", C.dev[g], C.dev[h], hipGetErrorString(pe)); return NMFX_ERR_HIP; }
            (void)hipGetLastError();
        }
    long lo[NMFX_MAX_GPUS + 1];
    lo[0] = 0;
    for (int g = 0; g < N; ++g) lo[g + 1] = lo[g] + n / N + (g < n % N ? 1 : 0);   // contiguous column blocks, as engine.shard_columns
    const size_t mK = (size_t)m * K, es = dsize(p->dtype);
    for (int g = 0; g < N; ++g) {
        NMFX_HIP(hipSetDevice(C.dev[g]));
        C.N = g + 1;
        TRY(pool_stream(C.dev[g], &C.st[g]));
        TRY(pool_event(C.dev[g], &C.evA[g])); TRY(pool_event(C.dev[g], &C.evB[g])); TRY(pool_event(C.dev[g], &C.evC[g]));
        const long nl = lo[g + 1] - lo[g];
        TRY(M.V[g].alloc((size_t)m * nl * 4)); TRY(M.W[g].alloc(mK * 4)); TRY(M.H[g].alloc((size_t)K * nl * 4)); TRY(M.tmp[g].alloc(SMALL_BYTES));
        C.tmp[g] = M.tmp[g].p;
        TRY(upload(C.st[g], static_cast<const char *>(p->V) + (size_t)m * lo[g] * es, p->dtype, M.V[g].as<float>(), (size_t)m * nl, vmax));   // V / max(V(:))
        TRY(upload(C.st[g], p->W_init, p->dtype, M.W[g].as<float>(), mK, 1.0));
        TRY(upload(C.st[g], static_cast<const char *>(p->H_init) + (size_t)K * lo[g] * es, p->dtype, M.H[g].as<float>(), (size_t)K * nl, 1.0));
        NMFX_HIP(hipStreamSynchronize(C.st[g]));
    }
    C.bar.n = N;
    C.bar.abort = &C.abort;
    // one host thread per shard; shard 0 reports straight into the caller's result, the others into scratch (identical by construction)
    std::vector<nmfx_result> res(N);
    std::vector<std::vector<double>> costs(N);
    std::vector<std::vector<int32_t>> tries(N);
    std::vector<nmfx_status> rc(N, NMFX_OK);
    std::vector<std::string> msg(N);
    std::vector<RankCtx> ctx(N);
    const size_t ntries = (size_t)p->maxiter;
    for (int g = 0; g < N; ++g) {
        ctx[g] = RankCtx{&C, g};
        res[g] = nmfx_result{};
        if (g == 0) { res[g].cost = r->cost; res[g].tries_H = r->tries_H; res[g].tries_W = r->tries_W; }
        else {
            costs[g].assign((size_t)p->maxiter + 1, 0.0);
            tries[g].assign(2 * ntries, 0);
            res[g].cost = costs[g].data(); res[g].tries_H = tries[g].data(); res[g].tries_W = tries[g].data() + ntries;
        }
    }
    auto worker = [&](int g) {
        nmfx_problem q = *p;
        q.n = lo[g + 1] - lo[g];
        q.dtype = NMFX_F32; q.V = q.W_init = q.H_init = nullptr;
        q.device = C.dev[g]; q.n_gpus = 0; q.device_ids = nullptr;
        (void)hipSetDevice(C.dev[g]);
        rc[g] = nmfx_nmfsc_dev(&q, M.V[g].as<float>(), M.W[g].as<float>(), M.H[g].as<float>(), n, C.st[g], thread_allreduce, &ctx[g], &res[g]);
        if (rc[g] != NMFX_OK) { msg[g] = nmfx_last_error(); C.abort.store(1); }
        (void)hipStreamSynchronize(C.st[g]);
        sc_thread_cleanup();
    };
    std::vector<std::thread> th;
    for (int g = 1; g < N; ++g) th.emplace_back(worker, g);
    worker(0);
    for (auto &t : th) t.join();
    for (int g = 0; g < N; ++g)
        if (rc[g] != NMFX_OK && msg[g].find("a peer shard failed") == std::string::npos) { set_error("%s", msg[g].c_str()); return rc[g]; }
    for (int g = 0; g < N; ++g) if (rc[g] != NMFX_OK) { set_error("%s", msg[g].c_str()); return rc[g]; }
    for (int g = 1; g < N; ++g)
        if (res[g].cost_len != res[0].cost_len || res[g].converged_early != res[0].converged_early) { set_error("nmfsc on %d devices: the shards took different branches", N); return NMFX_ERR_INVALID; }
    r->cost_len = res[0].cost_len; r->iters_run = res[0].iters_run; r->converged_early = res[0].converged_early;
    r->stepsize_H = res[0].stepsize_H; r->stepsize_W = res[0].stepsize_W;
    for (int g = 0; g < N; ++g) {
        NMFX_HIP(hipSetDevice(C.dev[g]));
        const long nl = lo[g + 1] - lo[g];
        if (g == 0) TRY(download(C.st[g], M.W[g].as<float>(), p->dtype, r->W, mK));
        TRY(download(C.st[g], M.H[g].as<float>(), p->dtype, static_cast<char *>(r->H) + (size_t)K * lo[g] * es, (size_t)K * nl));
    }
    return NMFX_OK;
}

}  // namespace nmfx

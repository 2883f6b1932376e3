// Host <-> device staging of the blocking (MATLAB-facing) entry points: fp32 / fp64 host buffers in pageable memory on one side, fp32
// in HBM on the other.
//
// The caller's arrays (a MATLAB mxArray, a NumPy array) are pageable, 8 GiB of doubles at BASELINE config 3, and the device wants
// floats.  So the conversion runs on the HOST, in threads, while the data is copied into one of two pinned staging buffers, and the DMA
// engine moves fp32 -- half the PCIe bytes of the doubles -- out of the other one at the same time:
//
//     chunk c   : host threads   V[c] (f64, pageable) --(float)(x / s)--> pinned[c & 1]     (memory-bandwidth bound, ~T cores)
//     chunk c-1 : DMA            pinned[(c-1) & 1] ----------------------> HBM              (hipMemcpyAsync, returns at once)
//
// One event per buffer says when its DMA has drained; nothing else synchronises, and no kernel runs.  The values are the same the
// device-side conversion produced before: IEEE double division, then round-to-nearest to float.
// Downloads mirror it (DMA of chunk c into one buffer while the threads widen chunk c-1 out of the other).
#include <atomic>
#include <mutex>
#include <thread>

#include "api_common.h"

namespace nmfx {

namespace {

constexpr size_t CHUNK_ELEMS = (size_t)16 << 20;   // 64 MiB of floats per pinned buffer (128 MiB of the caller's doubles)
constexpr size_t MIN_PER_THREAD = (size_t)1 << 18;

// two pinned, portable staging buffers per process, allocated on first use and kept (pinning 128 MiB costs tens of ms; a MATLAB session
// calls nmf() many times).  One transfer at a time uses them.
struct PinnedPool {
    std::mutex mu;
    float *buf[2] = {nullptr, nullptr};
    hipEvent_t ev[NMFX_MAX_GPUS][2] = {};   // an event belongs to the device it was created on: one pair per device that ever transfers
    int pend_dev[2] = {-1, -1};             // device whose DMA last used buffer b (-1: idle)
    nmfx_status get(int *dev_out) {
        int dev = 0;
        NMFX_HIP(hipGetDevice(&dev));
        if (dev < 0 || dev >= NMFX_MAX_GPUS) { set_error("host staging: device %d out of range", dev); return NMFX_ERR_INVALID; }
        for (int b = 0; b < 2; ++b) {
            if (!buf[b]) NMFX_HIP(hipHostMalloc(reinterpret_cast<void **>(&buf[b]), CHUNK_ELEMS * sizeof(float), hipHostMallocPortable));
            if (!ev[dev][b]) NMFX_HIP(hipEventCreateWithFlags(&ev[dev][b], hipEventDisableTiming));
        }
        *dev_out = dev;
        return NMFX_OK;
    }
    nmfx_status wait(int b) {   // the DMA that last used buffer b has drained
        if (pend_dev[b] >= 0) { NMFX_HIP(hipEventSynchronize(ev[pend_dev[b]][b])); pend_dev[b] = -1; }
        return NMFX_OK;
    }
    nmfx_status mark(int b, int dev, hipStream_t st) {
        NMFX_HIP(hipEventRecord(ev[dev][b], st));
        pend_dev[b] = dev;
        return NMFX_OK;
    }
};
PinnedPool g_pool;

int io_threads() {
    static const int n = [] {
        const char *e = getenv("NMFX_IO_THREADS");
        int t = e ? atoi(e) : (int)std::thread::hardware_concurrency();
        return t < 1 ? 1 : (t > 32 ? 32 : t);
    }();
    return n;
}

// run f(lo, hi) over [0, n) on up to io_threads() threads (the calling thread takes the first slice)
template <class F> void parallel_for(size_t n, F f) {
    int T = (int)std::min<size_t>((size_t)io_threads(), n / MIN_PER_THREAD + 1);
    if (T <= 1) { f((size_t)0, n); return; }
    const size_t per = ((n + T - 1) / T + 15) & ~(size_t)15;
    std::vector<std::thread> th;
    th.reserve(T - 1);
    for (int t = 1; t < T; ++t) {
        const size_t lo = std::min(n, per * t), hi = std::min(n, per * (t + 1));
        if (lo < hi) th.emplace_back([=] { f(lo, hi); });
    }
    f((size_t)0, std::min(n, per));
    for (auto &x : th) x.join();
}

void narrow(const void *src, int dtype, size_t off, float *dst, size_t n, double s) {
    if (dtype == NMFX_F64) {
        const double *p = static_cast<const double *>(src) + off;
        if (s == 1.0) parallel_for(n, [=](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) dst[i] = (float)p[i]; });
        else parallel_for(n, [=](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) dst[i] = (float)(p[i] / s); });
    } else {
        const float *p = static_cast<const float *>(src) + off;
        if (s == 1.0) parallel_for(n, [=](size_t lo, size_t hi) { memcpy(dst + lo, p + lo, (hi - lo) * sizeof(float)); });
        else parallel_for(n, [=](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) dst[i] = (float)((double)p[i] / s); });
    }
}

void widen(const float *src, int dtype, void *dst, size_t off, size_t n) {
    if (dtype == NMFX_F64) {
        double *p = static_cast<double *>(dst) + off;
        parallel_for(n, [=](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) p[i] = (double)src[i]; });
    } else {
        float *p = static_cast<float *>(dst) + off;
        parallel_for(n, [=](size_t lo, size_t hi) { memcpy(p + lo, src + lo, (hi - lo) * sizeof(float)); });
    }
}

thread_local IoStats g_io;

struct ResPool {
    std::mutex mu;
    std::vector<hipStream_t> st[NMFX_MAX_GPUS];
    std::vector<hipEvent_t> ev[NMFX_MAX_GPUS], evt[NMFX_MAX_GPUS];   // evt: events WITH timing (the exchange brackets of the multi-GPU calls)
};
ResPool g_res;

// NMFX_NO_POOL=1 (development switch, read once): streams and events are created per call and destroyed when handed back, as in round 3 before the pool --
// the configuration the host-sanitizer campaign runs in (tests/host_asan/, profiles/archive/r4_*): the pool must not be what keeps a lifetime bug from showing
bool pool_off() {
    static const bool off = [] { const char *e = getenv("NMFX_NO_POOL"); return e && e[0] == '1'; }();
    return off;
}

}  // namespace

nmfx_status pool_stream(int device, hipStream_t *st) {
    if (device < 0 || device >= NMFX_MAX_GPUS) { set_error("pool_stream: device %d out of range", device); return NMFX_ERR_INVALID; }
    if (!pool_off()) {
        std::lock_guard<std::mutex> lk(g_res.mu);
        if (!g_res.st[device].empty()) { *st = g_res.st[device].back(); g_res.st[device].pop_back(); return NMFX_OK; }
    }
    NMFX_HIP(hipSetDevice(device));
    NMFX_HIP(hipStreamCreateWithFlags(st, hipStreamNonBlocking));
    return NMFX_OK;
}
nmfx_status pool_event(int device, hipEvent_t *ev) {
    if (device < 0 || device >= NMFX_MAX_GPUS) { set_error("pool_event: device %d out of range", device); return NMFX_ERR_INVALID; }
    if (!pool_off()) {
        std::lock_guard<std::mutex> lk(g_res.mu);
        if (!g_res.ev[device].empty()) { *ev = g_res.ev[device].back(); g_res.ev[device].pop_back(); return NMFX_OK; }
    }
    NMFX_HIP(hipSetDevice(device));
    NMFX_HIP(hipEventCreateWithFlags(ev, hipEventDisableTiming));
    return NMFX_OK;
}
nmfx_status pool_event_timed(int device, hipEvent_t *ev) {
    if (device < 0 || device >= NMFX_MAX_GPUS) { set_error("pool_event_timed: device %d out of range", device); return NMFX_ERR_INVALID; }
    if (!pool_off()) {
        std::lock_guard<std::mutex> lk(g_res.mu);
        if (!g_res.evt[device].empty()) { *ev = g_res.evt[device].back(); g_res.evt[device].pop_back(); return NMFX_OK; }
    }
    NMFX_HIP(hipSetDevice(device));
    NMFX_HIP(hipEventCreate(ev));
    return NMFX_OK;
}
void unpool_event_timed(int device, hipEvent_t ev) {
    if (!ev || device < 0 || device >= NMFX_MAX_GPUS) return;
    if (pool_off()) { (void)hipEventDestroy(ev); return; }
    std::lock_guard<std::mutex> lk(g_res.mu);
    g_res.evt[device].push_back(ev);
}
void unpool_stream(int device, hipStream_t st) {
    if (!st || device < 0 || device >= NMFX_MAX_GPUS) return;
    if (pool_off()) { (void)hipStreamDestroy(st); return; }
    std::lock_guard<std::mutex> lk(g_res.mu);
    g_res.st[device].push_back(st);
}
void unpool_event(int device, hipEvent_t ev) {
    if (!ev || device < 0 || device >= NMFX_MAX_GPUS) return;
    if (pool_off()) { (void)hipEventDestroy(ev); return; }
    std::lock_guard<std::mutex> lk(g_res.mu);
    g_res.ev[device].push_back(ev);
}

IoStats &io_stats() { return g_io; }

// The staging buffers remember the event of their last DMA (PinnedPool::pend_dev) and wait for it before they are refilled -- by then possibly in a LATER call.
// A multi-device call records those events on ITS OWN streams, which go back to the pool (or, with NMFX_NO_POOL, are destroyed) when it ends: settle the
// bookkeeping while the streams still exist, so that no later call ever synchronises an event whose stream is gone.  The callers have drained their streams.
void staging_quiesce() {
    std::lock_guard<std::mutex> lk(g_pool.mu);
    for (int b = 0; b < 2; ++b)
        if (g_pool.wait(b) != NMFX_OK) { (void)hipGetLastError(); g_pool.pend_dev[b] = -1; }
}

// host (f32 / f64, pageable) -> device fp32, out = in / divide_by.  Returns once the last chunk is QUEUED on `st`; the caller's buffer is
// no longer referenced at that point (the DMA reads the pinned copies), and work queued on `st` afterwards sees the data.
nmfx_status upload(hipStream_t st, const void *host, int dtype, float *dev, size_t count, double divide_by) {
    std::lock_guard<std::mutex> lk(g_pool.mu);
    int cur = 0;
    TRY(g_pool.get(&cur));
    for (size_t off = 0, c = 0; off < count; off += CHUNK_ELEMS, ++c) {
        const int b = (int)(c & 1);
        const size_t n = std::min(CHUNK_ELEMS, count - off);
        TRY(g_pool.wait(b));
        narrow(host, dtype, off, g_pool.buf[b], n, divide_by);
        NMFX_HIP(hipMemcpyAsync(dev + off, g_pool.buf[b], n * sizeof(float), hipMemcpyHostToDevice, st));
        TRY(g_pool.mark(b, cur, st));
    }
    g_io.h2d_bytes_host += (double)count * (double)dsize(dtype);
    g_io.h2d_bytes_pcie += (double)count * 4.0;
    return NMFX_OK;
}

// device fp32 -> host (f32 / f64); complete when it returns
nmfx_status download(hipStream_t st, const float *dev, int dtype, void *host, size_t count) {
    std::lock_guard<std::mutex> lk(g_pool.mu);
    int cur = 0;
    TRY(g_pool.get(&cur));
    size_t prev_off = 0, prev_n = 0;
    int prev_b = -1;
    for (size_t off = 0, c = 0; off < count; off += CHUNK_ELEMS, ++c) {
        const int b = (int)(c & 1);
        const size_t n = std::min(CHUNK_ELEMS, count - off);
        TRY(g_pool.wait(b));
        NMFX_HIP(hipMemcpyAsync(g_pool.buf[b], dev + off, n * sizeof(float), hipMemcpyDeviceToHost, st));
        TRY(g_pool.mark(b, cur, st));
        if (prev_b >= 0) { TRY(g_pool.wait(prev_b)); widen(g_pool.buf[prev_b], dtype, host, prev_off, prev_n); }
        prev_b = b; prev_off = off; prev_n = n;
    }
    if (prev_b >= 0) { TRY(g_pool.wait(prev_b)); widen(g_pool.buf[prev_b], dtype, host, prev_off, prev_n); }
    g_io.d2h_bytes_host += (double)count * (double)dsize(dtype);
    return NMFX_OK;
}

// min / max of a host array (nmfsc.m:57-62: the sign check and the global rescale), on the same threads
void host_minmax(const void *host, int dtype, size_t count, double *vmin, double *vmax) {
    const int T = io_threads();
    std::vector<double> lo_(T + 1, INFINITY), hi_(T + 1, -INFINITY);
    std::atomic<int> slot{0};
    auto body = [&](size_t lo, size_t hi) {
        double mn = INFINITY, mx = -INFINITY;
        if (dtype == NMFX_F64) { const double *p = static_cast<const double *>(host); for (size_t i = lo; i < hi; ++i) { if (p[i] < mn) mn = p[i]; if (p[i] > mx) mx = p[i]; } }
        else { const float *p = static_cast<const float *>(host); for (size_t i = lo; i < hi; ++i) { if (p[i] < mn) mn = p[i]; if (p[i] > mx) mx = p[i]; } }
        const int s = slot.fetch_add(1);
        lo_[s] = mn; hi_[s] = mx;
    };
    parallel_for(count, body);
    *vmin = INFINITY; *vmax = -INFINITY;
    for (int s = 0; s <= T; ++s) { if (lo_[s] < *vmin) *vmin = lo_[s]; if (hi_[s] > *vmax) *vmax = hi_[s]; }
}

nmfx_status validate_problem(const nmfx_problem *p, const nmfx_result *r, bool nmfsc, bool need_H_init) {
    if (!p || !r) { set_error("null problem/result"); return NMFX_ERR_INVALID; }
    if (p->m <= 0 || p->n <= 0 || p->K_total <= 0 || p->T <= 0) { set_error("m, n, K_total, T must be positive"); return NMFX_ERR_INVALID; }
    if (!p->V || !p->W_init || (need_H_init && !p->H_init) || !r->W || !r->H || !r->cost) { set_error("V, W_init, H_init, result.W, result.H, result.cost are required"); return NMFX_ERR_INVALID; }
    if (p->dtype != NMFX_F32 && p->dtype != NMFX_F64) { set_error("dtype must be NMFX_F32 or NMFX_F64"); return NMFX_ERR_INVALID; }
    if (p->maxiter <= 0) { set_error("maxiter must be positive (the wrapper applies the reference default)"); return NMFX_ERR_INVALID; }
    if (p->multi_backend < 0 || p->multi_backend > 2) { set_error("multi_backend = %d: 0 (auto), 1 (peer exchange) or 2 (RCCL)", p->multi_backend); return NMFX_ERR_INVALID; }
    if (p->n_gpus < 0 || p->n_gpus > NMFX_MAX_GPUS) { set_error("n_gpus = %d: 0 .. %d", p->n_gpus, NMFX_MAX_GPUS); return NMFX_ERR_INVALID; }
    if (!nmfsc) {
        if (p->num_sources < 1) { set_error("num_sources must be >= 1"); return NMFX_ERR_INVALID; }
        if (p->num_sources > 1 && !p->K_s) { set_error("K_s is required when num_sources > 1"); return NMFX_ERR_INVALID; }
        if (p->K_s) {
            long sum = 0;
            for (int s = 0; s < p->num_sources; ++s) { if (p->K_s[s] <= 0) { set_error("K_s entries must be positive"); return NMFX_ERR_INVALID; } sum += p->K_s[s]; }
            if (sum != p->K_total) { set_error("sum(K_s) = %ld != K_total = %d", sum, p->K_total); return NMFX_ERR_INVALID; }
        }
        if (p->divergence == NMFX_DIV_AB && p->alpha == 0 && p->beta == 0) {   // nmf.m:120-122
            set_error("alpha = 0 and beta = 0 is not supported at this time.");
            return NMFX_ERR_INVALID;
        }
    }
    return NMFX_OK;
}

}  // namespace nmfx

extern "C" {

// Measurement hook (bench.py --api blocking): seconds the last blocking factorisation on this thread spent moving data in, iterating, and
// moving results out, and the bytes it took from / returned to host arrays.
nmfx_status nmfx_last_call_exchange(double *ms_per_exchange, int32_t *exchanges_timed, int32_t *backend) {
    const nmfx::IoStats &s = nmfx::io_stats();
    if (ms_per_exchange) *ms_per_exchange = s.exchanges_timed > 0 ? s.exchange_ms / s.exchanges_timed : 0.0;
    if (exchanges_timed) *exchanges_timed = s.exchanges_timed;
    if (backend) *backend = s.exchange_backend;
    return NMFX_OK;
}
nmfx_status nmfx_last_call_timing(double *ingest_s, double *iterate_s, double *egress_s, double *host_bytes_in, double *host_bytes_out) {
    const nmfx::IoStats &s = nmfx::io_stats();
    if (ingest_s) *ingest_s = s.ingest_s;
    if (iterate_s) *iterate_s = s.iterate_s;
    if (egress_s) *egress_s = s.egress_s;
    if (host_bytes_in) *host_bytes_in = s.h2d_bytes_host;
    if (host_bytes_out) *host_bytes_out = s.d2h_bytes_host;
    return NMFX_OK;
}

}  // extern "C"

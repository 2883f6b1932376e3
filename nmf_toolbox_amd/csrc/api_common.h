// Shared by the translation units behind the C ABI (engine.hip, host_io.hip, blocking.hip, sc.hip): device guard, workspace carver,
// hipEvent profiler, the engine object, host <-> device staging.
#pragma once
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <string>
#include <vector>

#include "nmfx_internal.h"

namespace nmfx {

nmfx_status check_device(int device);

// every entry point leaves the caller's current HIP device as it found it (torch and MATLAB hosts keep their own idea of "current")
struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); } }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

static inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

struct Carver {
    char *base;
    size_t off;
    explicit Carver(void *b) : base(static_cast<char *>(b)), off(0) {}
    template <class T> T *take(size_t count) {
        T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
        off += al256(sizeof(T) * count);
        return p;
    }
};

// grid.y of a fused pass over `blocks` 128-row blocks (engine.hip)
int fused_split(long blocks, long extent, int K, long *c_per_split);

}  // namespace nmfx


struct ProfEvent {
    int tag;
    hipEvent_t a, b;
};
// hipEvent pairs around launch groups, on the stream the kernels run on (bench.py's per-kernel durations)
struct Profiler {
    bool on = false;
    unsigned skip_mask = 0;    // coarse mode: launch groups whose tag bit is set (small kernels, K x K products) are not bracketed -- an event pair costs ~5 us of
                               // stream time, 14 pairs per iteration are 4 % of a 2.2 ms iteration on an 8-GPU shard
    hipStream_t st = nullptr;
    std::vector<ProfEvent> events;
    std::vector<hipEvent_t> pool;
    size_t pool_used = 0;
    void enable(bool e) { on = e; events.clear(); pool_used = 0; }
    void release() { for (hipEvent_t ev : pool) (void)hipEventDestroy(ev); pool.clear(); events.clear(); pool_used = 0; }
    nmfx_status read(int ntags, double *ms_per_tag, int32_t *count_per_tag) const {
        for (int t = 0; t < ntags; ++t) { ms_per_tag[t] = 0.0; count_per_tag[t] = 0; }
        for (const ProfEvent &pe : events) {
            float ms = 0.f;
            NMFX_HIP(hipEventElapsedTime(&ms, pe.a, pe.b));
            if (pe.tag >= 0 && pe.tag < ntags) { ms_per_tag[pe.tag] += ms; count_per_tag[pe.tag] += 1; }
        }
        return NMFX_OK;
    }
};
struct PScope {
    Profiler *p;
    int idx;
    PScope(Profiler *p_, int tag) : p(p_), idx(-1) {
        if (!p || !p->on || (tag >= 0 && tag < 32 && ((p->skip_mask >> tag) & 1u))) return;
        auto get = [&]() {
            if (p->pool_used == p->pool.size()) {
                hipEvent_t ev;
                if (hipEventCreate(&ev) != hipSuccess) return (hipEvent_t) nullptr;
                p->pool.push_back(ev);
            }
            return p->pool[p->pool_used++];
        };
        ProfEvent pe{tag, get(), get()};
        if (!pe.a || !pe.b) return;
        (void)hipEventRecord(pe.a, p->st);
        p->events.push_back(pe);
        idx = (int)p->events.size() - 1;
    }
    ~PScope() {
        if (idx >= 0) (void)hipEventRecord(p->events[idx].b, p->st);
    }
};

struct nmfx_engine {
    long m, n;                // n = local columns owned by this shard
    int hL, hR;               // halo columns of H on each side; V / V_hat carry hR extra columns on the right
    long nvalid;              // columns of V that exist globally (<= n + hR)
    float *Hext;              // base of the K x (hL + n + hR) buffer; H points at its centre
    int K, T, KT, div, algo;
    int K_valid;              // components k >= K_valid are zero padding (0 = none)
    double alpha, beta;       // NMFX_DIV_AB only; alpha == 0 selects the dual update equations (nmf.m:124-128)
    int device;
    hipStream_t st;
    const float *V;
    float *W, *H, *packed;
    // float64 master copies (workspace): the state between iterations is double, as in the reference (nmf.m:168-169,199); W / H are their fp32 images, the
    // operands of the MFMA passes and the caller's results.  H64 covers the shard's own columns (halos are read-only operands); nullptr for constrainednmf's
    // H (a gather of Z).  P64: W*(H*H') of the euclidean Gram paths in float64 (gemm64.hip)
    double *W64, *H64, *P64;
    int rank0;
    bool any_lamW, any_lamH;
    // workspace
    float *Vhat, *Gn, *Gp, *gemm_scratch;
    size_t gemm_scratch_bytes;
    float *lamW, *lamH;
    uint8_t *fixW, *fixH;
    bool all_fixW, all_fixH;
    double *sumsq, *f_out, *rowsum, *colsum, *Pvec, *Gpvec, *cost_partials, *cost, *l1W, *l1H;
    void *rr_scratch;
    int n_cost_partials, n_cost_used;
    // fused path (fused.hip): V_hat is never materialised
    bool fused, cost_valid, defer_hfinish;
    bool cost_dst2_done;      // set by the finisher that honoured cost_dst2
    double *cost_dst2;        // fused paths: the finisher of the next lagged cost also writes it here (the caller's cost vector), or nullptr
    bool tail_with_cost;      // fused KL: the finisher also converts rowsum(H) into the fp32 tail of `packed` (W-step partial passes only)
    bool dual;                // fused IS / alpha-beta: packed = [N | P], both contractions of a pass come out of one kernel (func 4 / 5)
    bool dualz;               // alpha-beta with alpha == 0 (the reference's dual update equations): a `dual2` engine whose passes are functor 17 (S -> S.^beta ./ V ->
                              // contraction) and functor 0 on V.^(beta-1) (kept in the Valpha slot); any K <= 256; the cost is the reference's +-Inf
    bool dual2;               // ... above K = 192 (a sub-mode of `dual`): the second accumulator set no longer fits, so every pass runs twice with ONE element map
                              // each (func 11 + 12 / 13 + 14): S = W*H is formed twice, V_hat still never reaches HBM
    float *slabs2, *Valpha;   // dual: slabs of the second contraction; alpha-beta with alpha ~= 1: V.^alpha (the kernels' data operand)
    double *sumVab;           // dual: the constant of the cost (IS: 0; alpha-beta: sum(V.^(alpha+beta)), nmf.m:214)
    bool gram;                // cnmf euclidean in Gram form: V_hat*Hs' = W_flat*(Hs*Hs'), sum_t W_t'*lshift(V_hat) from W_flat'*W_flat (no V_hat in HBM)
    float *CC;                // KT x KT Gram of the stacked W (gram path)
    // cnmf euclidean on the register-stationary kernels (fused_kernel TT > 1): numerator and cost passes with the shift-sum in LDS,
    // H-step numerator as ONE (KT x n x m) GEMM Q = W_flat' * V followed by the shift-sum over t
    bool fusedT, hpad_valid;
    bool dualw;               // IS / alpha-beta nmf with K > 256 (unsharded): klw's column-block chain for S, whose last block stores both element maps' values (functors
                              // 19 / 20; A in the V_hat buffer, B in Vhat2); numerator passes per column block on either; the H-step products as two GEMMs
    bool fusedT_dual;         // IS / alpha-beta cnmf (unsharded, the common (K, T) pairs): an S pass stores BOTH element maps' values (A in the V_hat buffer, B in Vhat2)
                              // and yields the cost of the state it starts from; the numerator passes and the H-step products contract those; V_hat itself is never formed
    float *Vhat2;
    bool fusedT_kl;           // KL cnmf on the fused passes: an S pass stores R = V./V_hat (in the V_hat buffer) and yields the cost of the state it
                              // started from (lagged, like the nmf fused path); the numerator passes then read R instead of V
    double *sumV_g, *colV_g;  // its closed-form cost term sum(V)
    bool klw;                 // KL nmf / lnmf / constrainednmf with K > 256 (nmf.m:152-153,183-184 have no K limit): V_hat is never formed either.  S = W*H is
                              // accumulated over column blocks of <= 256 components by the stationary kernel (functors 7 / 8, partial sums and then R = V./S in
                              // the V_hat buffer), the numerators R*H' run block by block on the same kernel, W'*R on the two-operand GEMM; the cost lags like
                              // the fused path's
    int klw_nb, klw_k0[8], klw_kb[8];   // its column blocks
    bool eucw;                // euclidean nmf / constrainednmf with K > 256 (a sub-mode of `gram`): the numerators V*H' and W'*V block by block on the stationary kernel (the
                              // latter over the transposed copy of V), the cost in Gram form out of the W update's column sums; the explicit residual behind it is the
                              // S chain of klw with functor 10.  Shares klw_nb / klw_k0 / klw_kb / klw_hsplit and the V' / W' / slab buffers
    bool klw_vt;              // ... with the H step on the transposed copy of V: R' = V'./(H'*W') straight from the same kernels, then (R'*W)' -- every V / R tile
                              // read along its contiguous dimension, no two-operand GEMM (needs V' and the W' copy)
    int klw_hsplit;
    long klw_hcps;
    float *slabsH;
    bool qgemm;               // cnmf, T > 1: H-step numerator sum_t W_t' * lshift_t(A) as ONE (KT x n x m) GEMM Q = W_flat' * A + a shift-sum over t
    float *Hpad, *Qbuf, *slabsT;
    int nsplit_T;
    long cps_T;
    // unsharded fused cnmf: the two Gram products that involve the stacked shifted H by LAG (aux.hip: gram_from_lags, lag_sum, gp_tail) --
    // T lag Grams instead of the T x T blocks of Hs*Hs', 2T-1 lag sums of W_flat'*W_flat instead of T^2 blocks in the H-step denominator
    bool lagram;
    float *Llag, *Elag;
    int nsplit_w, isplit_h;
    long cps_w, cps_h;        // streamed extent per split (multiples of 64; the last split may be shorter)
    int w_chunks;             // row chunks of the last W-step partial: packed = [chunk 0 (m/c x K) | chunk 1 | ... | tail]
    int chunk_parts;          // cost partials written by the chunks so far
    float *WT, *slabs, *GW;
    float *VT;                // euclidean fused path: V' (n x m), built once at init -- the H-step numerator W'*V runs as (V'*W)' on the W-step-form kernel
                              // (also IS / alpha-beta above K = 192, whose H step runs on it as 4 + 2 m*n*K: engine.hip, dual2)
    float *VTa;               // ... the transposed copy of V.^alpha next to it (alpha-beta with alpha ~= 1)
    bool use_vt;
    float *WTf;               // cnmf on the fused passes, euclidean: W_flat' (row i = its K*T floats), rebuilt before each Q product
    bool use_vtq;             // ... whose Q = W_flat'*V runs as (V'*W_flat)' on the W-step-form kernel, K-wide column blocks in grid.z
    int vtq_block;
    double *sumV, *colV;      // KL closed-form cost term: sum(V) (once) via per-column sums
    // euclidean cost in Gram form (fused nmf path): 0.5*||V - W*H||^2 = 0.5*||V||^2 - <W, V*H'> + 0.5*<W, W*(H*H')> -- every term is something the W
    // step computes anyway (the two inner products are the "diagonal" column sums of nmf.m:149-150), so the W-step pass needs no first product
    // W*H at all: half its MFMA work.  fp32 products resolve that difference of large numbers to the contract only while the residual is
    // not small against V (measured: absolute error <= 3e-9*||V||^2); a device-side flag, set once cost < GRAM_COST_RATIO_MIN * 0.5*||V||^2,
    // turns the explicit residual pass back on (conditional launch) -- deterministic, no host round trip, identical on every rank.
    bool gram_cost;           // the engine can run in this mode (euclidean, fused, W not all fixed)
    bool wstep_gram;          // the W step in flight runs in this mode (set by wstep_partial, read by wstep_finish)
    bool classic;             // host-side latch: the decision of two W updates ago had the flag set (or the caller chunks the W step): the one-pass kernel with the cost inside again
    unsigned decide_seq;      // gram_decide launches since init: decision s publishes (s+1) << 1 | flag into exact_flag_host[s & 7]
    bool p1gram;              // path 1 (materialised V_hat), euclidean nmf: the W-step denominators all the same as W*(H*H') in float64 (engine.hip::generic_wstep_partial)
    bool dist_seen, sumvv_global_set;   // column shards: the decision needs the GLOBAL ||V||^2 (nmfx_engine_sumvv_ptr); without it the mode stays off
    double *sumVV;            // device [2]: ||V_local||^2, ||V_global||^2
    double *dndp;             // device [2*K]: column sums dn = cs(W.*P), dp = cs(W.*N) of the last W update
    int *exact_flag;          // device
    int *exact_flag_host;     // host-mapped, 16 ints (owned: hipHostMalloc): slots [0, 8) receive the stamped decisions
    // constrainednmf (algo 3): H = Z*A with A the 0/1 label matrix of label-sorted samples; segment c = columns [seg[c], seg[c+1])
    float *Z;
    long nz;
    long *seg_dev;            // owned (hipMalloc) -- the only allocation the engine makes itself
    Profiler prof;
};

namespace nmfx {

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { release(); }
    void release() { if (p) { (void)hipFree(p); p = nullptr; } }
    nmfx_status alloc(size_t bytes) {
        hipError_t e = hipMalloc(&p, bytes ? bytes : 256);
        if (e != hipSuccess) { set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); p = nullptr; return NMFX_ERR_NOMEM; }
        return NMFX_OK;
    }
    template <class T> T *as() { return static_cast<T *>(p); }
};

inline size_t dsize(int dtype) { return dtype == NMFX_F64 ? 8 : 4; }

// host (f32 / f64, pageable) <-> device fp32 through two pinned staging buffers, conversion on host threads (host_io.hip); upload: out = in / divide_by
nmfx_status upload(hipStream_t st, const void *host, int dtype, float *dev, size_t count, double divide_by);
nmfx_status download(hipStream_t st, const float *dev, int dtype, void *host, size_t count);
void host_minmax(const void *host, int dtype, size_t count, double *vmin, double *vmax);
void staging_quiesce();   // the pinned staging buffers hold no reference to an event of a stream that is about to be handed back (host_io.hip)
// per-thread account of the last blocking call (nmfx_last_call_timing)
struct IoStats { double ingest_s = 0, iterate_s = 0, egress_s = 0, h2d_bytes_host = 0, h2d_bytes_pcie = 0, d2h_bytes_host = 0;
                 double exchange_ms = 0; int exchanges_timed = 0, exchange_backend = 0; };   // multi-GPU calls: the packed exchange (nmfx_last_call_exchange)
IoStats &io_stats();
nmfx_status validate_problem(const nmfx_problem *p, const nmfx_result *r, bool nmfsc, bool need_H_init = true);
// Streams and events of the single-process multi-GPU drivers come out of a process-wide pool and go back to it, never destroyed: a MATLAB session calls
// nmf() many times, and creating / destroying 8 streams + 32 events per call at a high call rate is what a rare host-heap corruption inside the runtime's
// teardown went with (scripts/fuzz_campaign_r3.py multi_edge).  Callers drain a stream before they hand it back.
nmfx_status pool_stream(int device, hipStream_t *st);
nmfx_status pool_event(int device, hipEvent_t *ev);
void unpool_stream(int device, hipStream_t st);
void unpool_event(int device, hipEvent_t ev);
nmfx_status pool_event_timed(int device, hipEvent_t *ev);   // the same with timing enabled (hipEventElapsedTime)
void unpool_event_timed(int device, hipEvent_t ev);
nmfx_status run_nmfsc_multi(const nmfx_problem *p, nmfx_result *r);   // multi_sc.hip
// RCCL behind the blocking multi-GPU calls, dlopen'ed (rccl_backend.hip)
bool rccl_usable(const int *devs, int n, std::string *why);
nmfx_status rccl_comms(const int *devs, int n, void **comms_out);   // checks the cached set out for the calling thread ...
void rccl_release(const int *devs, int n);                          // ... until this
nmfx_status rccl_allreduce_f32(void *const *comms, const int *devs, hipStream_t const *streams, float *const *bufs, int n, size_t count);
bool nmfsc_f64_eligible(const nmfx_problem *p);                       // sc64.hip: small problems run nmfsc.m in float64 end to end
nmfx_status run_nmfsc_f64(const nmfx_problem *p, nmfx_result *r);
void sc_thread_cleanup();                                              // sc.hip
void sc_hooks_reset();                                                 // sc.hip: empties this thread's nmfx_sc_iteration_seconds / nmfx_nmfsc_profile_read records
void sc_hooks_iteration_done(std::chrono::steady_clock::time_point since);

}  // namespace nmfx

// C ABI of libnmfx (include/nmfx.h): error plumbing, the device-resident engine (phases of one
// multiplicative-update iteration) and the blocking host-buffer entry points that the MEX gateway /
// Python ctypes wrapper bind.  Kernels live in gemm.hip / fused.hip / aux.hip / projfunc.hip.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "nmfx_internal.h"

namespace nmfx {

static thread_local char g_err[1024] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static nmfx_status check_device(int device) {
    int cnt = 0;
    hipError_t e = hipGetDeviceCount(&cnt);
    if (e != hipSuccess || cnt <= 0) {
        set_error("nmfx: no usable HIP device (hipGetDeviceCount: %s). There is no CPU fallback.", hipGetErrorString(e));
        (void)hipGetLastError();
        return NMFX_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= cnt) {
        set_error("nmfx: device %d out of range (have %d)", device, cnt);
        return NMFX_ERR_INVALID;
    }
    NMFX_HIP(hipSetDevice(device));
    return NMFX_OK;
}

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

}  // namespace nmfx

using namespace nmfx;

// =============================================================================================
// engine
// =============================================================================================
struct ProfEvent {
    int tag;
    hipEvent_t a, b;
};
// hipEvent pairs around launch groups, on the stream the kernels run on (bench.py's per-kernel durations)
struct Profiler {
    bool on = false;
    int skip_tag = -1;         // coarse mode: launch groups with this tag (the small kernels) are not bracketed -- an event pair costs ~5 us of
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
        if (!p || !p->on || tag == p->skip_tag) return;
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
    float *slabs2, *Valpha;   // dual: slabs of the second contraction; alpha-beta with alpha ~= 1: V.^alpha (the kernels' data operand)
    double *sumVab;           // dual: the constant of the cost (IS: 0; alpha-beta: sum(V.^(alpha+beta)), nmf.m:214)
    bool gram;                // cnmf euclidean in Gram form: V_hat*Hs' = W_flat*(Hs*Hs'), sum_t W_t'*lshift(V_hat) from W_flat'*W_flat (no V_hat in HBM)
    float *CC;                // KT x KT Gram of the stacked W (gram path)
    // cnmf euclidean on the register-stationary kernels (fused_kernel TT > 1): numerator and cost passes with the shift-sum in LDS,
    // H-step numerator as ONE (KT x n x m) GEMM Q = W_flat' * V followed by the shift-sum over t
    bool fusedT, hpad_valid;
    bool fusedT_kl;           // KL cnmf on the fused passes: an S pass stores R = V./V_hat (in the V_hat buffer) and yields the cost of the state it
                              // started from (lagged, like the nmf fused path); the numerator passes then read R instead of V
    double *sumV_g, *colV_g;  // its closed-form cost term sum(V)
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
    float *WT, *slabs, *Pbuf, *GW;
    float *VT;                // euclidean fused path: V' (n x m), built once at init -- the H-step numerator W'*V runs as (V'*W)' on the W-step-form kernel
    bool use_vt;
    float *WTf;               // cnmf on the fused passes, euclidean: W_flat' (row i = its K*T floats), rebuilt before each Q product
    bool use_vtq;             // ... whose Q = W_flat'*V runs as (V'*W_flat)' on the W-step-form kernel, K-wide column blocks in grid.z
    int vtq_block;
    double *sumV, *colV;      // KL closed-form cost term: sum(V) (once) via per-column sums
    // constrainednmf (algo 3): H = Z*A with A the 0/1 label matrix of label-sorted samples; segment c = columns [seg[c], seg[c+1])
    float *Z;
    long nz;
    long *seg_dev;            // owned (hipMalloc) -- the only allocation the engine makes itself
    Profiler prof;
};

enum ProfTag { TAG_RECON = 0, TAG_WNUM = 1, TAG_WDEN = 2, TAG_HNUM = 3, TAG_HDEN = 4, TAG_RECON_COST = 5, TAG_SMALL = 6,
               TAG_FUSED_W = 7, TAG_FUSED_H = 8, TAG_FUSED_COST = 9, TAG_GRAM = 10, TAG_COUNT = 11 };
static const char *const kTagNames[TAG_COUNT] = {"gemm:V_hat=W*H", "gemm:N=A*H'", "gemm:P=B*H'", "H-step numerator Gn=W'*A (two-operand GEMM, or the stationary kernel over V')", "gemm:Gp=W'*B",
                                                 "gemm:V_hat=W*H+cost", "small kernels", "fused:W-step (S=W*H -> R -> R*H')",
                                                 "fused:H-step (S=W*H -> R -> W'*R + update)", "fused:cost pass (S=W*H -> D(V||S))",
                                                 "gemm:Gram/K x K products"};

namespace {

struct Scope : PScope {
    Scope(nmfx_engine *e, int tag) : PScope(&e->prof, tag) {}
};

struct Layout {
    size_t total;
    size_t packed_count;
};

bool div_has_matrix_den(int div) { return div != NMFX_DIV_KL; }
int fused_split(long blocks, long extent, int K, long *c_per_split);

// carve (or just size, when ws == nullptr) the workspace
Layout layout(nmfx_engine *e, void *ws) {
    Carver c(ws);
    const size_t mn = (size_t)e->m * (e->n + e->hR), Kn = (size_t)e->K * e->n, mKT = (size_t)e->m * e->KT;
    e->Vhat = c.take<float>(mn);
    e->Gn = c.take<float>(Kn);
    e->Gp = div_has_matrix_den(e->div) ? c.take<float>(Kn) : nullptr;
    size_t gs = gemm_scratch_bytes(e->m, e->KT, e->n);
    size_t gs2 = gemm_scratch_bytes(e->K, e->n, (long)e->T * e->m);
    size_t gs3 = gemm_scratch_bytes(e->m, e->n + e->hR, e->KT);
    size_t gs4 = e->qgemm ? gemm_scratch_bytes(e->KT, e->n + e->hR, e->m) : 0;
    if (gs2 > gs) gs = gs2;
    if (gs3 > gs) gs = gs3;
    if (gs4 > gs) gs = gs4;
    e->gemm_scratch_bytes = gs;
    e->gemm_scratch = gs ? c.take<float>(gs / sizeof(float)) : nullptr;
    e->lamW = c.take<float>(e->K);
    e->lamH = c.take<float>(e->K);
    e->fixW = c.take<uint8_t>(e->K);
    e->fixH = c.take<uint8_t>(e->K);
    e->sumsq = c.take<double>(e->KT);
    e->f_out = c.take<double>(e->K);
    e->rowsum = c.take<double>(e->K);
    e->colsum = c.take<double>(e->KT);
    e->Pvec = c.take<double>(e->KT);
    e->Gpvec = c.take<double>(e->K);
    e->l1W = c.take<double>(e->KT);
    e->l1H = c.take<double>(e->K);
    e->cost = c.take<double>(4);
    e->n_cost_partials = (int)gemm_grid_blocks(e->m, e->n + e->hR);
    e->cost_partials = c.take<double>(e->n_cost_partials);
    e->rr_scratch = c.take<char>(row_reduce_scratch_bytes(e->K));
    Layout L;
    if (e->fused) {
        // V_hat, Gn/Gp of the generic path are not needed: rewind and carve the fused buffers instead
        Carver f(ws);
        e->Vhat = nullptr;
        e->WT = f.take<float>(mKT);
        // row-chunked W steps use more splits on fewer rows: rows*split per launch never exceeds max(nsplit_w, 2) * m / 2
        e->slabs = f.take<float>(std::max((size_t)std::max(e->nsplit_w, 2) * mKT, (size_t)e->isplit_h * Kn));
        e->slabs2 = e->dual ? f.take<float>(std::max((size_t)std::max(e->nsplit_w, 2) * mKT, (size_t)e->isplit_h * Kn)) : nullptr;
        e->Valpha = (e->dual && e->div == NMFX_DIV_AB && e->alpha != 1.0) ? f.take<float>((size_t)e->m * e->n) : nullptr;
        e->VT = e->use_vt ? f.take<float>((size_t)e->m * e->n) : nullptr;
        e->Gn = f.take<float>(Kn);
        const bool euc = e->div == NMFX_DIV_EUCLIDEAN;
        e->Gp = (euc || e->dual) ? f.take<float>(Kn) : nullptr;
        e->Pbuf = euc ? f.take<float>(mKT) : nullptr;
        e->GW = euc ? f.take<float>((size_t)e->K * e->K) : nullptr;
        size_t g1 = gemm_scratch_bytes(e->K, e->K, e->n), g2 = gemm_scratch_bytes(e->K, e->K, e->m), g3 = gemm_scratch_bytes(e->K, e->n, e->m);
        e->gemm_scratch_bytes = euc ? std::max(std::max(g1, g2), g3) : 0;
        e->gemm_scratch = e->gemm_scratch_bytes ? f.take<float>(e->gemm_scratch_bytes / sizeof(float)) : nullptr;
        e->lamW = f.take<float>(e->K); e->lamH = f.take<float>(e->K);
        e->fixW = f.take<uint8_t>(e->K); e->fixH = f.take<uint8_t>(e->K);
        e->sumsq = f.take<double>(e->KT); e->f_out = f.take<double>(e->K); e->rowsum = f.take<double>(e->K);
        e->colsum = f.take<double>(e->KT); e->Pvec = f.take<double>(e->KT); e->Gpvec = f.take<double>(e->K);
        e->l1W = f.take<double>(e->KT); e->l1H = f.take<double>(e->K); e->cost = f.take<double>(4);
        e->n_cost_partials = (int)((e->m + 127) / 128) + 8 * 1024;   // any decomposition of the W-step pass into <= 8 row chunks (blocks*split < 1024 each, or = blocks)
        e->cost_partials = f.take<double>(e->n_cost_partials);
        e->rr_scratch = f.take<char>(row_reduce_scratch_bytes(e->K));
        e->sumV = f.take<double>(1);
        e->sumVab = f.take<double>(1);
        e->colV = f.take<double>(e->n);
        L.total = f.off;
        L.packed_count = e->dual ? 2 * mKT : (euc ? mKT + (size_t)e->K * e->K : mKT + (size_t)e->KT);
        return L;
    }
    if (e->gram) {
        e->Pbuf = c.take<float>(mKT);
        e->CC = c.take<float>((size_t)e->KT * e->KT);
        size_t g4 = gemm_scratch_bytes(e->KT, e->KT, e->n), g5 = gemm_scratch_bytes(e->KT, e->KT, e->m);
        size_t gg = std::max(std::max(g4, g5), sizeof(float) * Kn * e->T);   // + T slabs of the z-batched H-step denominator
        if (e->lagram) gg = std::max(gg, std::max(gemm_scratch_bytes(e->K, e->KT, e->n), gemm_scratch_bytes(e->K, e->n, (long)(2 * e->T - 1) * e->K)));
        if (gg > e->gemm_scratch_bytes) { e->gemm_scratch_bytes = gg; e->gemm_scratch = c.take<float>(gg / sizeof(float)); }
        if (e->fusedT) {
            e->Hpad = c.take<float>((size_t)e->K * (e->n + (e->lagram ? 2 : 1) * (e->T - 1)));
            if (e->lagram) {
                e->Llag = c.take<float>((size_t)e->K * e->KT);
                e->Elag = c.take<float>((size_t)e->K * (2 * e->T - 1) * e->K);
            }
            e->slabsT = e->nsplit_T > 1 ? c.take<float>((size_t)e->nsplit_T * mKT) : nullptr;
            const int need = (int)((e->m + 127) / 128) * e->nsplit_T;
            if (need > e->n_cost_partials) { e->n_cost_partials = need; e->cost_partials = c.take<double>(need); }
        }
    }
    if (e->qgemm) e->Qbuf = c.take<float>((size_t)e->KT * (e->n + e->hR));
    if (e->use_vtq) { e->VT = c.take<float>((size_t)e->m * e->n); e->WTf = c.take<float>(mKT); }
    if (e->fusedT_kl) {
        e->Hpad = c.take<float>((size_t)e->K * (e->n + e->T - 1));
        e->slabsT = e->nsplit_T > 1 ? c.take<float>((size_t)e->nsplit_T * mKT) : nullptr;
        const int need = (int)((e->m + 127) / 128) * e->nsplit_T;
        if (need > e->n_cost_partials) { e->n_cost_partials = need; e->cost_partials = c.take<double>(need); }
        e->sumV_g = c.take<double>(1);
        e->colV_g = c.take<double>(e->n);
    }
    L.total = c.off;
    L.packed_count = e->gram ? mKT + (size_t)e->KT * e->KT : (div_has_matrix_den(e->div) ? 2 * mKT : mKT + (size_t)e->KT);
    return L;
}

nmfx_status fill_from_desc(nmfx_engine *e, const nmfx_engine_desc *d) {
    if (!d || d->m <= 0 || d->n_local <= 0 || d->K_total <= 0 || d->T <= 0) {
        set_error("nmfx_engine: m, n_local, K_total, T must be positive");
        return NMFX_ERR_INVALID;
    }
    if (d->divergence == NMFX_DIV_AB && d->alpha == 0 && d->beta == 0) {   // nmf.m:120-122
        set_error("alpha = 0 and beta = 0 is not supported at this time.");
        return NMFX_ERR_INVALID;
    }
    if (d->divergence < 0 || d->divergence > NMFX_DIV_EUCLIDEAN_NOCOST) {
        set_error("nmfx_engine: unknown divergence %d", d->divergence);
        return NMFX_ERR_INVALID;
    }
    if (d->T > 1 && d->n_local < d->T) {
        set_error("nmfx_engine: context_len %d exceeds the number of columns %ld", d->T, (long)d->n_local);
        return NMFX_ERR_INVALID;
    }
    e->m = d->m;
    e->n = d->n_local;
    e->hL = d->halo_left; e->hR = d->halo_right;
    e->nvalid = (d->halo_left || d->halo_right) ? d->n_valid : d->n_local;
    if (e->hL < 0 || e->hR < 0 || e->nvalid < d->n_local || e->nvalid > d->n_local + d->halo_right) {
        set_error("nmfx_engine: inconsistent halo description");
        return NMFX_ERR_INVALID;
    }
    if ((e->hL || e->hR) && d->algorithm != 1) {
        set_error("nmfx_engine: halos are only meaningful for cnmf");
        return NMFX_ERR_INVALID;
    }
    e->K = d->K_total;
    e->K_valid = (d->K_valid > 0 && d->K_valid < d->K_total) ? d->K_valid : 0;
    if (e->K_valid && d->algorithm == 1) { set_error("nmfx_engine: K padding is not defined for cnmf"); return NMFX_ERR_INVALID; }
    e->T = d->T;
    e->KT = d->K_total * d->T;
    e->div = d->divergence;
    e->device = d->device;
    e->st = static_cast<hipStream_t>(d->stream);
    e->prof.st = e->st;
    e->rank0 = 1;
    e->algo = d->algorithm;
    e->alpha = d->divergence == NMFX_DIV_AB ? d->alpha : 1.0;
    e->beta = d->divergence == NMFX_DIV_AB ? d->beta : 1.0;
    if (e->algo < 0 || e->algo > 3) { set_error("nmfx_engine: unknown algorithm %d", e->algo); return NMFX_ERR_INVALID; }
    if (e->algo == 3 && e->div == NMFX_DIV_AB && e->alpha != 0) {
        // constrainednmf.m:229 `W' * V.^alpha .* V_hat.^(beta-1) * A'` multiplies a K x n by an m x n matrix element-wise: MATLAB
        // raises a dimension error there (unless K == m), so there is no reference behaviour to reproduce
        set_error("constrainednmf: the alpha-beta update with alpha ~= 0 is ill-formed in the reference (constrainednmf.m:229); use alpha = 0 (dual form), euclidean, kl or is");
        return NMFX_ERR_UNSUPPORTED;
    }
    if (e->algo != 1 && e->T != 1) {
        set_error("nmfx_engine: algorithms nmf / lnmf / constrainednmf require T == 1");
        return NMFX_ERR_INVALID;
    }
    if (e->algo == 2 && e->div != NMFX_DIV_KL) {
        set_error("nmfx_engine: lnmf is defined for the KL divergence only (lnmf.m:69,76,81)");
        return NMFX_ERR_INVALID;
    }
    // fused path eligibility: nmf rules, KL or euclidean, K a multiple of 32 up to 256, tileable shard
    // IS and alpha-beta (alpha ~= 0: the dual form has other equations) need two element maps and two accumulator sets per pass: K <= 128
    e->dual = (e->div == NMFX_DIV_IS || (e->div == NMFX_DIV_AB && e->alpha != 0)) && e->K <= 128;
    const bool eligible = (e->algo == 0 || e->algo == 2 || e->algo == 3) && e->T == 1 && (e->div == NMFX_DIV_KL || e->div == NMFX_DIV_EUCLIDEAN || e->dual) && fused_supported(e->K) &&
                          e->hL == 0 && e->hR == 0 && ((e->m >= 64 && e->n >= 64) || d->path == 2);   // ragged m / n: masked-edge kernels
    if (d->path == 2 && !eligible && e->algo != 1) {   // cnmf: see the fused shift-sum passes below
        set_error("nmfx_engine: fused path requested but the problem is not eligible (nmf / lnmf / constrainednmf rules, kl or euclidean, K a multiple of 32 up to 256)");
        return NMFX_ERR_UNSUPPORTED;
    }
    e->fused = eligible && d->path != 1;
    if (!e->fused) e->dual = false;
    static const bool no_vt = getenv("NMFX_NO_VT") != nullptr;   // dev switch (A/B runs): H-step numerator on the pipelined GEMM, no transposed copy of V
    // the transposed copy of V (euclidean paths) is a luxury: only where the device clearly has the room for it next to V
    // itself (V may or may not be allocated yet at this point: 2.5 x its size + 1 GiB must be free either way)
    bool room_vt = true;
    {
        size_t free_b = 0, total_b = 0;
        DeviceGuard dg_;
        if (hipSetDevice(d->device) == hipSuccess && hipMemGetInfo(&free_b, &total_b) == hipSuccess)
            room_vt = (double)free_b >= 2.5 * 4.0 * (double)e->m * (double)e->n + (double)(1ull << 30);
        (void)hipGetLastError();
    }
    e->use_vt = e->fused && e->div == NMFX_DIV_EUCLIDEAN && !no_vt && room_vt;
    // euclidean problems the register-stationary kernels do not take (cnmf; nmf / constrainednmf with K > 256 or tiny shapes) still never
    // materialise V_hat: denominators from Gram products, the cost from a store-less residual pass
    e->gram = !e->fused && (e->algo == 0 || e->algo == 1 || e->algo == 3) && (e->div == NMFX_DIV_EUCLIDEAN || e->div == NMFX_DIV_EUCLIDEAN_NOCOST) && d->path != 1;
    static const bool no_fusedT = getenv("NMFX_CNMF_NO_FUSED") != nullptr;   // dev switch: Gram form on the generic GEMM only (A/B runs)
    e->fusedT = e->gram && e->T > 1 && fused_supported_T(e->K, e->T) && e->m >= 64 && e->n >= 64 && (e->hL == 0 || e->hL >= e->T - 1) && !no_fusedT;
    e->fusedT_kl = !e->fused && e->algo == 1 && e->div == NMFX_DIV_KL && e->T > 1 && fused_supported_T(e->K, e->T) && e->m >= 64 && e->n >= 64 &&
                   e->hL == 0 && e->hR == 0 && d->path != 1 && !no_fusedT;
    if (d->path == 2 && e->algo == 1 && !e->fusedT && !e->fusedT_kl) {
        set_error("nmfx_engine: fused cnmf kernels requested but the problem is not eligible (euclidean or unsharded kl, T > 1, an instantiated (K, T) pair)");
        return NMFX_ERR_UNSUPPORTED;
    }
    if (e->fusedT || e->fusedT_kl) e->nsplit_T = fused_split((e->m + 127) / 128, e->n, e->KT, &e->cps_T);
    static const bool no_lagram = getenv("NMFX_CNMF_NO_LAGRAM") != nullptr;   // dev switch (A/B runs): T x T block Gram products
    e->lagram = e->fusedT && e->hL == 0 && e->hR == 0 && e->nvalid == e->n && e->n >= 2L * e->T && !no_lagram;
    static const bool no_qgemm = getenv("NMFX_CNMF_NO_QGEMM") != nullptr;   // dev switch (A/B runs)
    e->qgemm = !e->fused && e->algo == 1 && e->T > 1 && e->K % 4 == 0 && e->m % 4 == 0 && d->path != 1 && !no_qgemm;
    {   // euclidean cnmf on the fused passes, unsharded: the Q product of the H step on a transposed copy of V (see nmfx_engine_hstep)
        static const bool no_vt = getenv("NMFX_NO_VT") != nullptr;
        static const int vtq_env = getenv("NMFX_VTQ_BLOCK") ? atoi(getenv("NMFX_VTQ_BLOCK")) : 0;   // dev switch: 128 | 256
        e->vtq_block = vtq_env ? vtq_env : 128;   // C4 (K*T = 512): four 128-wide blocks, two workgroups per CU, 0.526 ms; two 256-wide blocks 0.549; the two-operand GEMM 0.585
        e->use_vtq = e->fusedT && e->qgemm && e->hL == 0 && e->hR == 0 && !no_vt && room_vt && e->KT % e->vtq_block == 0 && fused_supported(e->vtq_block);
    }
    e->nsplit_w = e->isplit_h = 1;
    if (e->fused) {
        e->nsplit_w = fused_split((e->m + 127) / 128, e->n, e->K, &e->cps_w);
        e->isplit_h = fused_split((e->n + 127) / 128, e->m, e->K, &e->cps_h);
    }
    return NMFX_OK;
}

inline int norm_mode(const nmfx_engine *e) { return e->algo == 3 ? 0 : e->algo; }   // w_normalize: 0 L2 columns, 1 cnmf slabs, 2 L1 (lnmf)
inline int mdiv(const nmfx_engine *e) { return e->div == NMFX_DIV_EUCLIDEAN_NOCOST ? NMFX_DIV_EUCLIDEAN : e->div; }

// element maps (V, V_hat) -> numerator operand A, denominator operand B   (nmf.m:149-156, cnmf.m:191-192)
void num_view(const nmfx_engine *e, OpView &v) {
    v.p = e->V;
    v.p2 = nullptr;
    v.func = NMFX_PRO_NONE;
    if (mdiv(e) == NMFX_DIV_KL) { v.p2 = e->Vhat; v.func = NMFX_PRO_RATIO; }
    if (mdiv(e) == NMFX_DIV_IS) { v.p2 = e->Vhat; v.func = NMFX_PRO_RATIO_SQ; }
    if (mdiv(e) == NMFX_DIV_AB) {   // nmf.m:159-163: V.^(a-1).*V_hat.^b (dual, a == 0)  |  V.^a.*V_hat.^(b-1)
        v.p2 = e->Vhat; v.func = NMFX_PRO_POWPROD;
        if (e->alpha == 0) { v.e1 = (float)(e->alpha - 1); v.e2 = (float)e->beta; }
        else { v.e1 = (float)e->alpha; v.e2 = (float)(e->beta - 1); }
    }
}
void den_view(const nmfx_engine *e, OpView &v) {
    v.p = e->Vhat;
    v.p2 = nullptr;
    v.func = NMFX_PRO_NONE;
    if (mdiv(e) == NMFX_DIV_IS) { v.p = e->Vhat; v.p2 = e->Vhat; v.func = NMFX_PRO_RECIP2; }
    if (mdiv(e) == NMFX_DIV_AB) {   // V.^(a+b-1) (dual)  |  V_hat.^(a+b-1)
        v.p = e->alpha == 0 ? e->V : e->Vhat; v.p2 = v.p; v.func = NMFX_PRO_POWPROD;
        v.e1 = (float)(e->alpha + e->beta - 1); v.e2 = 0.f;
    }
}
inline float outer_exp(const nmfx_engine *e) {   // the .^(1/alpha) (.^(1/beta) in the dual form) around both gradients, nmf.m:159-163
    if (mdiv(e) != NMFX_DIV_AB) return 1.0f;
    return (float)(1.0 / (e->alpha == 0 ? e->beta : e->alpha));
}

// V_hat = sum_t W_t * rshift_t(H)    (RFD.m:31 / 36-38) ; optionally fused with the cost reduction
nmfx_status recon(nmfx_engine *e, bool with_cost, bool store = true) {
    Scope s(e, with_cost ? TAG_RECON_COST : TAG_RECON);
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.M = e->m; g.N = e->n + e->hR; g.Kc = e->KT;   // V_hat also on the right-halo columns: the H step of the last T-1 local columns needs it
    g.A = OpView{e->W, nullptr, e->m, VIEW_RC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f};
    if (e->T == 1) g.B = OpView{e->H, nullptr, (long)e->K, VIEW_KC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f};
    else g.B = OpView{e->H, nullptr, (long)e->K, VIEW_HSTACK_KC, e->K, 0, 0, NMFX_PRO_NONE, 0.f, 0.f, e->hL};
    g.C = e->Vhat; g.ldc = e->m;
    g.cost_ncols = e->hR ? e->n : 0;
    g.splitk = 1;
    if (with_cost) {
        g.epi = EPI_COST; g.store_c = store ? 1 : 0; g.cost_div = mdiv(e); g.Vref = e->V; g.ldv = e->m; g.cost_partials = e->cost_partials;
        g.cost_alpha = (float)e->alpha; g.cost_beta = (float)e->beta;
        long blocks = 0;
        nmfx_status rc = launch_gemm(e->st, g, &blocks);
        e->n_cost_used = (int)blocks;
        return rc;
    }
    g.epi = EPI_STORE;
    return launch_gemm(e->st, g);
}

// out (m x KT) = X * H_stack'   with X given by view x   (nmf.m:149 V*H', cnmf.m:191)
nmfx_status x_times_ht(nmfx_engine *e, OpView x, float *out, int tag) {
    Scope s(e, tag);
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.M = e->m; g.N = e->KT; g.Kc = e->n;
    x.ld = e->m; x.mode = VIEW_RC; x.blk = 0; x.tstride = 0; x.lim = 0;
    g.A = x;
    if (e->T == 1) g.B = OpView{e->H, nullptr, (long)e->K, VIEW_RC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f};
    else g.B = OpView{e->H, nullptr, (long)e->K, VIEW_HSTACK_RC, e->K, 0, 0, NMFX_PRO_NONE, 0.f, 0.f, e->hL};
    g.C = out; g.ldc = e->m; g.epi = EPI_STORE; g.splitk = 1;
    return gemm_auto(e->st, g, e->gemm_scratch, e->gemm_scratch_bytes);
}

// out (K x n) = sum_t W_t' * lshift_t(X)    (nmf.m:180 W'*V, cnmf.m:217-226)
nmfx_status wt_times_x(nmfx_engine *e, OpView x, float *out, int tag) {
    Scope s(e, tag);
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.M = e->K; g.N = e->n; g.Kc = (long)e->T * e->m;
    if (e->T == 1) {
        g.A = OpView{e->W, nullptr, e->m, VIEW_KC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f};
        x.ld = e->m; x.mode = VIEW_KC; x.blk = 0; x.tstride = 0; x.lim = 0;
    } else {
        g.A = OpView{e->W, nullptr, e->m, VIEW_WSTACK_KC, (int)e->m, e->m * e->K, 0, NMFX_PRO_NONE, 0.f, 0.f};
        x.ld = e->m; x.mode = VIEW_XSHIFT_KC; x.blk = (int)e->m; x.tstride = 0; x.lim = (int)e->nvalid;
    }
    g.B = x;
    g.C = out; g.ldc = e->K; g.epi = EPI_STORE; g.splitk = 1;
    return gemm_auto(e->st, g, e->gemm_scratch, e->gemm_scratch_bytes);
}

#define TRY(x) do { nmfx_status s_ = (x); if (s_ != NMFX_OK) return s_; } while (0)

// C (M x N) = A (M x Kc) * B (Kc x N) with plain views; small K x K products of the euclidean Gram form
nmfx_status small_gemm(nmfx_engine *e, long M, long N, long Kc, OpView A, OpView B, float *C, long ldc) {
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.M = M; g.N = N; g.Kc = Kc; g.A = A; g.B = B; g.C = C; g.ldc = ldc; g.epi = EPI_STORE; g.splitk = 1;
    return gemm_auto(e->st, g, e->gemm_scratch, e->gemm_scratch_bytes);
}

nmfx_status cost_from_partials(nmfx_engine *e, int nparts, bool kl_closed_form = false) {
    const bool useW = e->any_lamW && e->rank0, useH = e->any_lamH;
    if (e->cost_dst2) e->cost_dst2_done = true;
    if (useW) TRY(col_reduce(e->st, e->W, e->m, e->m, e->KT, 2, e->l1W));
    if (useH) {   // constrainednmf.m:251 charges Z_sparsity on |Z|, not on H = Z*A
        if (e->algo == 3) TRY(row_reduce(e->st, e->Z, e->K, e->K, e->nz, 2, e->l1H, e->rr_scratch));
        else TRY(row_reduce(e->st, e->H, e->K, e->K, e->n, 2, e->l1H, e->rr_scratch));
    }
    double scale = mdiv(e) == NMFX_DIV_EUCLIDEAN ? 0.5 : 1.0;
    if (mdiv(e) == NMFX_DIV_AB) scale = -1.0 / (e->alpha * e->beta);   // nmf.m:214
    if (e->fused && e->dual) {
        // fused IS: partials hold sum(V./V_hat - log(V./V_hat)); nmf.m:212 subtracts 1 per element.  Fused alpha-beta: partials hold
        // sum(V.^a.*V_hat.^b - b/(a+b)*V_hat.^(a+b)); nmf.m:214 subtracts (a*sum(V.^(a+b)) + b*m*n) / (a+b) inside the scaled sum
        const double cnt = (double)e->m * (double)e->n;
        double pa = 0.0, pb = -cnt;
        if (mdiv(e) == NMFX_DIV_AB) {
            const double ab = e->alpha + e->beta;
            if (ab != 0) { pa = -e->alpha / ab; pb = -e->beta * cnt / ab; }
            else { pa = 0.0; pb = -((e->alpha + 2.0 * e->beta) * cnt) / ab; }   // nmf.m:214 divides by alpha + beta: +-Inf cost, like the reference
        }
        return finish_cost(e->st, e->cost_partials, nparts, scale, useW ? e->l1W : nullptr, e->KT, e->lamW, useH ? e->l1H : nullptr, e->K,
                           e->lamH, e->cost, nullptr, nullptr, 0, nullptr, mdiv(e) == NMFX_DIV_AB ? e->sumVab : nullptr, pa, pb, e->cost_dst2);
    }
    // fused KL: partials hold sum V.*log(V./V_hat); sum(V_hat) - sum(V) = sum_k colsum(W)_k * rowsum(H_local)_k - sum(V_local)
    const bool tail = e->fused && kl_closed_form && e->tail_with_cost;
    return finish_cost(e->st, e->cost_partials, nparts, scale, useW ? e->l1W : nullptr, e->KT, e->lamW, useH ? e->l1H : nullptr, e->K,
                       e->lamH, e->cost, kl_closed_form ? e->Gpvec : nullptr, e->rowsum, e->K, e->sumV, nullptr, 0.0, 0.0,
                       e->cost_dst2, tail ? e->rowsum : nullptr, tail ? e->packed + (size_t)e->m * e->KT : nullptr, e->K);
}

// grid.y of a fused pass over `blocks` 128-row blocks: enough workgroups for 256 CUs while every slice keeps whole 64-column tiles
int fused_split(long blocks, long extent, int K, long *c_per_split) {
    const long target = K <= 128 ? 512 : 256;   // K <= 128 kernels fit two workgroups per CU
    const long tiles = (extent + 63) / 64;
    long s = 1;
    while (blocks * s < target && s * 2 <= tiles) s *= 2;   // every split keeps at least one 64-wide tile
    const long per = (tiles + s - 1) / s;                   // tiles per split; trailing splits that would be empty are dropped
    *c_per_split = per * 64;
    return (int)((tiles + per - 1) / per);
}

// fused W-step pass (K2) or cost-only pass over rows [row0, row0 + rows) of the local shard.  N of those rows goes to `out`
// as a contiguous rows x K block; cost partials are appended at e->chunk_parts.
nmfx_status fused_wpass_rows(nmfx_engine *e, bool do_g2, long row0, long rows, float *out) {
    long cps = 0;
    const long blocks = (rows + 127) / 128;
    const int split = fused_split(blocks, e->n, e->K, &cps);
    if ((size_t)split * rows * e->K > (size_t)std::max(e->nsplit_w, 2) * e->m * e->K || e->chunk_parts + blocks * split > e->n_cost_partials) {
        set_error("fused W-step: row chunk too small for the workspace");
        return NMFX_ERR_INVALID;
    }
    FusedParams f;
    memset(&f, 0, sizeof(f));
    f.X = e->W + row0; f.xs_r = 1; f.xs_k = e->m;
    f.Y = e->H; f.D = e->V + row0; f.ldd = e->m; f.R = rows; f.Cn = e->n; f.K = e->K;
    f.c_per_split = cps;
    f.out = split == 1 ? out : e->slabs;
    f.slab_stride = rows * (long)e->K; f.os_r = 1; f.os_k = rows;
    f.cost_partials = e->cost_partials + e->chunk_parts;
    int func = e->div == NMFX_DIV_KL ? 3 : 1;
    float *out2 = nullptr;
    if (e->dual) {   // IS / alpha-beta: the denominators come out of the same pass, into the second half of `packed`
        if (rows != e->m) { set_error("fused IS / alpha-beta W step: row chunks are not supported"); return NMFX_ERR_UNSUPPORTED; }
        func = mdiv(e) == NMFX_DIV_IS ? 4 : 5;
        out2 = out + (size_t)e->m * e->K;
        f.out2 = split == 1 ? out2 : e->slabs2;
        f.ab_alpha = (float)e->alpha; f.ab_beta = (float)e->beta; f.inv_exp = 1.0f;
        if (e->Valpha) f.D = e->Valpha + row0;
    }
    {
        Scope s(e, do_g2 ? TAG_FUSED_W : TAG_FUSED_COST);
        TRY(launch_fused(e->st, f, split, true, func, do_g2, 0));
    }
    e->chunk_parts += (int)(blocks * split);
    if (do_g2 && split > 1) {
        Scope s(e, TAG_SMALL);
        TRY(reduce_slabs(e->st, e->slabs, split, f.slab_stride, f.slab_stride, out, 0));
        if (e->dual) TRY(reduce_slabs(e->st, e->slabs2, split, f.slab_stride, f.slab_stride, out2, 0));
    }
    return NMFX_OK;
}
// after the last row chunk: rowsum(H) (KL: also the W-step denominator, nmf.m:153) and the cost of the CURRENT (W, H)
nmfx_status fused_wpass_finish(nmfx_engine *e) {
    Scope s(e, TAG_SMALL);
    const bool kl = e->div == NMFX_DIV_KL;
    if (kl) TRY(row_reduce(e->st, e->H, e->K, e->K, e->n, 0, e->rowsum, e->rr_scratch));
    TRY(cost_from_partials(e, e->chunk_parts, kl));
    e->cost_valid = true;
    return NMFX_OK;
}
nmfx_status fused_wpass(nmfx_engine *e, bool do_g2) {
    e->chunk_parts = 0;
    e->tail_with_cost = do_g2;   // a W-step partial: the cost finisher also fills the fp32 tail [rowsum(H)] of `packed`
    TRY(fused_wpass_rows(e, do_g2, 0, e->m, e->packed));
    return fused_wpass_finish(e);
}

// cnmf fused passes (fused_kernel TT > 1) over the local columns: do_g2 -> N_all = V * H_stack' into `out` (m x KT), else the residual cost
// partials of the CURRENT (W, H).  H's T-1 columns to the left of the shard are its halo, or zeros (Hpad) on the first / only shard.
enum FusedTMode { FT_NUM = 0, FT_COST_EUC = 1, FT_S_KL = 2, FT_COST_KL = 3 };
nmfx_status ensure_hpad(nmfx_engine *e) {   // Hpad = [T-1 zero columns | H | T-1 zero columns (lag-form Gram products only)]
    if (e->hpad_valid) return NMFX_OK;      // H changed since the last pass (init, H step)
    Scope s(e, TAG_SMALL);
    TRY(pad_left(e->st, e->H, e->K, e->n, e->T - 1, e->Hpad, e->lagram ? e->T - 1 : 0));
    e->hpad_valid = true;
    return NMFX_OK;
}
nmfx_status fusedT_pass(nmfx_engine *e, int mode, float *out) {
    const bool do_g2 = mode == FT_NUM;
    const float *Hy = e->H;
    if (e->hL < e->T - 1) {
        TRY(ensure_hpad(e));
        Hy = e->Hpad + (size_t)e->K * (e->T - 1);
    }
    FusedParams f;
    memset(&f, 0, sizeof(f));
    f.X = e->W; f.xs_r = 1; f.xs_k = e->m; f.xs_t = e->m * (long)e->K; f.T = e->T;
    f.Y = Hy; f.ldd = e->m; f.R = e->m; f.Cn = e->n; f.K = e->KT;
    f.D = (mode == FT_NUM && e->fusedT_kl) ? e->Vhat : e->V;      // KL: the numerators contract R = V./V_hat (left in the V_hat buffer by the S pass)
    if (mode == FT_S_KL) f.Rout = e->Vhat;
    f.c_per_split = e->cps_T;
    const long mKT = e->m * (long)e->KT;
    f.out = e->nsplit_T == 1 ? out : e->slabsT;
    f.slab_stride = mKT; f.os_r = 1; f.os_k = e->m; f.os_t = e->m * (long)e->K;
    f.cost_partials = do_g2 ? nullptr : e->cost_partials;
    const int func = mode == FT_NUM ? 0 : (mode == FT_COST_EUC ? 1 : 3);
    {
        Scope s(e, do_g2 ? TAG_FUSED_W : TAG_FUSED_COST);
        TRY(launch_fused(e->st, f, e->nsplit_T, true, func, do_g2, 0));
    }
    if (do_g2 && e->nsplit_T > 1) {
        Scope s(e, TAG_SMALL);
        TRY(reduce_slabs(e->st, e->slabsT, e->nsplit_T, mKT, mKT, out, 0));
    }
    if (!do_g2) e->n_cost_used = (int)((e->m + 127) / 128) * e->nsplit_T;
    return NMFX_OK;
}
// KL cnmf on the fused passes: the cost of the CURRENT (W, H) from the S pass's partials, sum(V.*log(V./V_hat)), plus the closed form
// sum(V_hat) - sum(V) = sum_{t,k} colsum(W_t)_k * sum_{j < n-t} H(k, j) - sum(V)    (the rshift of RFD.m:37 drops the last t columns of H)
nmfx_status fusedT_kl_cost(nmfx_engine *e) {
    Scope s(e, TAG_SMALL);
    TRY(row_reduce(e->st, e->H, e->K, e->K, e->n, 0, e->rowsum, e->rr_scratch));
    TRY(kl_pvec(e->st, e->rowsum, e->H, e->K, e->n, e->T, e->Pvec, 0));
    TRY(col_reduce(e->st, e->W, e->m, e->m, e->KT, 0, e->colsum));
    const bool useW = e->any_lamW && e->rank0, useH = e->any_lamH;
    if (useW) TRY(col_reduce(e->st, e->W, e->m, e->m, e->KT, 2, e->l1W));
    if (useH) TRY(row_reduce(e->st, e->H, e->K, e->K, e->n, 2, e->l1H, e->rr_scratch));
    TRY(finish_cost(e->st, e->cost_partials, e->n_cost_used, 1.0, useW ? e->l1W : nullptr, e->KT, e->lamW, useH ? e->l1H : nullptr, e->K, e->lamH, e->cost,
                    e->colsum, e->Pvec, e->KT, e->sumV_g, nullptr, 0.0, 0.0, e->cost_dst2));
    e->cost_valid = true;
    return NMFX_OK;
}

nmfx_status refresh_w_derived(nmfx_engine *e, bool have_colsum = false) {   // W^T copy (streamed operand of the H step) + KL / Gram denominators
    TRY(transpose_f32(e->st, e->W, e->m, e->K, e->WT));
    if (e->div == NMFX_DIV_KL && !have_colsum) {   // (after a W update the update kernel has already left colsum(W) in Gpvec)
        TRY(col_reduce(e->st, e->W, e->m, e->m, e->K, 0, e->Gpvec));   // T == 1: colsum(W) is the H-step denominator as is
    }
    return NMFX_OK;
}

}  // namespace

extern "C" {

const char *nmfx_last_error(void) { return g_err; }
int32_t nmfx_version(void) { return NMFX_VERSION; }
int32_t nmfx_device_count(void) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return cnt;
}

nmfx_status nmfx_engine_workspace_bytes(const nmfx_engine_desc *d, size_t *bytes) {
    nmfx_engine tmp{};
    TRY(fill_from_desc(&tmp, d));
    *bytes = layout(&tmp, nullptr).total;
    return NMFX_OK;
}
nmfx_status nmfx_engine_packed_count(const nmfx_engine_desc *d, size_t *count) {
    nmfx_engine tmp{};
    TRY(fill_from_desc(&tmp, d));
    *count = layout(&tmp, nullptr).packed_count;
    return NMFX_OK;
}

nmfx_status nmfx_engine_create(const nmfx_engine_desc *d, const float *V, float *W, float *H, void *workspace, size_t workspace_bytes,
                               float *packed, nmfx_engine **out) {
    if (!out || !V || !W || !H || !workspace || !packed) { set_error("nmfx_engine_create: null pointer"); return NMFX_ERR_INVALID; }
    DeviceGuard dg_;
    TRY(check_device(d ? d->device : 0));
    nmfx_engine *e = new nmfx_engine{};
    nmfx_status s = fill_from_desc(e, d);
    if (s != NMFX_OK) { delete e; return s; }
    e->V = V; e->W = W; e->Hext = H; e->H = H + (size_t)e->K * e->hL; e->packed = packed;
    // the transposed copy of V is optional: it is used when the workspace the caller brought has the room for it (nmfx_engine_workspace_bytes
    // asks for it when the device looked roomy at that moment; a caller that allocated less simply gets the path without it)
    if ((e->use_vt || e->use_vtq) && layout(e, nullptr).total > workspace_bytes) e->use_vt = e->use_vtq = false;
    else if (!e->use_vt && !e->use_vtq) {   // ... and the other way round: memory looked tight now, but the workspace was sized with the copy
        nmfx_engine probe = *e;
        probe.use_vt = probe.fused && probe.div == NMFX_DIV_EUCLIDEAN && getenv("NMFX_NO_VT") == nullptr;
        probe.use_vtq = probe.fusedT && probe.qgemm && probe.hL == 0 && probe.hR == 0 && getenv("NMFX_NO_VT") == nullptr && probe.KT % probe.vtq_block == 0 && fused_supported(probe.vtq_block);
        if ((probe.use_vt || probe.use_vtq) && layout(&probe, nullptr).total <= workspace_bytes) { e->use_vt = probe.use_vt; e->use_vtq = probe.use_vtq; }
    }
    Layout L = layout(e, workspace);
    if (L.total > workspace_bytes) {
        set_error("nmfx_engine_create: workspace too small (%zu < %zu)", workspace_bytes, L.total);
        delete e;
        return NMFX_ERR_INVALID;
    }
    std::vector<float> lw(e->K, 0.f), lh(e->K, 0.f);
    std::vector<uint8_t> fw(e->K, 0), fh(e->K, 0);
    e->all_fixW = e->all_fixH = true;
    for (int k = 0; k < e->K; ++k) {
        if (d->lamW_col) lw[k] = d->lamW_col[k];
        if (d->lamH_row) lh[k] = d->lamH_row[k];
        if (d->fixW_col) fw[k] = d->fixW_col[k] ? 1 : 0;
        if (d->fixH_row) fh[k] = d->fixH_row[k] ? 1 : 0;
        e->any_lamW |= lw[k] != 0.f;
        e->any_lamH |= lh[k] != 0.f;
        e->all_fixW &= fw[k] != 0;
        e->all_fixH &= fh[k] != 0;
    }
    hipError_t he = hipMemcpyAsync(e->lamW, lw.data(), sizeof(float) * e->K, hipMemcpyHostToDevice, e->st);
    if (he == hipSuccess) he = hipMemcpyAsync(e->lamH, lh.data(), sizeof(float) * e->K, hipMemcpyHostToDevice, e->st);
    if (he == hipSuccess) he = hipMemcpyAsync(e->fixW, fw.data(), e->K, hipMemcpyHostToDevice, e->st);
    if (he == hipSuccess) he = hipMemcpyAsync(e->fixH, fh.data(), e->K, hipMemcpyHostToDevice, e->st);
    if (he == hipSuccess) he = hipStreamSynchronize(e->st);  // host vectors go out of scope
    if (he != hipSuccess) { set_error("nmfx_engine_create: %s", hipGetErrorString(he)); delete e; return NMFX_ERR_HIP; }
    *out = e;
    return NMFX_OK;
}

void nmfx_engine_destroy(nmfx_engine *e) {
    if (!e) return;
    if (e->seg_dev) (void)hipFree(e->seg_dev);
    e->prof.release();
    delete e;
}

// constrainednmf (algorithm 3): segments of label-sorted columns and the device cluster matrix Z (K x nz, column-major).
// seg_host[0] = 0 < seg_host[1] < ... < seg_host[nz] = n_local; call before nmfx_engine_init.
nmfx_status nmfx_engine_set_constraint(nmfx_engine *e, const int64_t *seg_host, int64_t nz, float *Z_dev) {
    if (!e || e->algo != 3) { set_error("nmfx_engine_set_constraint: engine was not created with algorithm 3"); return NMFX_ERR_INVALID; }
    if (!seg_host || !Z_dev || nz <= 0 || nz > e->n) { set_error("nmfx_engine_set_constraint: bad arguments"); return NMFX_ERR_INVALID; }
    if (seg_host[0] != 0 || seg_host[nz] != e->n) { set_error("nmfx_engine_set_constraint: segments must cover [0, n)"); return NMFX_ERR_INVALID; }
    for (int64_t c = 0; c < nz; ++c)
        if (seg_host[c + 1] <= seg_host[c]) { set_error("nmfx_engine_set_constraint: empty segment %ld", (long)c); return NMFX_ERR_INVALID; }
    DeviceGuard dg_;
    NMFX_HIP(hipSetDevice(e->device));
    if (e->seg_dev) { (void)hipFree(e->seg_dev); e->seg_dev = nullptr; }
    std::vector<long> sg(seg_host, seg_host + nz + 1);
    NMFX_HIP(hipMalloc(&e->seg_dev, sizeof(long) * (nz + 1)));
    NMFX_HIP(hipMemcpyAsync(e->seg_dev, sg.data(), sizeof(long) * (nz + 1), hipMemcpyHostToDevice, e->st));
    NMFX_HIP(hipStreamSynchronize(e->st));
    e->Z = Z_dev; e->nz = nz;
    return NMFX_OK;
}

nmfx_status nmfx_engine_set_rank0(nmfx_engine *e, int32_t is_rank0) { e->rank0 = is_rank0; return NMFX_OK; }

// nmf.m:130-139 / cnmf.m:155-171: normalise W (all sources, fixed or not), cnmf also rescales H; then V_hat
nmfx_status nmfx_engine_init(nmfx_engine *e) {
    DeviceGuard dg_;
    NMFX_HIP(hipSetDevice(e->device));
    e->hpad_valid = false;
    if (e->algo == 3) {
        if (!e->Z) { set_error("nmfx_engine_init: constrainednmf needs nmfx_engine_set_constraint first"); return NMFX_ERR_INVALID; }
        TRY(z_update(e->st, e->Z, e->H, nullptr, nullptr, nullptr, e->K, e->nz, e->seg_dev, nullptr, nullptr, 1.0f, 1));   // H = Z*A, constrainednmf.m:177
    }
    {
        Scope s(e, TAG_SMALL);
        TRY(col_reduce(e->st, e->W, e->m, e->m, e->KT, e->algo == 2 ? 0 : 1, e->sumsq));   // lnmf.m:59: L1 sums
        TRY(w_normalize(e->st, e->W, e->m, e->K, e->T, e->sumsq, nullptr, norm_mode(e), e->f_out, e->K_valid));
        if (e->algo == 1) TRY(scale_rows(e->st, e->Hext, e->K, e->hL + e->n + e->hR, e->f_out));   // halos too: every rank applies the same factors
        if (e->fused) {
            e->cost_valid = false;
            if (e->VT) TRY(transpose_f32(e->st, e->V, e->m, e->n, e->VT));   // once: V is constant over the iterations
            if (e->div == NMFX_DIV_KL) {   // sum(V_local), once
                TRY(col_reduce(e->st, e->V, e->m, e->m, (int)e->n, 0, e->colV));
                TRY(sum_vec(e->st, e->colV, e->n, e->sumV));
            }
            if (e->dual && e->div == NMFX_DIV_AB) {   // sum(V.^(alpha+beta)) for the cost, V.^alpha as the kernels' data operand; once
                TRY(col_reduce_pow(e->st, e->V, e->m, e->m, (int)e->n, (float)(e->alpha + e->beta), e->colV));
                TRY(sum_vec(e->st, e->colV, e->n, e->sumVab));
                if (e->Valpha) TRY(pow_map(e->st, e->V, e->Valpha, (long)e->m * e->n, (float)e->alpha));
            }
            return refresh_w_derived(e);
        }
    }
    if (e->gram) {                 // no V_hat state on the Gram path
        if (e->use_vtq) TRY(transpose_f32(e->st, e->V, e->m, e->n, e->VT));   // once: V is constant over the iterations
        return NMFX_OK;
    }
    if (e->fusedT_kl) {            // nor here: sum(V) for the closed-form part of the KL cost, once
        Scope s(e, TAG_SMALL);
        e->cost_valid = false;
        TRY(col_reduce(e->st, e->V, e->m, e->m, (int)e->n, 0, e->colV_g));
        return sum_vec(e->st, e->colV_g, e->n, e->sumV_g);
    }
    return recon(e, false);
}

// local sums of the W step: packed = [N | P]  or  [N | Pvec]        nmf.m:149-164 / cnmf.m:187-192
static nmfx_status fused_wstep_tail(nmfx_engine *e);
static nmfx_status generic_wstep_partial(nmfx_engine *e);
nmfx_status nmfx_engine_wstep_partial(nmfx_engine *e) {
    DeviceGuard dg_;
    NMFX_HIP(hipSetDevice(e->device));
    if (e->fused) {
        // one pass over V: N = (V./(W*H)) * H' (KL) or V*H' (euclidean), and the cost of the current (W, H) as a by-product
        e->w_chunks = 1;
        TRY(fused_wpass(e, true));
        return fused_wstep_tail(e);
    }
    return generic_wstep_partial(e);
}

// what follows the last row chunk of a fused W-step partial: the small tail of `packed`
static nmfx_status fused_wstep_tail(nmfx_engine *e) {
    const size_t mKT = (size_t)e->m * e->KT;
    if (e->dual) return NMFX_OK;   // [N | P] is complete: both halves came out of the pass
    if (e->div == NMFX_DIV_KL) {
        // rowsum(H) was formed by fused_wpass_finish, whose cost finisher has also written it into the tail of `packed` (tail_with_cost)
    } else {   // Gram form: V_hat*H' = W*(H*H'); the K x K Gram is what gets all-reduced   
        Scope s(e, TAG_GRAM);
        TRY(small_gemm(e, e->K, e->K, e->n, OpView{e->H, nullptr, (long)e->K, VIEW_RC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f},
                       OpView{e->H, nullptr, (long)e->K, VIEW_RC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f}, e->packed + mKT, e->K));
    }
    return NMFX_OK;
}

// row-chunked form of the fused W-step partial (overlap of the all-reduce with compute on column shards): chunk c of nchunks
// computes rows [c*m/nchunks, (c+1)*m/nchunks) of N into the contiguous block packed + c*(m/nchunks)*K; after the last chunk the
// tail ([rowsum(H)] or [H*H']) and the lagged cost are ready.  wstep_finish reads the chunked layout.
nmfx_status nmfx_engine_wstep_partial_chunk(nmfx_engine *e, int32_t chunk, int32_t nchunks) {
    DeviceGuard dg_;
    NMFX_HIP(hipSetDevice(e->device));
    if (!e->fused || e->dual) { set_error("nmfx_engine_wstep_partial_chunk: fused kl / euclidean path only"); return NMFX_ERR_UNSUPPORTED; }
    if (nchunks < 1 || chunk < 0 || chunk >= nchunks || e->m % (128L * nchunks) != 0) { set_error("nmfx_engine_wstep_partial_chunk: m must split into nchunks multiples of 128 rows"); return NMFX_ERR_INVALID; }
    const long rows = e->m / nchunks;
    if (chunk == 0) { e->chunk_parts = 0; e->w_chunks = nchunks; e->cost_valid = false; }
    TRY(fused_wpass_rows(e, true, rows * chunk, rows, e->packed + (size_t)chunk * rows * e->K));
    if (chunk + 1 < nchunks) return NMFX_OK;
    e->tail_with_cost = true;
    TRY(fused_wpass_finish(e));
    return fused_wstep_tail(e);
}
// element range of `packed` that becomes final with chunk c (the last one carries the tail): what the caller all-reduces
nmfx_status nmfx_engine_packed_chunk(nmfx_engine *e, int32_t chunk, int32_t nchunks, size_t *offset, size_t *count) {
    if (nchunks < 1 || chunk < 0 || chunk >= nchunks || e->m % nchunks != 0) { set_error("nmfx_engine_packed_chunk: bad chunk"); return NMFX_ERR_INVALID; }
    const size_t per = (size_t)(e->m / nchunks) * e->KT, mKT = (size_t)e->m * e->KT;
    size_t tail = 0;
    if (e->fused) tail = e->dual ? mKT : (e->div == NMFX_DIV_EUCLIDEAN ? (size_t)e->K * e->K : (size_t)e->KT);
    else if (nchunks != 1) { set_error("nmfx_engine_packed_chunk: only the fused path chunks its W step"); return NMFX_ERR_UNSUPPORTED; }
    else tail = e->gram ? (size_t)e->KT * e->KT : (div_has_matrix_den(e->div) ? mKT : (size_t)e->KT);
    *offset = per * chunk;
    *count = per + (chunk + 1 == nchunks ? tail : 0);
    return NMFX_OK;
}

static nmfx_status generic_wstep_partial(nmfx_engine *e) {
    const size_t mKT = (size_t)e->m * e->KT;
    if (e->fusedT_kl) {   // S pass: R = V./V_hat into the V_hat buffer + the (lagged) cost of the state this iteration starts from
        TRY(fusedT_pass(e, e->all_fixW ? FT_COST_KL : FT_S_KL, nullptr));
        TRY(fusedT_kl_cost(e));
    }
    if (e->all_fixW) return NMFX_OK;
    OpView a{}, b{};
    num_view(e, a);
    if (e->fusedT || e->fusedT_kl) TRY(fusedT_pass(e, FT_NUM, e->packed));   // all T numerators in one pass over V (KL: over R), the shifted H tile in LDS
    else TRY(x_times_ht(e, a, e->packed, TAG_WNUM));
    if (e->lagram) {   // Hs*Hs' from the T lag Grams L_d = sum_u H(:,u) H(:,u+d)' (K x T*K, contraction n, on the zero-padded copy) + boundary terms
        TRY(ensure_hpad(e));
        Scope s(e, TAG_GRAM);
        const float *Hc = e->Hpad + (size_t)e->K * (e->T - 1);
        TRY(small_gemm(e, e->K, e->KT, e->n, OpView{Hc, nullptr, (long)e->K, VIEW_RC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f},
                       OpView{Hc + (size_t)e->K * (e->T - 1), nullptr, (long)e->K, VIEW_HSTACK_RC, e->K, 0, 0, NMFX_PRO_NONE, 0.f, 0.f, e->T - 1}, e->Llag, e->K));
        TRY(gram_from_lags(e->st, e->Llag, e->H, e->K, e->T, e->n, e->packed + mKT));
    } else if (e->gram) {   // Hs*Hs' (KT x KT): what gets all-reduced instead of V_hat*Hs'
        Scope s(e, TAG_GRAM);
        OpView hs{e->H, nullptr, (long)e->K, e->T == 1 ? VIEW_RC : VIEW_HSTACK_RC, e->K, 0, 0, NMFX_PRO_NONE, 0.f, 0.f, e->hL};
        TRY(small_gemm(e, e->KT, e->KT, e->n, hs, hs, e->packed + mKT, e->KT));
    } else if (div_has_matrix_den(e->div)) {
        den_view(e, b);
        TRY(x_times_ht(e, b, e->packed + mKT, TAG_WDEN));
    } else {
        Scope s(e, TAG_SMALL);
        TRY(row_reduce(e->st, e->H, e->K, e->K, e->n, 0, e->rowsum, e->rr_scratch));
        TRY(kl_pvec(e->st, e->rowsum, e->H, e->K, e->n, e->T, e->Pvec, e->hL));
        TRY(d2f(e->st, e->Pvec, e->packed + mKT, e->KT));
    }
    return NMFX_OK;
}

// replicated part of the W step (after the all-reduce of packed): nmf.m:168-173 / cnmf.m:193-204
nmfx_status nmfx_engine_wstep_finish(nmfx_engine *e) {
    DeviceGuard dg_;
    NMFX_HIP(hipSetDevice(e->device));
    if (e->fused) {
        if (e->all_fixW) return NMFX_OK;
        const size_t mK = (size_t)e->m * e->K;
        WUpdateParams p{};
        p.W = e->W; p.N = e->packed; p.m = e->m; p.K = e->K; p.T = 1;
        p.n_chunks = e->w_chunks > 1 ? e->w_chunks : 1;   // row-chunked partial: N is stored as contiguous (m/chunks x K) blocks
        p.lamW = e->lamW; p.fixW = e->fixW; p.sumsq = e->sumsq; p.inv_exp = 1.0f;
        if (e->dual) {
            p.P = e->packed + mK;       // nmf.m:155-156,162-163: the all-reduced denominators
            p.inv_exp = outer_exp(e);
        } else if (e->div == NMFX_DIV_KL) {
            Scope s(e, TAG_SMALL);
            p.Pvecf = e->packed + mK;   // the all-reduced rowsum(H), still fp32 as it travelled
        } else {
            Scope s(e, TAG_GRAM);   // P = W * (H*H')
            TRY(small_gemm(e, e->m, e->K, e->K, OpView{e->W, nullptr, e->m, VIEW_RC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f},
                           OpView{e->packed + mK, nullptr, (long)e->K, VIEW_KC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f}, e->Pbuf, e->m));
            p.P = e->Pbuf;
        }
        Scope s(e, TAG_SMALL);
        p.rule = e->algo == 2 ? 1 : 0;
        // update, column normalisation (nmf.m:169 / lnmf.m:70) and, for KL, the column sums of the final W (H-step denominator) in ONE launch
        p.fuse_norm = norm_mode(e) == 2 ? 2 : 1;
        p.colsum_out = e->div == NMFX_DIV_KL ? e->Gpvec : nullptr;
        TRY(w_update(e->st, p));
        e->cost_valid = false;
        return refresh_w_derived(e, true);
    }
    if (!e->all_fixW) {
        Scope s(e, TAG_SMALL);
        const size_t mKT = (size_t)e->m * e->KT;
        WUpdateParams p{};
        p.W = e->W; p.N = e->packed; p.m = e->m; p.K = e->K; p.T = e->T;
        p.lamW = e->lamW; p.fixW = e->fixW; p.sumsq = e->sumsq; p.inv_exp = outer_exp(e);
        if (e->gram) {   // P_all = W_flat * (Hs*Hs')
            TRY(small_gemm(e, e->m, e->KT, e->KT, OpView{e->W, nullptr, e->m, VIEW_RC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f},
                           OpView{e->packed + mKT, nullptr, (long)e->KT, VIEW_KC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f}, e->Pbuf, e->m));
            p.P = e->Pbuf;
        } else if (div_has_matrix_den(e->div)) p.P = e->packed + mKT;
        else {
            TRY(f2d(e->st, e->packed + mKT, e->Pvec, e->KT));
            p.Pvec = e->Pvec;
        }
        p.rule = e->algo == 2 ? 1 : 0;
        TRY(w_update(e->st, p));
        TRY(w_normalize(e->st, e->W, e->m, e->K, e->T, e->sumsq, e->fixW, norm_mode(e), nullptr));
    }
    if (e->gram || e->fusedT_kl) return NMFX_OK;
    return recon(e, false);
}

nmfx_status nmfx_engine_hstep_finish(nmfx_engine *e);
// H step + V_hat refresh + local cost partial: nmf.m:176-218 / cnmf.m:207-251
nmfx_status nmfx_engine_hstep(nmfx_engine *e) {
    DeviceGuard dg_;
    NMFX_HIP(hipSetDevice(e->device));
    if (e->fused) {
        if (e->all_fixH) return NMFX_OK;
        if (e->div == NMFX_DIV_EUCLIDEAN) {   // W'*V_hat = (W'*W)*H   
            Scope s(e, TAG_GRAM);
            TRY(small_gemm(e, e->K, e->K, e->m, OpView{e->W, nullptr, e->m, VIEW_KC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f},
                           OpView{e->W, nullptr, e->m, VIEW_KC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f}, e->GW, e->K));
            TRY(small_gemm(e, e->K, e->n, e->K, OpView{e->GW, nullptr, (long)e->K, VIEW_RC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f},
                           OpView{e->H, nullptr, (long)e->K, VIEW_KC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f}, e->Gp, e->K));
        }
        FusedParams f;
        memset(&f, 0, sizeof(f));
        f.X = e->H; f.xs_r = e->K; f.xs_k = 1;
        f.Y = e->WT; f.D = e->V; f.ldd = e->m; f.R = e->n; f.Cn = e->m; f.K = e->K;
        f.c_per_split = e->cps_h;
        const int func = e->dual ? (mdiv(e) == NMFX_DIV_IS ? 4 : 5) : (e->div == NMFX_DIV_KL ? 2 : 0);
        const bool kl = e->div == NMFX_DIV_KL;
        if (e->dual) {
            f.ab_alpha = (float)e->alpha; f.ab_beta = (float)e->beta; f.inv_exp = outer_exp(e);
            if (e->Valpha) f.D = e->Valpha;
        }
        static const bool euc_fused_h = getenv("NMFX_EUC_HSTEP_FUSED") != nullptr;   // dev switch: previous behaviour
        if (func == 0 && e->VT && !euc_fused_h) {
            // euclidean: the numerator W'*V has no first product.  The H-step form of the stationary kernel reads its V tile with the lanes
            // ACROSS columns (16-byte pieces at stride m) and, with half the MFMA work per tile to hide that under, ran 0.61 ms at C2; the
            // pipelined two-operand GEMM 0.60 ms (0.75 of peak).  V never changes, so a transposed copy made once turns the product into
            // (V'*W)' on the W-STEP form -- lanes along the contiguous dimension, the pass V*H' already runs at 0.86 of peak:
            //   stationary rows = columns j of V (rows of V'), streamed rows = rows i of W (the W' copy), out(k, j) at Gn[k + K*j]
            FusedParams g;
            memset(&g, 0, sizeof(g));
            g.Y = e->WT; g.D = e->VT; g.ldd = e->n; g.R = e->n; g.Cn = e->m; g.K = e->K; g.c_per_split = e->cps_h;
            g.out = e->isplit_h == 1 ? e->Gn : e->slabs; g.slab_stride = (long)e->K * e->n; g.os_r = e->K; g.os_k = 1;
            {
                Scope s(e, TAG_HNUM);
                TRY(launch_fused(e->st, g, e->isplit_h, true, 0, true, 0));
            }
            Scope s(e, TAG_SMALL);
            const bool fuse_sum = e->isplit_h > 1 && e->algo != 3;   // h_update sums the slabs on the fly
            if (e->isplit_h > 1 && !fuse_sum) TRY(reduce_slabs(e->st, e->slabs, e->isplit_h, g.slab_stride, g.slab_stride, e->Gn, 0));
            if (fuse_sum) TRY(h_update(e->st, e->H, e->slabs, e->Gp, nullptr, e->K, e->n, e->lamH, e->fixH, 1.0f, e->isplit_h, g.slab_stride));
            else if (e->algo == 3) TRY(z_update(e->st, e->Z, e->H, e->Gn, e->Gp, e->Gpvec, e->K, e->nz, e->seg_dev, e->lamH, e->fixH, 1.0f, 0));
            else TRY(h_update(e->st, e->H, e->Gn, e->Gp, nullptr, e->K, e->n, e->lamH, e->fixH, 1.0f));
        } else if (func == 0 && !euc_fused_h && e->K % 64 == 0) {   // K % 64 != 0 would drop the GEMM to its unaligned (general) kernel
            // euclidean: the numerator W'*V needs no first product, so the register-stationary kernel has half the MFMA work
            // per tile barrier; the pipelined GEMM runs this plain contraction faster (C2: 0.87 -> ~0.6 ms)
            {
                Scope s(e, TAG_HNUM);
                GemmParams g;
                memset(&g, 0, sizeof(g));
                g.M = e->K; g.N = e->n; g.Kc = e->m;
                g.A = OpView{e->W, nullptr, e->m, VIEW_KC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f};
                g.B = OpView{e->V, nullptr, e->m, VIEW_KC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f};
                g.C = e->Gn; g.ldc = e->K; g.epi = EPI_STORE; g.splitk = 1;
                TRY(gemm_auto(e->st, g, e->gemm_scratch, e->gemm_scratch_bytes));
            }
            Scope s(e, TAG_SMALL);
            if (e->algo == 3) TRY(z_update(e->st, e->Z, e->H, e->Gn, e->Gp, e->Gpvec, e->K, e->nz, e->seg_dev, e->lamH, e->fixH, 1.0f, 0));
            else TRY(h_update(e->st, e->H, e->Gn, e->Gp, nullptr, e->K, e->n, e->lamH, e->fixH, 1.0f));
        } else if (e->isplit_h == 1 && e->algo != 3) {
            f.Hio = e->H; f.den = kl ? nullptr : e->Gp; f.denvec = kl ? e->Gpvec : nullptr; f.lam = e->lamH; f.fix = e->fixH;
            f.sqrt_rule = e->algo == 2;
            Scope s(e, TAG_FUSED_H);
            TRY(launch_fused(e->st, f, 1, false, func, true, 1));
        } else if (e->dual) {   // split over the rows of W, or constrainednmf: numerator and denominator slabs, then the generic update
            f.out = e->isplit_h == 1 ? e->Gn : e->slabs; f.out2 = e->isplit_h == 1 ? e->Gp : e->slabs2;
            f.slab_stride = (long)e->K * e->n; f.os_r = e->K; f.os_k = 1;
            {
                Scope s(e, TAG_FUSED_H);
                TRY(launch_fused(e->st, f, e->isplit_h, false, func, true, 0));
            }
            Scope s(e, TAG_SMALL);
            if (e->isplit_h > 1) {
                TRY(reduce_slabs(e->st, e->slabs, e->isplit_h, f.slab_stride, f.slab_stride, e->Gn, 0));
                TRY(reduce_slabs(e->st, e->slabs2, e->isplit_h, f.slab_stride, f.slab_stride, e->Gp, 0));
            }
            if (e->algo == 3) TRY(z_update(e->st, e->Z, e->H, e->Gn, e->Gp, e->Gpvec, e->K, e->nz, e->seg_dev, e->lamH, e->fixH, outer_exp(e), 0));
            else TRY(h_update(e->st, e->H, e->Gn, e->Gp, nullptr, e->K, e->n, e->lamH, e->fixH, outer_exp(e)));
        } else {
            f.out = e->isplit_h == 1 ? e->Gn : e->slabs; f.slab_stride = (long)e->K * e->n; f.os_r = e->K; f.os_k = 1;
            {
                Scope s(e, TAG_FUSED_H);
                TRY(launch_fused(e->st, f, e->isplit_h, false, func, true, 0));
            }
            Scope s(e, TAG_SMALL);
            const bool fuse_sum = e->isplit_h > 1 && e->algo != 3;   // h_update sums the slabs on the fly
            if (e->isplit_h > 1 && !fuse_sum) TRY(reduce_slabs(e->st, e->slabs, e->isplit_h, f.slab_stride, f.slab_stride, e->Gn, 0));
            if (fuse_sum) TRY(h_update(e->st, e->H, e->slabs, kl ? nullptr : e->Gp, kl ? e->Gpvec : nullptr, e->K, e->n, e->lamH, e->fixH, e->algo == 2 ? -2.0f : 1.0f,
                                       e->isplit_h, f.slab_stride));
            else if (e->algo == 3) TRY(z_update(e->st, e->Z, e->H, e->Gn, kl ? nullptr : e->Gp, e->Gpvec, e->K, e->nz, e->seg_dev, e->lamH, e->fixH, 1.0f, 0));
            else TRY(h_update(e->st, e->H, e->Gn, kl ? nullptr : e->Gp, kl ? e->Gpvec : nullptr, e->K, e->n, e->lamH, e->fixH, e->algo == 2 ? -2.0f : 1.0f));
        }
        e->cost_valid = false;
        return NMFX_OK;
    }
    if (!e->all_fixH) {
        OpView a{}, b{};
        num_view(e, a);
        if (e->lagram) TRY(ensure_hpad(e));   // the denominator below reads the padded copy of the CURRENT H
        if (e->fusedT_kl) {   // R = V./V_hat with the W just updated
            TRY(fusedT_pass(e, FT_S_KL, nullptr));
            a = OpView{e->Vhat, nullptr, e->m, VIEW_KC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f};
        }
        e->hpad_valid = false;
        if (e->qgemm) {
            // sum_t W_t' * lshift_t(V) as ONE well-shaped GEMM Q = W_flat' * V (KT x n, contraction m) + a shift-sum over t, instead of a
            // (K x n) GEMM with contraction T*m whose 64-row output starves the tiles
            if (e->use_vtq && a.p == e->V && !a.p2 && a.func == NMFX_PRO_NONE && e->nvalid == e->n) {
                // Q' = V' * W_flat on the W-step form of the stationary kernel (rows of V' stationary, rows of W_flat streamed as K-wide column
                // blocks, one block per grid.z): its V tile is read along the contiguous dimension, which the two-operand GEMM (0.77 of peak
                // here) and the H-step form cannot offer
                {
                    Scope s2(e, TAG_SMALL);
                    TRY(transpose_f32(e->st, e->W, e->m, e->KT, e->WTf));
                }
                FusedParams q;
                memset(&q, 0, sizeof(q));
                const int kb = e->vtq_block;
                q.Y = e->WTf; q.y_stride = e->KT; q.nz = e->KT / kb; q.yz_stride = kb; q.oz_stride = kb;
                q.D = e->VT; q.ldd = e->n; q.R = e->n; q.Cn = e->m; q.K = kb;
                long cps = 0;
                const int split = fused_split(((e->n + 127) / 128) * q.nz, e->m, kb, &cps);
                const bool can_split = split > 1 && e->gemm_scratch_bytes >= sizeof(float) * (size_t)split * e->KT * e->n;
                q.c_per_split = can_split ? cps : (e->m + 63) / 64 * 64;
                q.out = can_split ? e->gemm_scratch : e->Qbuf; q.slab_stride = (long)e->KT * e->n; q.os_r = e->KT; q.os_k = 1;
                {
                    Scope s2(e, TAG_HNUM);
                    TRY(launch_fused(e->st, q, can_split ? split : 1, true, 0, true, 0));
                }
                if (can_split) { Scope s2(e, TAG_SMALL); TRY(reduce_slabs(e->st, e->gemm_scratch, split, q.slab_stride, q.slab_stride, e->Qbuf, 0)); }
            } else {
                Scope s(e, TAG_HNUM);
                GemmParams g;
                memset(&g, 0, sizeof(g));
                g.M = e->KT; g.N = e->nvalid; g.Kc = e->m;
                g.A = OpView{e->W, nullptr, e->m, VIEW_KC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f};
                a.ld = e->m; a.mode = VIEW_KC; a.blk = 0; a.tstride = 0; a.lim = 0;   // the numerator operand (V, V./V_hat, ...) un-shifted
                g.B = a;
                g.C = e->Qbuf; g.ldc = e->KT; g.epi = EPI_STORE; g.splitk = 1;
                TRY(gemm_auto(e->st, g, e->gemm_scratch, e->gemm_scratch_bytes));
            }
            Scope s(e, TAG_SMALL);
            TRY(shift_sum(e->st, e->Qbuf, e->K, e->T, e->n, e->nvalid, e->Gn));
        } else TRY(wt_times_x(e, a, e->Gn, TAG_HNUM));
        if (e->gram) {
            // sum_t W_t' * lshift_t(V_hat) = sum_t D_t * lshift_t(Hs),  D = W_flat' * W_flat  (cnmf.m:217-226 without V_hat)
            Scope s(e, TAG_GRAM);
            OpView wf{e->W, nullptr, e->m, VIEW_KC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f};
            TRY(small_gemm(e, e->KT, e->KT, e->m, wf, wf, e->CC, e->KT));
            if (e->lagram) {
                // by lag: E_d = sum_{t-t'=d} D_(t,t'), Gp = sum_d E_d * H(:, j+d) as ONE K x n GEMM with contraction (2T-1)*K over the padded H;
                // the last T-1 columns (where lshift_t drops terms) term by term
                TRY(lag_sum(e->st, e->CC, e->K, e->T, e->Elag));
                const float *Hc = e->Hpad + (size_t)e->K * (e->T - 1);
                TRY(small_gemm(e, e->K, e->n, (long)(2 * e->T - 1) * e->K, OpView{e->Elag, nullptr, (long)e->K, VIEW_RC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f},
                               OpView{Hc + (size_t)e->K * (e->T - 1), nullptr, (long)e->K, VIEW_HSTACK_KC, e->K, 0, 0, NMFX_PRO_NONE, 0.f, 0.f, 2 * (e->T - 1)}, e->Gp, e->K));
                TRY(gp_tail(e->st, e->CC, e->H, e->K, e->T, e->n, e->Gp));
            } else {
            GemmParams g;
            memset(&g, 0, sizeof(g));
            g.M = e->K; g.N = e->n; g.Kc = e->KT;   // columns j >= n - t are masked by the view (lshift zero fill), so N stays tileable
            g.A = OpView{e->CC, nullptr, (long)e->KT, VIEW_RC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f};
            g.B = OpView{e->H, nullptr, (long)e->K, VIEW_HSTACK_KC, e->K, e->nvalid, 0, NMFX_PRO_NONE, 0.f, 0.f, e->hL};
            g.ldc = e->K; g.epi = EPI_STORE; g.splitk = 1;
            const size_t Kn = (size_t)e->K * e->n;
            if (e->T > 1 && gemm_pipe_eligible(g) && e->gemm_scratch_bytes >= Kn * e->T * sizeof(float)) {
                // all T shifts in ONE launch (blockIdx.z = t, slab t), then a deterministic slab sum
                g.zbatch = e->T; g.zA_off = e->K; g.zB_off = e->K; g.zB_lim = 1; g.zB_tstride = -1;
                g.C = e->gemm_scratch; g.slab_stride = (long)Kn;
                TRY(launch_gemm(e->st, g));
                TRY(reduce_slabs(e->st, e->gemm_scratch, e->T, (long)Kn, (long)Kn, e->Gp, 0));
            } else {
                for (int t = 0; t < e->T; ++t) {
                    GemmParams gt = g;
                    gt.A.p = e->CC + (long)t * e->K;
                    gt.B.p = e->H + (long)e->K * t; gt.B.tstride = e->nvalid - t; gt.B.lim = t;
                    gt.C = e->Gp; gt.accumulate = t > 0;
                    TRY(launch_gemm(e->st, gt));
                }
            }
            }
        } else if (div_has_matrix_den(e->div)) {
            den_view(e, b);
            TRY(wt_times_x(e, b, e->Gp, TAG_HDEN));
        }
        Scope s(e, TAG_SMALL);
        if (!div_has_matrix_den(e->div)) {
            TRY(col_reduce(e->st, e->W, e->m, e->m, e->KT, 0, e->colsum));
            TRY(sum_over_t(e->st, e->colsum, e->K, e->T, e->Gpvec));
        }
        if (e->algo == 3) TRY(z_update(e->st, e->Z, e->H, e->Gn, e->Gp, e->Gpvec, e->K, e->nz, e->seg_dev, e->lamH, e->fixH, outer_exp(e), 0));
        else TRY(h_update(e->st, e->H, e->Gn, e->Gp, div_has_matrix_den(e->div) ? nullptr : e->Gpvec, e->K, e->n, e->lamH, e->fixH, e->algo == 2 ? -2.0f : outer_exp(e)));
    }
    if (e->defer_hfinish) return NMFX_OK;   // the caller refreshes H's halos first, then calls nmfx_engine_hstep_finish
    return nmfx_engine_hstep_finish(e);
}

// second half of the H step on the generic paths: V_hat refresh (+ cost) with the NEW H -- on a column shard the halo
// columns of H must have been refreshed from the neighbours before this runs (V_hat near the shard edges depends on them)
nmfx_status nmfx_engine_hstep_finish(nmfx_engine *e) {
    DeviceGuard dg_;
    NMFX_HIP(hipSetDevice(e->device));
    if (e->fused) return NMFX_OK;
    const bool nocost = e->div == NMFX_DIV_EUCLIDEAN_NOCOST;
    if (e->fusedT_kl) { e->cost_valid = false; return NMFX_OK; }   // the cost lags: by-product of the next S pass, or nmfx_engine_cost_pass
    if (e->fusedT) { if (!nocost) TRY(fusedT_pass(e, FT_COST_EUC, nullptr)); }   // S = sum_t W_t * rshift_t(H) in registers -> residual
    else if (e->gram) { if (!nocost) TRY(recon(e, true, false)); }   // residual reduction only, V_hat is not stored
    else TRY(recon(e, !nocost));
    Scope s(e, TAG_SMALL);
    e->cost_valid = true;
    return cost_from_partials(e, nocost ? 0 : e->n_cost_used);
}
nmfx_status nmfx_engine_defer_hstep_finish(nmfx_engine *e, int32_t defer) { e->defer_hfinish = defer != 0; return NMFX_OK; }

// make e->cost hold the cost of the CURRENT (W, H): free on the generic path (hstep already did it), one S = W*H pass on the
// fused path unless the last wstep_partial just produced it
nmfx_status nmfx_engine_cost_pass(nmfx_engine *e) {
    DeviceGuard dg_;
    NMFX_HIP(hipSetDevice(e->device));
    if (e->cost_valid) return NMFX_OK;
    if (e->fused) return fused_wpass(e, false);
    if (e->fusedT_kl) { TRY(fusedT_pass(e, FT_COST_KL, nullptr)); return fusedT_kl_cost(e); }
    set_error("nmfx_engine_cost_pass: no cost available yet (call hstep first)");
    return NMFX_ERR_INVALID;
}
int32_t nmfx_engine_is_fused(nmfx_engine *e) { return e->fused ? 1 : (e->fusedT_kl ? 4 : (e->fusedT ? 3 : (e->gram ? 2 : 0))); }   // 1 fused kernels, 3 fused cnmf passes + Gram denominators, 2 Gram form on the GEMM, 0 materialised V_hat

nmfx_status nmfx_engine_cost_ptr(nmfx_engine *e, double **dev_cost) { *dev_cost = e->cost; return NMFX_OK; }
nmfx_status nmfx_engine_copy_cost(nmfx_engine *e, double *dst_dev) {
    NMFX_HIP(hipMemcpyAsync(dst_dev, e->cost, sizeof(double), hipMemcpyDeviceToDevice, e->st));
    return NMFX_OK;
}

// the whole stretch between two all-reduces of a column-sharded run as ONE call: replicated W update, local H step, and (unless
// `last`) the next iteration's W-step partial.  Not for cnmf shards, whose H step is split around the halo exchange.
nmfx_status nmfx_engine_between_allreduces(nmfx_engine *e, int32_t last) {
    if (e->hL || e->hR) { set_error("nmfx_engine_between_allreduces: not for shards with halos"); return NMFX_ERR_UNSUPPORTED; }
    TRY(nmfx_engine_wstep_finish(e));
    TRY(nmfx_engine_hstep(e));
    if (!last) TRY(nmfx_engine_wstep_partial(e));
    return NMFX_OK;
}

nmfx_status nmfx_engine_iterate(nmfx_engine *e, int32_t iters, double *dev_cost_out) {
    const bool lag = e->fused || e->fusedT_kl;   // the cost of iteration i is a by-product of the first pass of iteration i+1
    for (int it = 0; it < iters; ++it) {
        // fused path: the W-step pass also produces the cost of the state it starts from, i.e. of iteration it-1; its finisher writes it
        // straight into the caller's vector (no separate 8-byte copy)
        e->cost_dst2 = (lag && it > 0 && dev_cost_out) ? dev_cost_out + it - 1 : nullptr;
        nmfx_status ws_ = nmfx_engine_wstep_partial(e);
        e->cost_dst2 = nullptr;
        TRY(ws_);
        TRY(nmfx_engine_wstep_finish(e));
        // un-lagged paths: the cost of this iteration is finished inside the H step; its finisher writes the caller's slot too
        e->cost_dst2 = (!lag && dev_cost_out) ? dev_cost_out + it : nullptr;
        e->cost_dst2_done = false;
        nmfx_status hs_ = nmfx_engine_hstep(e);
        e->cost_dst2 = nullptr;
        TRY(hs_);
        if (!lag && dev_cost_out && !e->cost_dst2_done) NMFX_HIP(hipMemcpyAsync(dev_cost_out + it, e->cost, sizeof(double), hipMemcpyDeviceToDevice, e->st));
    }
    if (lag && iters > 0 && dev_cost_out) {   // cost of the last iteration: one extra S = W*H pass
        TRY(nmfx_engine_cost_pass(e));
        NMFX_HIP(hipMemcpyAsync(dev_cost_out + iters - 1, e->cost, sizeof(double), hipMemcpyDeviceToDevice, e->st));
    }
    return NMFX_OK;
}

// ---- profiling: hipEvent pairs around every launch group, on the engine's stream -----------------
nmfx_status nmfx_engine_profile(nmfx_engine *e, int32_t enable) {   // 0 off | 1 every launch group | 2 the MFMA launch groups only
    e->prof.skip_tag = enable == 2 ? (int)TAG_SMALL : -1;
    e->prof.enable(enable != 0);
    return NMFX_OK;
}
int32_t nmfx_engine_profile_ntags(void) { return TAG_COUNT; }
const char *nmfx_engine_profile_tag_name(int32_t tag) { return (tag >= 0 && tag < TAG_COUNT) ? kTagNames[tag] : ""; }
// after the stream is synchronised: total ms and launch count per tag
nmfx_status nmfx_engine_profile_read(nmfx_engine *e, double *ms_per_tag, int32_t *count_per_tag) {
    return e->prof.read(TAG_COUNT, ms_per_tag, count_per_tag);
}
// algorithmic flops of ONE launch of the GEMM behind `tag` (2*M*N*Kc by formula) and its algorithmic HBM bytes
nmfx_status nmfx_engine_tag_work(nmfx_engine *e, int32_t tag, double *flops, double *bytes) {
    const double m = (double)e->m, n = (double)e->n, KT = (double)e->KT;
    const double f = 2.0 * m * n * KT;
    const bool two_in = mdiv(e) != NMFX_DIV_EUCLIDEAN;
    double b = 0.0;
    switch (tag) {
    case TAG_RECON: b = 4.0 * (m * n + m * KT + e->K * n); break;
    case TAG_RECON_COST: b = 4.0 * (2.0 * m * n + m * KT + e->K * n); break;
    case TAG_WNUM: b = 4.0 * ((two_in ? 2.0 : 1.0) * m * n + m * KT + e->K * n); break;
    case TAG_WDEN: b = 4.0 * (m * n + m * KT + e->K * n); break;
    case TAG_HNUM: b = 4.0 * ((two_in ? 2.0 : 1.0) * m * n + m * KT + 2.0 * e->K * n); break;
    case TAG_HDEN: b = 4.0 * (m * n + m * KT + 2.0 * e->K * n); break;
    // fused passes: V streamed once; both contractions counted when both are issued (KL; euclidean W step with cost)
    case TAG_FUSED_W: {   // one launch covers m / w_chunks rows when the partial is row-chunked
        const double ch = e->w_chunks > 1 ? (double)e->w_chunks : 1.0;
        if (e->fusedT || e->fusedT_kl) { *flops = f; *bytes = 4.0 * (m * n + m * KT + e->K * n); return NMFX_OK; }   // cnmf numerator pass: one contraction
        if (e->dual) { *flops = 3.0 * f; *bytes = 4.0 * (m * n + 3.0 * m * KT + e->K * n); return NMFX_OK; }   // S + two contractions
        *flops = 2.0 * f / ch; *bytes = 4.0 * (m * n / ch + 2.0 * m * KT / ch + e->K * n); return NMFX_OK;
    }
    case TAG_FUSED_H: *flops = (e->dual ? 3.0 : (mdiv(e) == NMFX_DIV_KL ? 2.0 : 1.0)) * f; *bytes = 4.0 * (m * n + m * KT + 2.0 * e->K * n); return NMFX_OK;
    case TAG_FUSED_COST: *flops = f; *bytes = 4.0 * (m * n + m * KT + e->K * n); return NMFX_OK;
    default: *flops = 0; *bytes = 0; return NMFX_OK;
    }
    *flops = f;
    *bytes = b;
    return NMFX_OK;
}

// ---- kernel-level entry point (tests) -------------------------------------------------------------
nmfx_status nmfx_gemm_f32(void *stream, int32_t opA, int32_t opB, int64_t M, int64_t N, int64_t Kc, const float *A, const float *A2,
                          int64_t lda, int32_t proA, const float *B, const float *B2, int64_t ldb, int32_t proB, float *C, int64_t ldc,
                          int32_t accumulate, void *workspace, size_t workspace_bytes) {
    DeviceGuard dg_;
    hipPointerAttribute_t attr;
    if (!C || hipPointerGetAttributes(&attr, C) != hipSuccess) { (void)hipGetLastError(); set_error("nmfx_gemm_f32: C is not a device pointer"); return NMFX_ERR_INVALID; }
    TRY(check_device(attr.device));      // the device the buffers live on, not device 0
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.M = M; g.N = N; g.Kc = Kc;
    g.A = OpView{A, A2, (long)lda, opA == NMFX_OP_N ? VIEW_RC : VIEW_KC, 0, 0, 0, proA, 0.f, 0.f};
    g.B = OpView{B, B2, (long)ldb, opB == NMFX_OP_N ? VIEW_KC : VIEW_RC, 0, 0, 0, proB, 0.f, 0.f};
    g.C = C; g.ldc = ldc; g.accumulate = accumulate; g.epi = EPI_STORE; g.splitk = 1;
    return gemm_auto(static_cast<hipStream_t>(stream), g, workspace, workspace_bytes);
}

}  // extern "C"

// =============================================================================================
// blocking host-buffer API
// =============================================================================================
namespace {

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    nmfx_status alloc(size_t bytes) {
        hipError_t e = hipMalloc(&p, bytes ? bytes : 256);
        if (e != hipSuccess) { set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); p = nullptr; return NMFX_ERR_NOMEM; }
        return NMFX_OK;
    }
    template <class T> T *as() { return static_cast<T *>(p); }
};

size_t dsize(int dtype) { return dtype == NMFX_F64 ? 8 : 4; }

// host (f32/f64) -> device fp32, converted on the device through a staging buffer; out = in / divide_by
nmfx_status upload(hipStream_t st, const void *host, int dtype, float *dev, size_t count, double divide_by, DevBuf &stage, size_t stage_elems) {
    const char *h = static_cast<const char *>(host);
    for (size_t off = 0; off < count; off += stage_elems) {
        size_t c = count - off < stage_elems ? count - off : stage_elems;
        NMFX_HIP(hipMemcpyAsync(stage.p, h + off * dsize(dtype), c * dsize(dtype), hipMemcpyHostToDevice, st));
        TRY(cvt_to_f32(st, stage.p, dtype, dev + off, (long)c, divide_by));
        NMFX_HIP(hipStreamSynchronize(st));
    }
    return NMFX_OK;
}
nmfx_status download(hipStream_t st, const float *dev, int dtype, void *host, size_t count, DevBuf &stage, size_t stage_elems) {
    char *h = static_cast<char *>(host);
    if (dtype == NMFX_F32) {
        NMFX_HIP(hipMemcpyAsync(h, dev, count * 4, hipMemcpyDeviceToHost, st));
        NMFX_HIP(hipStreamSynchronize(st));
        return NMFX_OK;
    }
    for (size_t off = 0; off < count; off += stage_elems) {
        size_t c = count - off < stage_elems ? count - off : stage_elems;
        TRY(cvt_to_f64(st, dev + off, stage.as<double>(), (long)c));
        NMFX_HIP(hipMemcpyAsync(h + off * 8, stage.p, c * 8, hipMemcpyDeviceToHost, st));
        NMFX_HIP(hipStreamSynchronize(st));
    }
    return NMFX_OK;
}

constexpr size_t STAGE_ELEMS = (size_t)8 << 20;  // 64 MiB of doubles

nmfx_status validate_problem(const nmfx_problem *p, const nmfx_result *r, bool nmfsc, bool need_H_init = true) {
    if (!p || !r) { set_error("null problem/result"); return NMFX_ERR_INVALID; }
    if (p->m <= 0 || p->n <= 0 || p->K_total <= 0 || p->T <= 0) { set_error("m, n, K_total, T must be positive"); return NMFX_ERR_INVALID; }
    if (!p->V || !p->W_init || (need_H_init && !p->H_init) || !r->W || !r->H || !r->cost) { set_error("V, W_init, H_init, result.W, result.H, result.cost are required"); return NMFX_ERR_INVALID; }
    if (p->dtype != NMFX_F32 && p->dtype != NMFX_F64) { set_error("dtype must be NMFX_F32 or NMFX_F64"); return NMFX_ERR_INVALID; }
    if (p->maxiter <= 0) { set_error("maxiter must be positive (the wrapper applies the reference default)"); return NMFX_ERR_INVALID; }
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

nmfx_status run_mu(const nmfx_problem *p, nmfx_result *r, int algorithm, const int64_t *seg = nullptr, int64_t nz = 0, const void *Z_init = nullptr,
                   void *Z_out = nullptr) {
    TRY(validate_problem(p, r, false, algorithm != 3));
    if (algorithm != 1 && p->T != 1) { set_error("nmf / lnmf / constrainednmf: T must be 1"); return NMFX_ERR_INVALID; }
    if (algorithm == 3) {
        if (!seg || !Z_init || !Z_out || nz <= 0) { set_error("constrainednmf: segments, Z_init and Z_out are required"); return NMFX_ERR_INVALID; }
        if (p->num_sources != 1) { set_error("constrainednmf: single source only (constrainednmf.m has no multi-source form)"); return NMFX_ERR_INVALID; }
        if (p->divergence == NMFX_DIV_EUCLIDEAN_NOCOST) { set_error("constrainednmf: unknown divergence (constrainednmf.m:204-205)"); return NMFX_ERR_INVALID; }
    }
    if (algorithm == 0 && p->divergence == NMFX_DIV_EUCLIDEAN_NOCOST) { set_error("nmf: unknown divergence (nmf.m:165-166)"); return NMFX_ERR_INVALID; }
    DeviceGuard dg_;
    TRY(check_device(p->device));
    const int Kt = p->K_total, S = p->num_sources;
    // K rounded up to a multiple of 32 with zero, fixed components opens the fused kernels to any K <= 256 on tileable shapes: the
    // padding contributes exact zeros to W*H and to every sum, and is never updated (it is stripped again on the way out)
    const int dv = p->divergence;
    const bool dual_ok = (dv == NMFX_DIV_IS || (dv == NMFX_DIV_AB && p->alpha != 0)) && Kt <= 128;   // fused IS / alpha-beta: K <= 128
    const bool pad = algorithm != 1 && Kt % 32 != 0 && Kt <= 256 && ((p->m >= 64 && p->n >= 64) || p->path == 2) && p->path != 1 &&
                     (dv == NMFX_DIV_KL || dv == NMFX_DIV_EUCLIDEAN || dual_ok);
    const int K = pad ? (Kt + 31) / 32 * 32 : Kt;
    std::vector<float> lw(K, 0.f), lh(K, 0.f);
    std::vector<uint8_t> fw(K, 0), fh(K, 0);
    for (int k = Kt; k < K; ++k) fw[k] = fh[k] = 1;
    for (int s = 0, k0 = 0; s < S; ++s) {
        const int Ks = p->K_s ? p->K_s[s] : Kt;
        for (int k = k0; k < k0 + Ks; ++k) {
            if (p->W_sparsity) lw[k] = (float)p->W_sparsity[s];
            if (p->H_sparsity) lh[k] = (float)p->H_sparsity[s];
            if (p->W_fixed) fw[k] = p->W_fixed[s];
            if (p->H_fixed) fh[k] = p->H_fixed[s];
        }
        k0 += Ks;
    }
    nmfx_engine_desc d{};
    d.m = p->m; d.n_local = p->n; d.K_total = K; d.T = p->T; d.divergence = p->divergence; d.alpha = p->alpha; d.beta = p->beta;
    d.lamW_col = lw.data(); d.lamH_row = lh.data(); d.fixW_col = fw.data(); d.fixH_row = fh.data();
    d.device = p->device; d.stream = nullptr; d.algorithm = algorithm; d.path = p->path;
    d.K_valid = pad ? Kt : 0;
    size_t ws_bytes = 0, packed_count = 0;
    TRY(nmfx_engine_workspace_bytes(&d, &ws_bytes));
    TRY(nmfx_engine_packed_count(&d, &packed_count));
    const size_t mn = (size_t)p->m * p->n, mKT = (size_t)p->m * K * p->T, Kn = (size_t)K * p->n;
    DevBuf V, W, H, Z, ws, packed, stage;
    TRY(V.alloc(mn * 4)); TRY(W.alloc(mKT * 4)); TRY(H.alloc(Kn * 4)); TRY(ws.alloc(ws_bytes)); TRY(packed.alloc(packed_count * 4));
    TRY(stage.alloc(STAGE_ELEMS * 8));
    hipStream_t st = nullptr;
    TRY(upload(st, p->V, p->dtype, V.as<float>(), mn, 1.0, stage, STAGE_ELEMS));
    const size_t mKt = (size_t)p->m * Kt * p->T, Ktn = (size_t)Kt * p->n;
    DevBuf tmp;   // K x cols staging of the un-padded row-interleaved arrays (H, Z)
    if (pad) TRY(tmp.alloc(std::max(Ktn, (size_t)Kt * (size_t)(algorithm == 3 ? nz : 0)) * 4));
    TRY(upload(st, p->W_init, p->dtype, W.as<float>(), mKt, 1.0, stage, STAGE_ELEMS));   // the first K columns of the m x K_pad array
    if (pad) NMFX_HIP(hipMemsetAsync(W.as<float>() + mKt, 0, (mKT - mKt) * 4, st));
    if (algorithm != 3) {
        if (pad) {
            TRY(upload(st, p->H_init, p->dtype, tmp.as<float>(), Ktn, 1.0, stage, STAGE_ELEMS));
            TRY(repack_rows(st, tmp.as<float>(), Kt, H.as<float>(), K, p->n));
        } else TRY(upload(st, p->H_init, p->dtype, H.as<float>(), Kn, 1.0, stage, STAGE_ELEMS));
    } else {   // H = Z*A is formed on the device by nmfx_engine_init (constrainednmf.m:174-177)
        TRY(Z.alloc((size_t)K * nz * 4));
        if (pad) {
            TRY(upload(st, Z_init, p->dtype, tmp.as<float>(), (size_t)Kt * nz, 1.0, stage, STAGE_ELEMS));
            TRY(repack_rows(st, tmp.as<float>(), Kt, Z.as<float>(), K, nz));
        } else TRY(upload(st, Z_init, p->dtype, Z.as<float>(), (size_t)K * nz, 1.0, stage, STAGE_ELEMS));
    }
    nmfx_engine *e = nullptr;
    TRY(nmfx_engine_create(&d, V.as<float>(), W.as<float>(), H.as<float>(), ws.p, ws_bytes, packed.as<float>(), &e));
    nmfx_status s = algorithm == 3 ? nmfx_engine_set_constraint(e, seg, nz, Z.as<float>()) : NMFX_OK;
    if (s == NMFX_OK) s = nmfx_engine_init(e);
    int it = 0;
    r->iters_run = 0;
    auto read_cost = [&](int idx) -> nmfx_status {
        hipError_t he = hipMemcpy(&r->cost[idx], e->cost, sizeof(double), hipMemcpyDeviceToHost);   // syncs the iteration
        if (he != hipSuccess) { set_error("cost readback: %s", hipGetErrorString(he)); return NMFX_ERR_HIP; }
        r->iters_run = idx + 1;
        return NMFX_OK;
    };
    // nmf.m:221-224 / cnmf.m:254-257
    auto stop = [&](int idx) {
        if (p->tolerance < 0 || idx == 0) return false;
        if (algorithm == 2) return r->cost[idx] <= r->cost[idx - 1] && r->cost[idx - 1] - r->cost[idx] <= p->tolerance;   // lnmf.m:84
        return r->cost[idx] < r->cost[idx - 1] && r->cost[idx - 1] - r->cost[idx] < p->tolerance;
    };
    bool stopped = false;
    const bool lag = e && (e->fused || e->fusedT_kl);
    for (it = 0; s == NMFX_OK && it < p->maxiter; ++it) {
        if ((s = nmfx_engine_wstep_partial(e)) != NMFX_OK) break;
        if (lag && it > 0) {
            // the fused W-step pass of iteration it also yields cost(it-1); W and H are untouched until wstep_finish, so
            // stopping here returns exactly the state of iteration it-1 (the numerators just computed are discarded)
            if ((s = read_cost(it - 1)) != NMFX_OK) break;
            if (stop(it - 1)) { stopped = true; break; }
        }
        if ((s = nmfx_engine_wstep_finish(e)) != NMFX_OK) break;
        if ((s = nmfx_engine_hstep(e)) != NMFX_OK) break;
        if (!lag) {
            if ((s = read_cost(it)) != NMFX_OK) break;
            if (stop(it)) { stopped = true; break; }
        }
    }
    if (s == NMFX_OK && lag && !stopped) {
        s = nmfx_engine_cost_pass(e);
        if (s == NMFX_OK) s = read_cost(p->maxiter - 1);
    }
    r->cost_len = r->iters_run;
    if (algorithm == 2) {   // lnmf.m:84-86 breaks WITHOUT trimming: the cost vector keeps its maxiter length, zero after the stop
        for (int i = r->iters_run; i < p->maxiter; ++i) r->cost[i] = 0.0;
        r->cost_len = p->maxiter;
    }
    if (s == NMFX_OK) s = download(st, W.as<float>(), p->dtype, r->W, mKt, stage, STAGE_ELEMS);
    if (s == NMFX_OK && pad) {
        s = repack_rows(st, H.as<float>(), K, tmp.as<float>(), Kt, p->n);
        if (s == NMFX_OK) s = download(st, tmp.as<float>(), p->dtype, r->H, Ktn, stage, STAGE_ELEMS);
        if (s == NMFX_OK && algorithm == 3) s = repack_rows(st, Z.as<float>(), K, tmp.as<float>(), Kt, nz);
        if (s == NMFX_OK && algorithm == 3) s = download(st, tmp.as<float>(), p->dtype, Z_out, (size_t)Kt * nz, stage, STAGE_ELEMS);
    } else {
        if (s == NMFX_OK) s = download(st, H.as<float>(), p->dtype, r->H, Kn, stage, STAGE_ELEMS);
        if (s == NMFX_OK && algorithm == 3) s = download(st, Z.as<float>(), p->dtype, Z_out, (size_t)K * nz, stage, STAGE_ELEMS);
    }
    nmfx_engine_destroy(e);
    return s;
}

// ---- nmfx_problem.n_gpus > 1: one process, one host thread, one stream + engine per device (what a MEX caller of nmf() needs) -------
// V and H are column-sharded over the devices, W is replicated.  Per iteration ONE exchange of the packed W-step sums
// (SURVEY 8(e)), done here without a collective library: every device reduces its own 1/N slice of `packed` straight out of its
// peers' HBM over xGMI (all links in parallel, fixed summation order), then copies the other N-1 reduced slices from their owners.
// Each slice has exactly one owner, so all replicas of W stay bit-identical.  device_ids may name one device several times
// (N shards on one GPU): that is how the 1-GPU test box exercises this path.
struct MultiDev {
    int ndev = 0;
    int dev[NMFX_MAX_GPUS];
    hipStream_t st[NMFX_MAX_GPUS] = {};
    hipEvent_t evP[NMFX_MAX_GPUS] = {}, evR[NMFX_MAX_GPUS] = {}, evG[NMFX_MAX_GPUS] = {};
    nmfx_engine *eng[NMFX_MAX_GPUS] = {};
    DevBuf V[NMFX_MAX_GPUS], W[NMFX_MAX_GPUS], H[NMFX_MAX_GPUS], ws[NMFX_MAX_GPUS], packed[NMFX_MAX_GPUS], costh[NMFX_MAX_GPUS];
    long lo[NMFX_MAX_GPUS + 1];
    ~MultiDev() {
        for (int g = 0; g < ndev; ++g) {
            (void)hipSetDevice(dev[g]);
            if (eng[g]) nmfx_engine_destroy(eng[g]);
            if (evP[g]) (void)hipEventDestroy(evP[g]);
            if (evR[g]) (void)hipEventDestroy(evR[g]);
            if (evG[g]) (void)hipEventDestroy(evG[g]);
            if (st[g]) (void)hipStreamDestroy(st[g]);
        }
    }
};

nmfx_status multi_allreduce(MultiDev &M, size_t count) {
    const int N = M.ndev;
    PeerPtrs ptrs{};
    for (int g = 0; g < N; ++g) ptrs.p[g] = M.packed[g].as<float>();
    const long per = (long)(((count + N - 1) / N + 3) & ~(size_t)3);   // slice length, a multiple of 4 floats
    auto slice = [&](int g, long *off, long *cnt) { *off = std::min((long)count, per * g); *cnt = std::min((long)count, per * (g + 1)) - *off; };
    for (int g = 0; g < N; ++g) { NMFX_HIP(hipSetDevice(M.dev[g])); NMFX_HIP(hipEventRecord(M.evP[g], M.st[g])); }
    for (int g = 0; g < N; ++g) {   // reduce-scatter: device g owns slice g
        NMFX_HIP(hipSetDevice(M.dev[g]));
        for (int h = 0; h < N; ++h) if (h != g) NMFX_HIP(hipStreamWaitEvent(M.st[g], M.evP[h], 0));
        long off, cnt;
        slice(g, &off, &cnt);
        TRY(peer_reduce(M.st[g], ptrs, N, g, off, cnt));
        NMFX_HIP(hipEventRecord(M.evR[g], M.st[g]));
    }
    for (int g = 0; g < N; ++g) {   // all-gather: fetch the slices the others reduced
        NMFX_HIP(hipSetDevice(M.dev[g]));
        for (int h = 0; h < N; ++h) {
            if (h == g) continue;
            long off, cnt;
            slice(h, &off, &cnt);
            NMFX_HIP(hipStreamWaitEvent(M.st[g], M.evR[h], 0));
            if (cnt > 0) NMFX_HIP(hipMemcpyPeerAsync(ptrs.p[g] + off, M.dev[g], ptrs.p[h] + off, M.dev[h], (size_t)cnt * 4, M.st[g]));
        }
        NMFX_HIP(hipEventRecord(M.evG[g], M.st[g]));
    }
    for (int g = 0; g < N; ++g) {   // nobody refills its `packed` (next W-step partial) before every peer has copied out of it
        NMFX_HIP(hipSetDevice(M.dev[g]));
        for (int h = 0; h < N; ++h) if (h != g) NMFX_HIP(hipStreamWaitEvent(M.st[g], M.evG[h], 0));
    }
    return NMFX_OK;
}

nmfx_status run_mu_multi(const nmfx_problem *p, nmfx_result *r, int algorithm) {
    TRY(validate_problem(p, r, false, true));
    if (algorithm != 0 && algorithm != 2) { set_error("n_gpus > 1 is implemented for nmf and lnmf (cnmf / nmfsc shard through the device-level API)"); return NMFX_ERR_UNSUPPORTED; }
    if (p->T != 1) { set_error("nmf / lnmf: T must be 1"); return NMFX_ERR_INVALID; }
    if (algorithm == 0 && p->divergence == NMFX_DIV_EUCLIDEAN_NOCOST) { set_error("nmf: unknown divergence (nmf.m:165-166)"); return NMFX_ERR_INVALID; }
    const int N = p->n_gpus;
    if (N > NMFX_MAX_GPUS || N > p->n) { set_error("n_gpus = %d: at most %d devices and one column per device", N, NMFX_MAX_GPUS); return NMFX_ERR_INVALID; }
    DeviceGuard dg_;
    MultiDev M;
    for (int g = 0; g < N; ++g) {
        M.dev[g] = p->device_ids ? p->device_ids[g] : g;
        TRY(check_device(M.dev[g]));
    }
    for (int g = 0; g < N; ++g)      // peer mappings: the reduce kernel reads the other devices' `packed` in place
        for (int h = 0; h < N; ++h) {
            if (M.dev[g] == M.dev[h]) continue;
            int can = 0;
            NMFX_HIP(hipDeviceCanAccessPeer(&can, M.dev[g], M.dev[h]));
            if (!can) { set_error("device %d cannot access device %d as a peer", M.dev[g], M.dev[h]); return NMFX_ERR_UNSUPPORTED; }
            NMFX_HIP(hipSetDevice(M.dev[g]));
            hipError_t pe = hipDeviceEnablePeerAccess(M.dev[h], 0);
            if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) { set_error("hipDeviceEnablePeerAccess(%d -> %d): %s", M.dev[g], M.dev[h], hipGetErrorString(pe)); return NMFX_ERR_HIP; }
            (void)hipGetLastError();
        }
    const int Kt = p->K_total, S = p->num_sources, dv = p->divergence;
    const long m = p->m, n = p->n;
    M.lo[0] = 0;
    for (int g = 0; g < N; ++g) M.lo[g + 1] = M.lo[g] + n / N + (g < n % N ? 1 : 0);   // contiguous column blocks, as engine.shard_columns
    long nmin = n;
    for (int g = 0; g < N; ++g) nmin = std::min(nmin, M.lo[g + 1] - M.lo[g]);
    const bool dual_ok = (dv == NMFX_DIV_IS || (dv == NMFX_DIV_AB && p->alpha != 0)) && Kt <= 128;
    const bool pad = Kt % 32 != 0 && Kt <= 256 && ((m >= 64 && nmin >= 64) || p->path == 2) && p->path != 1 && (dv == NMFX_DIV_KL || dv == NMFX_DIV_EUCLIDEAN || dual_ok);
    const int K = pad ? (Kt + 31) / 32 * 32 : Kt;
    std::vector<float> lw(K, 0.f), lh(K, 0.f);
    std::vector<uint8_t> fw(K, 0), fh(K, 0);
    for (int k = Kt; k < K; ++k) fw[k] = fh[k] = 1;
    for (int s = 0, k0 = 0; s < S; ++s) {
        const int Ks = p->K_s ? p->K_s[s] : Kt;
        for (int k = k0; k < k0 + Ks; ++k) {
            if (p->W_sparsity) lw[k] = (float)p->W_sparsity[s];
            if (p->H_sparsity) lh[k] = (float)p->H_sparsity[s];
            if (p->W_fixed) fw[k] = p->W_fixed[s];
            if (p->H_fixed) fh[k] = p->H_fixed[s];
        }
        k0 += Ks;
    }
    const size_t mK = (size_t)m * K, mKt = (size_t)m * Kt;
    size_t packed_count = 0;
    int kind = -1;
    for (int g = 0; g < N; ++g) {
        NMFX_HIP(hipSetDevice(M.dev[g]));
        M.ndev = g + 1;
        NMFX_HIP(hipStreamCreateWithFlags(&M.st[g], hipStreamNonBlocking));
        NMFX_HIP(hipEventCreateWithFlags(&M.evP[g], hipEventDisableTiming));
        NMFX_HIP(hipEventCreateWithFlags(&M.evR[g], hipEventDisableTiming));
        NMFX_HIP(hipEventCreateWithFlags(&M.evG[g], hipEventDisableTiming));
        const long nl = M.lo[g + 1] - M.lo[g];
        nmfx_engine_desc d{};
        d.m = m; d.n_local = nl; d.K_total = K; d.T = 1; d.divergence = dv; d.alpha = p->alpha; d.beta = p->beta;
        d.lamW_col = lw.data(); d.lamH_row = lh.data(); d.fixW_col = fw.data(); d.fixH_row = fh.data();
        d.device = M.dev[g]; d.stream = M.st[g]; d.algorithm = algorithm; d.path = p->path; d.K_valid = pad ? Kt : 0; d.col_offset = M.lo[g];
        size_t wsb = 0, pc = 0;
        TRY(nmfx_engine_workspace_bytes(&d, &wsb));
        TRY(nmfx_engine_packed_count(&d, &pc));
        if (g == 0) packed_count = pc;
        else if (pc != packed_count) { set_error("n_gpus: shards disagree on the packed layout"); return NMFX_ERR_INVALID; }
        DevBuf stage, tmp;
        TRY(M.V[g].alloc((size_t)m * nl * 4)); TRY(M.W[g].alloc(mK * 4)); TRY(M.H[g].alloc((size_t)K * nl * 4)); TRY(M.ws[g].alloc(wsb));
        TRY(M.packed[g].alloc(pc * 4)); TRY(M.costh[g].alloc(64)); TRY(stage.alloc(STAGE_ELEMS * 8));
        const char *Vh = static_cast<const char *>(p->V) + (size_t)m * M.lo[g] * dsize(p->dtype);           // a column block is a contiguous slab
        const char *Hh = static_cast<const char *>(p->H_init) + (size_t)Kt * M.lo[g] * dsize(p->dtype);
        TRY(upload(M.st[g], Vh, p->dtype, M.V[g].as<float>(), (size_t)m * nl, 1.0, stage, STAGE_ELEMS));
        TRY(upload(M.st[g], p->W_init, p->dtype, M.W[g].as<float>(), mKt, 1.0, stage, STAGE_ELEMS));
        if (pad) {
            NMFX_HIP(hipMemsetAsync(M.W[g].as<float>() + mKt, 0, (mK - mKt) * 4, M.st[g]));
            TRY(tmp.alloc((size_t)Kt * nl * 4));
            TRY(upload(M.st[g], Hh, p->dtype, tmp.as<float>(), (size_t)Kt * nl, 1.0, stage, STAGE_ELEMS));
            TRY(repack_rows(M.st[g], tmp.as<float>(), Kt, M.H[g].as<float>(), K, nl));
            NMFX_HIP(hipStreamSynchronize(M.st[g]));
        } else TRY(upload(M.st[g], Hh, p->dtype, M.H[g].as<float>(), (size_t)K * nl, 1.0, stage, STAGE_ELEMS));
        TRY(nmfx_engine_create(&d, M.V[g].as<float>(), M.W[g].as<float>(), M.H[g].as<float>(), M.ws[g].p, wsb, M.packed[g].as<float>(), &M.eng[g]));
        TRY(nmfx_engine_set_rank0(M.eng[g], g == 0));
        const int kd = nmfx_engine_is_fused(M.eng[g]);
        if (kind < 0) kind = kd;
        else if (kd != kind) { set_error("n_gpus: shards picked different kernel paths; pass path = 1"); return NMFX_ERR_UNSUPPORTED; }
        TRY(nmfx_engine_init(M.eng[g]));
    }
    const bool lag = kind == 1;
    std::vector<double> hc(N);
    auto read_cost = [&](int idx) -> nmfx_status {   // cost = sum of the shards' partials (the lambda*|W| term lives on device 0 only)
        for (int g = 0; g < N; ++g) {
            NMFX_HIP(hipSetDevice(M.dev[g]));
            NMFX_HIP(hipMemcpyAsync(&hc[g], M.eng[g]->cost, sizeof(double), hipMemcpyDeviceToHost, M.st[g]));
        }
        double c = 0.0;
        for (int g = 0; g < N; ++g) { NMFX_HIP(hipSetDevice(M.dev[g])); NMFX_HIP(hipStreamSynchronize(M.st[g])); c += hc[g]; }
        r->cost[idx] = c;
        r->iters_run = idx + 1;
        return NMFX_OK;
    };
    auto stop = [&](int idx) {
        if (p->tolerance < 0 || idx == 0) return false;
        if (algorithm == 2) return r->cost[idx] <= r->cost[idx - 1] && r->cost[idx - 1] - r->cost[idx] <= p->tolerance;   // lnmf.m:84
        return r->cost[idx] < r->cost[idx - 1] && r->cost[idx - 1] - r->cost[idx] < p->tolerance;                         // nmf.m:221
    };
    r->iters_run = 0;
    bool stopped = false;
    for (int it = 0; it < p->maxiter; ++it) {
        for (int g = 0; g < N; ++g) TRY(nmfx_engine_wstep_partial(M.eng[g]));
        if (lag && it > 0) {
            TRY(read_cost(it - 1));
            if (stop(it - 1)) { stopped = true; break; }
        }
        TRY(multi_allreduce(M, packed_count));
        for (int g = 0; g < N; ++g) { TRY(nmfx_engine_wstep_finish(M.eng[g])); TRY(nmfx_engine_hstep(M.eng[g])); }
        if (!lag) {
            TRY(read_cost(it));
            if (stop(it)) { stopped = true; break; }
        }
    }
    if (lag && !stopped) {
        for (int g = 0; g < N; ++g) TRY(nmfx_engine_cost_pass(M.eng[g]));
        TRY(read_cost(p->maxiter - 1));
    }
    r->cost_len = r->iters_run;
    if (algorithm == 2) {
        for (int i = r->iters_run; i < p->maxiter; ++i) r->cost[i] = 0.0;
        r->cost_len = p->maxiter;
    }
    for (int g = 0; g < N; ++g) {
        NMFX_HIP(hipSetDevice(M.dev[g]));
        const long nl = M.lo[g + 1] - M.lo[g];
        DevBuf stage, tmp;
        TRY(stage.alloc(STAGE_ELEMS * 8));
        if (g == 0) TRY(download(M.st[g], M.W[g].as<float>(), p->dtype, r->W, mKt, stage, STAGE_ELEMS));
        char *Hh = static_cast<char *>(r->H) + (size_t)Kt * M.lo[g] * dsize(p->dtype);
        if (pad) {
            TRY(tmp.alloc((size_t)Kt * nl * 4));
            TRY(repack_rows(M.st[g], M.H[g].as<float>(), K, tmp.as<float>(), Kt, nl));
            TRY(download(M.st[g], tmp.as<float>(), p->dtype, Hh, (size_t)Kt * nl, stage, STAGE_ELEMS));
        } else TRY(download(M.st[g], M.H[g].as<float>(), p->dtype, Hh, (size_t)K * nl, stage, STAGE_ELEMS));
    }
    return NMFX_OK;
}

// 0.5*||V - V_hat||^2 from the per-block partials of an EPI_COST GEMM (host double)
// The line searches read every objective on the host (nmfsc.m:164,215 decide on it).  The finishing kernel publishes the value into a
// pinned, device-mapped slot followed by a sequence number, and the host thread polls that number: no hipStreamSynchronize (its wake-up
// cost ~0.1 ms per evaluation, 8 % of a config-5 iteration) and no device-to-host copy.
struct PinnedSlot {   // 64 bytes per host thread, kept for the life of the process (freeing it from a thread_local destructor would race the runtime's own teardown)
    double *host = nullptr, *dev = nullptr;
    unsigned long long seq = 0;
    nmfx_status get() {
        if (host) return NMFX_OK;
        NMFX_HIP(hipHostMalloc(reinterpret_cast<void **>(&host), 64, hipHostMallocMapped));
        memset(host, 0, 64);
        NMFX_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&dev), host, 0));
        return NMFX_OK;
    }
};
static thread_local PinnedSlot g_obj_slot;
nmfx_status read_obj(hipStream_t st, const double *partials, int count, double *cost_dev, double *out, Comm *comm = nullptr) {
    TRY(g_obj_slot.get());
    PinnedSlot &sl = g_obj_slot;
    const unsigned long long seq = ++sl.seq;
    if (comm && comm->active()) {
        TRY(publish_obj(st, partials, count, 0.5, nullptr, cost_dev, nullptr, 0));
        TRY(comm->allreduce(cost_dev, 1, NMFX_F64, NMFX_REDUCE_SUM));   // column shards: the objective is a sum over ranks
        TRY(publish_obj(st, nullptr, 0, 1.0, cost_dev, nullptr, sl.dev, seq));
    } else TRY(publish_obj(st, partials, count, 0.5, nullptr, cost_dev, sl.dev, seq));
    const volatile unsigned long long *flag = reinterpret_cast<const volatile unsigned long long *>(sl.host) + 1;
    for (unsigned long spin = 1;; ++spin) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) break;
        if ((spin & 0x3fff) == 0) {   // a failed launch or a fault must not spin forever
            const hipError_t q = hipStreamQuery(st);
            if (q == hipSuccess) { NMFX_HIP(hipStreamSynchronize(st)); break; }
            if (q != hipErrorNotReady) { set_error("objective read-back: %s", hipGetErrorString(q)); return NMFX_ERR_HIP; }
        }
    }
    *out = *reinterpret_cast<const volatile double *>(sl.host);
    return NMFX_OK;
}

// nmfsc launch groups (bench.py --workload c5)
enum ScTag { SC_OBJ = 0, SC_PROJ = 1, SC_HTERMS = 2, SC_WTERMS = 3, SC_SMALL = 4, SC_COUNT = 5 };
static const char *const kScTagNames[SC_COUNT] = {"fused:objective pass (S=W*H -> 0.5||V-S||^2)", "projfunc (Hoyer projection of the rows of H)",
                                                  "H-step terms (sparse H: fused residual pass dH = W'*(W*H-V) + objective; MU: W'*V, (W'*W)*H)", "W-step terms (MU: V*H', W*(H*H'); sparse W: fused residual pass)", "small kernels (transposes, updates)"};
static thread_local Profiler g_sc_prof;

// device-resident inputs of nmfx_nmfsc_dev: a column shard per rank, W replicated, collectives through the caller's callback
struct ScDev {
    const float *V;      // m x n_local, already divided by the GLOBAL max (nmfsc.m:62)
    float *W, *H;        // in/out
    long n_total;        // global column count (L1s, nmfsc.m:102-106, is defined on whole rows of H)
    Comm comm;
    hipStream_t st;
};

// nmfsc.m:57-245
nmfx_status run_nmfsc(const nmfx_problem *p, nmfx_result *r, ScDev *dev = nullptr) {
    if (!dev) TRY(validate_problem(p, r, true));
    if (p->T != 1) { set_error("nmfsc: T must be 1"); return NMFX_ERR_INVALID; }
    const long m = p->m, n = p->n;
    // any K <= 256 runs on the fused kernels: K is rounded up to a multiple of 32 with zero columns of W / zero rows of H.  They add exact
    // zeros to W*H, to every gradient and to every Gram product, stay zero under both update rules, and are kept away from the only two
    // places that would resurrect them: projfunc (a zero vector does NOT project to zero) and the row-norm rescale of nmfsc.m:185-187 (0/0)
    const int Kv = p->K_total;
    const bool padK = p->path != 1 && Kv % 32 != 0 && Kv <= 256 && ((m >= 64 && n >= 64) || p->path == 2);
    const int K = padK ? (Kv + 31) / 32 * 32 : Kv;
    const size_t mn = (size_t)m * n, mK = (size_t)m * K, Kn = (size_t)K * n, mKv = (size_t)m * Kv, Kvn = (size_t)Kv * n;
    Comm nocomm{};
    Comm &comm = dev ? dev->comm : nocomm;
    double vmin = INFINITY, vmax = -INFINITY;   // nmfsc.m:57-62
    if (dev) { vmin = 0; vmax = 1; }
    else if (p->dtype == NMFX_F64) { const double *v = static_cast<const double *>(p->V); for (size_t i = 0; i < mn; ++i) { if (v[i] < vmin) vmin = v[i]; if (v[i] > vmax) vmax = v[i]; } }
    else { const float *v = static_cast<const float *>(p->V); for (size_t i = 0; i < mn; ++i) { if (v[i] < vmin) vmin = v[i]; if (v[i] > vmax) vmax = v[i]; } }
    if (vmin < 0) { set_error("Negative values in data!"); return NMFX_ERR_NEGATIVE; }
    DeviceGuard dg_;
    TRY(check_device(p->device));
    hipStream_t st = dev ? dev->st : nullptr;
    g_sc_prof.st = st;
    if (g_sc_prof.on) { g_sc_prof.events.clear(); g_sc_prof.pool_used = 0; }
    Profiler *pf = &g_sc_prof;
    const long n_total = dev ? dev->n_total : n;
    double sW = p->sc_W_sparsity, sH = p->sc_H_sparsity;
    double L1a = 0, L1s = 0;
    if (sW > 0) { if (sW > 1) sW = 1; L1a = std::sqrt((double)m) - (std::sqrt((double)m) - 1) * sW; }   // nmfsc.m:89-93
    if (sH > 0) { if (sH > 1) sH = 1; L1s = std::sqrt((double)n_total) - (std::sqrt((double)n_total) - 1) * sH; }   // nmfsc.m:102-106
    const bool fixW = p->W_fixed && p->W_fixed[0], fixH = p->H_fixed && p->H_fixed[0];

    DevBuf V, W, Hk, HT, HnT, G1, G2, Vh, Wn, stage, part, costd, scratch, pfv, pff, pfr;
    const bool fast = p->path != 1 && fused_supported(K) && ((m >= 64 && n >= 64) || p->path == 2);   // ragged m / n: masked-edge kernels
    if (p->path == 2 && !fast) { set_error("nmfsc: fused path requested but shape not eligible"); return NMFX_ERR_UNSUPPORTED; }
    if (comm.active() && !fast) {
        set_error("nmfsc on column shards runs on the fused kernels only: K <= 256, m and n_local >= 64");
        return NMFX_ERR_UNSUPPORTED;
    }
    if (!dev) TRY(V.alloc(mn * 4));
    if (!fast) TRY(Vh.alloc(mn * 4));
    if (comm.active() && sH > 0) { TRY(pfv.alloc(Kn * 8)); TRY(pff.alloc(Kn)); TRY(pfr.alloc(sizeof(double) * 6 * K + 64)); }
    TRY(W.alloc(mK * 4)); TRY(Wn.alloc(mK * 4)); TRY(Hk.alloc(Kn * 4)); TRY(HT.alloc(Kn * 4));
    TRY(HnT.alloc(Kn * 4));
    const size_t gmax = (Kn > mK ? Kn : mK) + (size_t)K * K;   // + K*K: [V*H' | H*H'] travel as ONE all-reduce on column shards
    TRY(G1.alloc(gmax * 4)); TRY(G2.alloc(gmax * 4));
    TRY(stage.alloc(STAGE_ELEMS * 8));
    const int nparts = (int)gemm_grid_blocks(m, n);
    TRY(part.alloc(sizeof(double) * nparts)); TRY(costd.alloc(64 + sizeof(double) * K));
    size_t sb = gemm_scratch_bytes(n, K, m), sb2 = gemm_scratch_bytes(m, K, n), sb3 = gemm_scratch_bytes(K, n, m);
    if (sb2 > sb) sb = sb2;
    if (sb3 > sb) sb = sb3;
    TRY(scratch.alloc(sb));
    DevBuf hpk;   // Kv x n staging of the un-padded, row-interleaved H
    if (padK) {
        TRY(hpk.alloc(Kvn * 4));
        NMFX_HIP(hipMemsetAsync(W.p, 0, mK * 4, st)); NMFX_HIP(hipMemsetAsync(Wn.p, 0, mK * 4, st));
        NMFX_HIP(hipMemsetAsync(HT.p, 0, Kn * 4, st)); NMFX_HIP(hipMemsetAsync(HnT.p, 0, Kn * 4, st));   // the padding of every buffer that is only ever
    }                                                                                                 // written through projfunc stays zero
    if (!dev) {
        TRY(upload(st, p->V, p->dtype, V.as<float>(), mn, vmax, stage, STAGE_ELEMS));   // V = V / max(V(:))
        TRY(upload(st, p->W_init, p->dtype, W.as<float>(), mKv, 1.0, stage, STAGE_ELEMS));   // the first Kv columns of the m x K array
        if (padK) {
            TRY(upload(st, p->H_init, p->dtype, hpk.as<float>(), Kvn, 1.0, stage, STAGE_ELEMS));
            TRY(repack_rows(st, hpk.as<float>(), Kv, Hk.as<float>(), K, n));
        } else TRY(upload(st, p->H_init, p->dtype, Hk.as<float>(), Kn, 1.0, stage, STAGE_ELEMS));
    } else {
        NMFX_HIP(hipMemcpyAsync(W.p, dev->W, mKv * 4, hipMemcpyDeviceToDevice, st));
        if (padK) TRY(repack_rows(st, dev->H, Kv, Hk.as<float>(), K, n));
        else NMFX_HIP(hipMemcpyAsync(Hk.p, dev->H, Kn * 4, hipMemcpyDeviceToDevice, st));
    }
    const float *Vp = dev ? dev->V : V.as<float>();
    float *Wd = W.as<float>(), *Wnew = Wn.as<float>(), *HTd = HT.as<float>(), *HnewT = HnT.as<float>();
    // rows of H (stored as the columns of an n_local x K transposed copy) through projfunc; on column shards every reduction of
    // projfunc.m:22-53 is a sum over ranks
    auto project_H = [&](float *HxT) -> nmfx_status {
        if (comm.active()) return projfunc_cols_dist(st, HxT, n, Kv, n_total, L1s, 1.0, 1, comm, pfv.as<double>(), pff.as<unsigned char>(), pfr.as<double>());
        return projfunc_cols(st, HxT, n, Kv, L1s, 1.0, 1, nullptr);
    };
    // out = projection of (base + mu*dir), rows of H as the columns of the n x K transposed copies   (nmfsc.m:154-157)
    // (dir64: the direction as doubles, small K; the step is formed in fp64 while loading)
    auto step_project_H = [&](const float *baseT, const float *dirT, const double *dir64, double mu, float *outT) -> nmfx_status {
        if (comm.active())
            return projfunc_cols_dist(st, outT, n, Kv, n_total, L1s, 1.0, 1, comm, pfv.as<double>(), pff.as<unsigned char>(), pfr.as<double>(), dirT, mu, baseT, dir64);
        PScope ps(pf, SC_PROJ);
        return projfunc_cols(st, outT, n, Kv, L1s, 1.0, 1, nullptr, dirT, mu, baseT, dir64);
    };
    TRY(transpose_f32(st, Hk.as<float>(), K, n, HTd));
    const bool resume = dev && p->sc_resume;   // W / H are the state a previous call left: already projected, nothing to initialise
    if (sW > 0 && !resume) TRY(projfunc_cols(st, Wd, m, Kv, L1a, 1.0, 1, nullptr));     // nmfsc.m:94-96  (W is replicated: every rank projects the same columns)
    if (sH > 0 && !resume) TRY(project_H(HTd));                                        // nmfsc.m:107-109

    // V_hat = Wx * Hx (Hx given transposed, n x K) with the residual objective; returns 0.5*||V - V_hat||^2
    auto recon_obj = [&](const float *Wx, const float *HxT, double *obj) -> nmfx_status {
        GemmParams g;
        memset(&g, 0, sizeof(g));
        g.M = m; g.N = n; g.Kc = K;
        g.A = OpView{Wx, nullptr, m, VIEW_RC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f};
        g.B = OpView{HxT, nullptr, n, VIEW_RC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f};
        g.C = Vh.as<float>(); g.ldc = m; g.epi = EPI_COST; g.store_c = 1; g.cost_div = NMFX_DIV_EUCLIDEAN; g.Vref = Vp; g.ldv = m;
        g.cost_partials = part.as<double>(); g.splitk = 1;
        long blocks = 0;
        TRY(launch_gemm(st, g, &blocks));
        return read_obj(st, part.as<double>(), (int)blocks, costd.as<double>(), obj);
    };
    // outT (n x K) = f(V, V_hat)' * W
    auto xt_w = [&](const float *x, const float *x2, int func, float *outT) -> nmfx_status {
        GemmParams g;
        memset(&g, 0, sizeof(g));
        g.M = n; g.N = K; g.Kc = m;
        g.A = OpView{x, x2, m, VIEW_KC, 0, 0, 0, func, 0.f, 0.f};
        g.B = OpView{Wd, nullptr, m, VIEW_KC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f};
        g.C = outT; g.ldc = n; g.epi = EPI_STORE; g.splitk = 1;
        return gemm_auto(st, g, scratch.p, sb);
    };
    // out (m x K) = f(V, V_hat) * H'
    auto x_ht = [&](const float *x, const float *x2, int func, float *out) -> nmfx_status {
        GemmParams g;
        memset(&g, 0, sizeof(g));
        g.M = m; g.N = K; g.Kc = n;
        g.A = OpView{x, x2, m, VIEW_RC, 0, 0, 0, func, 0.f, 0.f};
        g.B = OpView{HTd, nullptr, n, VIEW_KC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f};
        g.C = out; g.ldc = m; g.epi = EPI_STORE; g.splitk = 1;
        return gemm_auto(st, g, scratch.p, sb);
    };

    // ---- fast path: the same algorithm on the fused kernels (V_hat never materialised) -----------------------------------
    //   objective         0.5*||V - W*H||^2          fused cost-only pass (S = W*H in registers)
    //   W'*V, V*H'        fused H-step / W-step passes with R = V
    //   W'*V_hat, V_hat*H' (W'*W)*H and W*(H*H')     K x K Gram products
    DevBuf WTb, slabs, Gb, Denb, KKb, fparts, g64h, g64w, s64;
    const bool smallk = fast && Kv <= smallk_max();   // a handful of components: gradients + objective in fp64 (aux.hip::smallk_grad), handed to projfunc as doubles
    // ... and small problems with any K (m*n*K <= 2^27 fp64 FMAs per evaluation), unless the fused kernels are asked for by name (path 2):
    // R64 = W*H - V as doubles, both contractions on it in fp64 (aux.hip::resid64 / r64_wt / r64_ht).  What this buys is parity -- the Hoyer
    // projection amplifies the accumulation noise of an fp32 MFMA contraction 10-100x on short vectors
    const bool small64 = fast && !smallk && p->path != 2 && (double)m * (double)n_total * (double)Kv <= (double)(1 << 27);   // n_total: every rank of a sharded run must take the same path (the all-reduced buffers differ)
    const bool use64 = smallk || small64;
    const int nch_w64 = small64 ? (int)std::min<long>(std::max<long>(1, 1024 / (((m + 255) / 256) * Kv)), (n + 63) / 64) : 1;
    DevBuf r64b;
    int nsplit_w = 1, isplit_h = 1;
    long cps_w = n, cps_h = m;
    if (fast) {
        nsplit_w = fused_split((m + 127) / 128, n, K, &cps_w);
        isplit_h = fused_split((n + 127) / 128, m, K, &cps_h);
        TRY(WTb.alloc(mK * 4)); TRY(slabs.alloc(std::max((size_t)nsplit_w * mK, (size_t)isplit_h * Kn) * 4)); TRY(Gb.alloc(Kn * 4)); TRY(Denb.alloc(Kn * 4));
        TRY(KKb.alloc((size_t)K * K * 4)); TRY(fparts.alloc(sizeof(double) * std::max<long>(std::max(((m + 127) / 128) * nsplit_w, ((n + 127) / 128) * isplit_h), smallk ? smallk_partials(m, n) : (small64 ? resid64_blocks(m, n) : 0))));
        if (use64) { TRY(g64h.alloc(sizeof(double) * n * Kv)); TRY(g64w.alloc(sizeof(double) * m * Kv)); TRY(s64.alloc(sizeof(double) * (size_t)(smallk ? smallk_dw_chunks(m, n) : nch_w64) * m * Kv)); }
        if (small64) TRY(r64b.alloc(sizeof(double) * mn));
    }
    // 0.5*||V - Wx*Hx||^2 with Hx given as K x n (column-major)
    auto fast_obj = [&](const float *Wx, const float *Hx, double *obj) -> nmfx_status {
        if (use64) {
            int np_ = 0;
            {
                PScope ps(pf, SC_OBJ);
                if (smallk) TRY(smallk_grad(st, Kv, Vp, m, n, Wx, Hx, K, nullptr, nullptr, nullptr, fparts.as<double>(), &np_));
                else TRY(resid64(st, Vp, m, n, Wx, Hx, Kv, K, nullptr, fparts.as<double>(), &np_));
            }
            return read_obj(st, fparts.as<double>(), np_, costd.as<double>(), obj, &comm);
        }
        FusedParams f;
        memset(&f, 0, sizeof(f));
        f.X = Wx; f.xs_r = 1; f.xs_k = m; f.Y = Hx; f.D = Vp; f.ldd = m; f.R = m; f.Cn = n; f.K = K; f.c_per_split = cps_w;
        f.cost_partials = fparts.as<double>();
        {
            PScope ps(pf, SC_OBJ);
            TRY(launch_fused(st, f, nsplit_w, true, 1, false, 0));
        }
        return read_obj(st, fparts.as<double>(), (int)(((m + 127) / 128) * nsplit_w), costd.as<double>(), obj, &comm);
    };
    // The gradients of the line-search branches in RESIDUAL form, one fused pass each (func 6: S = W*H in registers -> R = S - V -> second
    // contraction): dH = W'*(W*H - V) (nmfsc.m:144-148) and dW = (W*H - V)*H' (nmfsc.m:194-200).  The Gram form pos - neg = (W'W)H - W'V
    // subtracts two products rounded separately, and their difference goes to zero as the fit converges while they do not: it cost parity
    // on small K (scripts/fuzz_campaign_sc.py: H off by 1.3e-5 at K = 3).  The same pass yields 0.5*||V - W*H||^2 of the point it is taken at.
    // Den (K x n) = Wx' * (Wx*Hx - V); *obj = the objective at (Wx, Hx) when asked for
    auto resid_h = [&](const float *Wx, const float *Hx, double *obj) -> nmfx_status {
        if (use64) {   // g64h = dH' (n x Kv doubles)
            int np_ = 0;
            {
                PScope ps(pf, SC_HTERMS);
                if (smallk) TRY(smallk_grad(st, Kv, Vp, m, n, Wx, Hx, K, g64h.as<double>(), nullptr, nullptr, fparts.as<double>(), &np_));
                else {
                    TRY(resid64(st, Vp, m, n, Wx, Hx, Kv, K, r64b.as<double>(), fparts.as<double>(), &np_));
                    TRY(r64_wt(st, r64b.as<double>(), m, n, Wx, Kv, g64h.as<double>()));
                }
            }
            return obj ? read_obj(st, fparts.as<double>(), np_, costd.as<double>(), obj, &comm) : NMFX_OK;
        }
        {
            PScope ps(pf, SC_HTERMS);
            TRY(transpose_f32(st, Wx, m, K, WTb.as<float>()));
            FusedParams f;
            memset(&f, 0, sizeof(f));
            f.X = Hx; f.xs_r = K; f.xs_k = 1; f.Y = WTb.as<float>(); f.D = Vp; f.ldd = m; f.R = n; f.Cn = m; f.K = K; f.c_per_split = cps_h;
            f.out = isplit_h == 1 ? Denb.as<float>() : slabs.as<float>(); f.slab_stride = (long)K * n; f.os_r = K; f.os_k = 1;
            f.cost_partials = fparts.as<double>();
            TRY(launch_fused(st, f, isplit_h, false, 6, true, 0));
            if (isplit_h > 1) TRY(reduce_slabs(st, slabs.as<float>(), isplit_h, f.slab_stride, f.slab_stride, Denb.as<float>(), 0));
        }
        if (!obj) return NMFX_OK;
        return read_obj(st, fparts.as<double>(), (int)(((n + 127) / 128) * isplit_h), costd.as<double>(), obj, &comm);
    };
    // dW_ (m x K) = (Wx*Hx - V) * Hx', summed over the column shards; *obj as above
    // (reduce = false: the sum over the column shards is left to the caller -- a speculative evaluation inside the H line search, see below)
    auto resid_w = [&](const float *Wx, const float *Hx, float *dW_, double *obj, bool reduce = true) -> nmfx_status {
        if (use64) {   // g64w = dW (m x Kv doubles)
            int np_ = 0;
            {
                PScope ps(pf, SC_WTERMS);
                if (smallk) TRY(smallk_grad(st, Kv, Vp, m, n, Wx, Hx, K, nullptr, g64w.as<double>(), s64.as<double>(), fparts.as<double>(), &np_));
                else {
                    TRY(resid64(st, Vp, m, n, Wx, Hx, Kv, K, r64b.as<double>(), fparts.as<double>(), &np_));
                    TRY(r64_ht(st, r64b.as<double>(), m, n, Hx, Kv, K, s64.as<double>(), nch_w64, g64w.as<double>()));
                }
                if (comm.active() && reduce) TRY(comm.allreduce(g64w.p, (long)m * Kv, NMFX_F64, NMFX_REDUCE_SUM));
            }
            return obj ? read_obj(st, fparts.as<double>(), np_, costd.as<double>(), obj, &comm) : NMFX_OK;
        }
        {
            PScope ps(pf, SC_WTERMS);
            FusedParams f;
            memset(&f, 0, sizeof(f));
            f.X = Wx; f.xs_r = 1; f.xs_k = m; f.Y = Hx; f.D = Vp; f.ldd = m; f.R = m; f.Cn = n; f.K = K; f.c_per_split = cps_w;
            f.out = nsplit_w == 1 ? dW_ : slabs.as<float>(); f.slab_stride = (long)m * K; f.os_r = 1; f.os_k = m;
            f.cost_partials = fparts.as<double>();
            TRY(launch_fused(st, f, nsplit_w, true, 6, true, 0));
            if (nsplit_w > 1) TRY(reduce_slabs(st, slabs.as<float>(), nsplit_w, f.slab_stride, f.slab_stride, dW_, 0));
            if (comm.active() && reduce) TRY(comm.allreduce(dW_, (long)mK, NMFX_F32, NMFX_REDUCE_SUM));   // the ONE large exchange of an outer iteration
        }
        if (!obj) return NMFX_OK;
        return read_obj(st, fparts.as<double>(), (int)(((m + 127) / 128) * nsplit_w), costd.as<double>(), obj, &comm);
    };
    auto kk_gemm = [&](long M_, long N_, long Kc_, OpView A_, OpView B_, float *C_, long ldc_) -> nmfx_status {
        GemmParams g;
        memset(&g, 0, sizeof(g));
        g.M = M_; g.N = N_; g.Kc = Kc_; g.A = A_; g.B = B_; g.C = C_; g.ldc = ldc_; g.epi = EPI_STORE; g.splitk = 1;
        return gemm_auto(st, g, scratch.p, sb);
    };
    // G (K x n) = Wx' * V and Den (K x n) = (Wx'*Wx) * Hx      (Hx: K x n)
    auto fast_h_terms = [&](const float *Wx, const float *Hx) -> nmfx_status {
        PScope ps(pf, SC_HTERMS);
        static const bool sc_fused_env = getenv("NMFX_SC_FUSED_HTERMS") != nullptr;   // dev switch: A/B
        const bool sc_fused_terms = sc_fused_env || K % 64 != 0;   // the GEMM is only pipelined for tile-aligned outputs
        if (sc_fused_terms) {
        TRY(transpose_f32(st, Wx, m, K, WTb.as<float>()));
        FusedParams f;
        memset(&f, 0, sizeof(f));
        f.X = Hx; f.xs_r = K; f.xs_k = 1; f.Y = WTb.as<float>(); f.D = Vp; f.ldd = m; f.R = n; f.Cn = m; f.K = K; f.c_per_split = cps_h;
        f.out = isplit_h == 1 ? Gb.as<float>() : slabs.as<float>(); f.slab_stride = (long)K * n; f.os_r = K; f.os_k = 1;
        TRY(launch_fused(st, f, isplit_h, false, 0, true, 0));
        if (isplit_h > 1) TRY(reduce_slabs(st, slabs.as<float>(), isplit_h, f.slab_stride, f.slab_stride, Gb.as<float>(), 0));
        } else
        // W'*V has no first product: the pipelined GEMM beats the register-stationary kernel on a plain contraction
        TRY(kk_gemm(K, n, m, OpView{Wx, nullptr, m, VIEW_KC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f}, OpView{Vp, nullptr, m, VIEW_KC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f},
                    Gb.as<float>(), K));
        TRY(kk_gemm(K, K, m, OpView{Wx, nullptr, m, VIEW_KC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f}, OpView{Wx, nullptr, m, VIEW_KC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f}, KKb.as<float>(), K));
        return kk_gemm(K, n, K, OpView{KKb.as<float>(), nullptr, (long)K, VIEW_RC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f},
                       OpView{Hx, nullptr, (long)K, VIEW_KC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f}, Denb.as<float>(), K);
    };
    // N (m x K) = V * Hx' and P (m x K) = Wx * (Hx*Hx'); N_ has room for K*K more floats: [N | Hx*Hx'] is what column shards sum
    auto fast_w_terms = [&](const float *Wx, const float *Hx, float *N_, float *P_) -> nmfx_status {
        PScope ps(pf, SC_WTERMS);
        float *KK = N_ + mK;
        // V*H' (N = K wide): the register-stationary kernel (R = V, two workgroups per CU at K <= 128) beats the split-K GEMM here
        static const bool sc_fused_terms = getenv("NMFX_SC_GEMM_WTERMS") == nullptr;   // dev switch: A/B
        if (sc_fused_terms) {
        FusedParams f;
        memset(&f, 0, sizeof(f));
        f.X = Wx; f.xs_r = 1; f.xs_k = m; f.Y = Hx; f.D = Vp; f.ldd = m; f.R = m; f.Cn = n; f.K = K; f.c_per_split = cps_w;
        f.out = nsplit_w == 1 ? N_ : slabs.as<float>(); f.slab_stride = (long)m * K; f.os_r = 1; f.os_k = m;
        TRY(launch_fused(st, f, nsplit_w, true, 0, true, 0));
        if (nsplit_w > 1) TRY(reduce_slabs(st, slabs.as<float>(), nsplit_w, f.slab_stride, f.slab_stride, N_, 0));
        } else
        TRY(kk_gemm(m, K, n, OpView{Vp, nullptr, m, VIEW_RC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f}, OpView{Hx, nullptr, (long)K, VIEW_RC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f},
                    N_, m));
        TRY(kk_gemm(K, K, n, OpView{Hx, nullptr, (long)K, VIEW_RC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f}, OpView{Hx, nullptr, (long)K, VIEW_RC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f}, KK, K));
        if (comm.active()) TRY(comm.allreduce(N_, (long)(mK + (size_t)K * K), NMFX_F32, NMFX_REDUCE_SUM));   // the ONE large exchange of an outer iteration
        return kk_gemm(m, K, K, OpView{Wx, nullptr, m, VIEW_RC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f},
                       OpView{KK, nullptr, (long)K, VIEW_KC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f}, P_, m);
    };
    double stepH = p->sc_stepsize_H0 > 0 ? p->sc_stepsize_H0 : 1.0, stepW = p->sc_stepsize_W0 > 0 ? p->sc_stepsize_W0 : 1.0;   // nmfsc.m:133-134
    if (fast) {
        DevBuf Hcb;
        TRY(Hcb.alloc(Kn * 4));
        float *Hcur = Hk.as<float>(), *Hcand = Hcb.as<float>();
        double *nrm2 = costd.as<double>() + 8;
        TRY(transpose_f32(st, HTd, n, K, Hcur));
        // the objective of (W, H) and the gradient dH the next sparse-H line search starts from come out of the same pass
        const bool lsH = !fixH && sH > 0;
        // With BOTH line searches active every objective evaluation is made by the residual pass of the OTHER factor: the try that is
        // accepted (4 of 5 are) has then already produced the gradient the next line search starts from, and a 4*mnK pass per search is gone.
        static const bool no_spec = getenv("NMFX_SC_NO_SPEC") != nullptr;   // dev switch (A/B runs)
        const bool spec = lsH && !fixW && sW > 0 && !no_spec;
        bool have_dW = false;   // G2 (g64w) = dW of the current (Wd, Hcur), not yet summed over the shards
        bool have_dH = false;   // Denb = dH of the current (Wd, Hcur)
        if (lsH && p->maxiter >= 1) { TRY(resid_h(Wd, Hcur, &r->cost[0])); have_dH = true; }
        else TRY(fast_obj(Wd, Hcur, &r->cost[0]));                                              // nmfsc.m:138-139
        int ncost = p->maxiter + 1, nH = 0, nW = 0;
        bool early = false;
        for (int it = 1; it <= p->maxiter && !early; ++it) {
            double cur_obj = r->cost[it - 1];
            if (!fixH) {
                if (sH > 0) {
                    if (!have_dH) TRY(resid_h(Wd, Hcur, nullptr));                              // dH = W'*V_hat - W'*V   nmfsc.m:144-148
                    have_dH = false;
                    if (!use64) {
                        PScope ps(pf, SC_SMALL);
                        TRY(transpose_f32(st, Denb.as<float>(), K, n, G1.as<float>()));         // dH' (n x K): rows of H are contiguous there
                    }
                    const double begobj = cur_obj;                                              // nmfsc.m:149
                    int tries = 0;
                    double newobj = 0;
                    for (;;) {
                        ++tries;
                        TRY(step_project_H(HTd, use64 ? nullptr : G1.as<float>(), use64 ? g64h.as<double>() : nullptr, -stepH, HnewT));   // nmfsc.m:154-157
                        {
                            PScope ps(pf, SC_SMALL);
                            TRY(transpose_f32(st, HnewT, n, K, Hcand));
                        }
                        if (spec) TRY(resid_w(Wd, Hcand, G2.as<float>(), &newobj, false));          // nmfsc.m:160-161 (+ dW at the candidate)
                        else TRY(fast_obj(Wd, Hcand, &newobj));
                        if (newobj <= begobj) break;                                                // nmfsc.m:164
                        stepH /= 2;                                                                 // nmfsc.m:169
                        if (stepH < 1e-200) { early = true; break; }                                // nmfsc.m:170-174
                    }
                    if (r->tries_H) r->tries_H[nH] = tries;
                    ++nH;
                    if (early) { ncost = it; break; }
                    stepH *= 1.2;                                                                   // nmfsc.m:178
                    std::swap(HTd, HnewT); std::swap(Hcur, Hcand);                                  // nmfsc.m:179
                    cur_obj = newobj;
                    have_dW = spec;
                } else {
                    TRY(fast_h_terms(Wd, Hcur));                                                    // W'*V, W'*V_hat        nmfsc.m:144-145
                    TRY(mu_plain(st, Hcur, Gb.as<float>(), Denb.as<float>(), (long)Kn));            // nmfsc.m:182
                    TRY(transpose_f32(st, Hcur, K, n, HTd));
                    TRY(col_reduce(st, HTd, n, n, Kv, 1, nrm2));                                    // nmfsc.m:185
                    if (comm.active()) TRY(comm.allreduce(nrm2, Kv, NMFX_F64, NMFX_REDUCE_SUM));    // row norms of H span the shards
                    TRY(scale_cols(st, HTd, n, Kv, nrm2, 1, 1));                                    // nmfsc.m:186
                    TRY(scale_cols(st, Wd, m, Kv, nrm2, 1, 0));                                     // nmfsc.m:187
                    TRY(transpose_f32(st, HTd, n, K, Hcur));
                    cur_obj = NAN;
                }
            }
            if (!fixW) {
                if (sW > 0) {
                    if (!have_dW) TRY(resid_w(Wd, Hcur, G2.as<float>(), cur_obj == cur_obj ? nullptr : &cur_obj));   // dW = V_hat*H' - V*H' (+ begobj)   nmfsc.m:193-200
                    else if (comm.active()) {
                        if (use64) TRY(comm.allreduce(g64w.p, (long)m * Kv, NMFX_F64, NMFX_REDUCE_SUM));
                        else TRY(comm.allreduce(G2.p, (long)mK, NMFX_F32, NMFX_REDUCE_SUM));
                    }
                    have_dW = false;
                    const bool spec_h = spec && it < p->maxiter;
                    const double begobj = cur_obj;
                    int tries = 0;
                    double newobj = 0;
                    for (;;) {
                        ++tries;
                        {
                            PScope ps(pf, SC_PROJ);
                            TRY(projfunc_cols(st, Wnew, m, Kv, L1a, 1.0, 1, nullptr, use64 ? nullptr : G2.as<float>(), -stepW, Wd, use64 ? g64w.as<double>() : nullptr));   // nmfsc.m:205-208
                        }
                        if (spec_h) TRY(resid_h(Wnew, Hcur, &newobj));                              // nmfsc.m:211-212 (+ dH at the candidate)
                        else TRY(fast_obj(Wnew, Hcur, &newobj));
                        if (newobj <= begobj) break;                                                // nmfsc.m:215
                        stepW /= 2;
                        if (stepW < 1e-200) { early = true; break; }                                // nmfsc.m:221-225
                    }
                    if (r->tries_W) r->tries_W[nW] = tries;
                    ++nW;
                    if (early) { ncost = it; break; }
                    stepW *= 1.2;                                                                   // nmfsc.m:228
                    std::swap(Wd, Wnew);                                                            // nmfsc.m:229
                    cur_obj = newobj;
                    have_dH = spec_h;
                } else {
                    TRY(fast_w_terms(Wd, Hcur, G1.as<float>(), G2.as<float>()));                    // V*H', V_hat*H'       nmfsc.m:194-195
                    PScope ps(pf, SC_SMALL);
                    TRY(mu_plain(st, Wd, G1.as<float>(), G2.as<float>(), (long)mK));                // nmfsc.m:232
                    cur_obj = NAN;
                }
            }
            if (cur_obj == cur_obj) r->cost[it] = cur_obj;                                          // same (W, H) as the accepted objective
            else if (lsH && it < p->maxiter) { TRY(resid_h(Wd, Hcur, &r->cost[it])); have_dH = true; }   // + the next iteration's dH
            else TRY(fast_obj(Wd, Hcur, &r->cost[it]));                                             // nmfsc.m:237-238
            if (p->tolerance >= 0 && it > 1 && r->cost[it] < r->cost[it - 1] && r->cost[it - 1] - r->cost[it] < p->tolerance) {   // nmfsc.m:241-244
                ncost = it + 1;
                break;
            }
        }
        r->cost_len = ncost;
        r->iters_run = ncost - 1;
        r->stepsize_H = stepH; r->stepsize_W = stepW;
        r->converged_early = early ? 1 : 0;
        if (r->tries_H) for (int i = nH; i < p->maxiter; ++i) r->tries_H[i] = 0;
        if (r->tries_W) for (int i = nW; i < p->maxiter; ++i) r->tries_W[i] = 0;
        if (dev) {
            NMFX_HIP(hipMemcpyAsync(dev->W, Wd, mKv * 4, hipMemcpyDeviceToDevice, st));
            if (padK) TRY(repack_rows(st, Hcur, K, dev->H, Kv, n));
            else NMFX_HIP(hipMemcpyAsync(dev->H, Hcur, Kn * 4, hipMemcpyDeviceToDevice, st));
            NMFX_HIP(hipStreamSynchronize(st));
            return NMFX_OK;
        }
        TRY(download(st, Wd, p->dtype, r->W, mKv, stage, STAGE_ELEMS));
        if (padK) {
            TRY(repack_rows(st, Hcur, K, hpk.as<float>(), Kv, n));
            TRY(download(st, hpk.as<float>(), p->dtype, r->H, Kvn, stage, STAGE_ELEMS));
        } else TRY(download(st, Hcur, p->dtype, r->H, Kn, stage, STAGE_ELEMS));
        return NMFX_OK;
    }
    TRY(recon_obj(Wd, HTd, &r->cost[0]));   // nmfsc.m:138-139
    int ncost = p->maxiter + 1, nH = 0, nW = 0;
    bool early = false;
    r->converged_early = 0;
    for (int it = 1; it <= p->maxiter && !early; ++it) {
        if (!fixH) {
            if (sH > 0) {
                TRY(xt_w(Vp, Vh.as<float>(), NMFX_PRO_DIFF, G1.as<float>()));   // dH' = (V_hat - V)' * W   nmfsc.m:144-148
                const double begobj = r->cost[it - 1];                                      // nmfsc.m:149
                int tries = 0;
                for (;;) {
                    ++tries;
                    TRY(projfunc_cols(st, HnewT, n, Kv, L1s, 1.0, 1, nullptr, G1.as<float>(), -stepH, HTd));   // nmfsc.m:154-157 (step formed in fp64 while loading)
                    double newobj;
                    TRY(recon_obj(Wd, HnewT, &newobj));                                         // nmfsc.m:160-161
                    if (newobj <= begobj) break;                                                // nmfsc.m:164
                    stepH /= 2;                                                                 // nmfsc.m:169
                    if (stepH < 1e-200) { early = true; break; }                                // nmfsc.m:170-174
                }
                if (r->tries_H) r->tries_H[nH] = tries;
                ++nH;
                if (early) { ncost = it; break; }
                stepH *= 1.2;                                                                   // nmfsc.m:178
                std::swap(HTd, HnewT);                                                          // nmfsc.m:179
            } else {
                TRY(xt_w(Vp, nullptr, NMFX_PRO_NONE, G1.as<float>()));               // (W'*V)'       nmfsc.m:144
                TRY(xt_w(Vh.as<float>(), nullptr, NMFX_PRO_NONE, G2.as<float>()));              // (W'*V_hat)'   nmfsc.m:145
                TRY(mu_plain(st, HTd, G1.as<float>(), G2.as<float>(), (long)Kn));               // nmfsc.m:182
                double *nrm2 = costd.as<double>() + 8;
                TRY(col_reduce(st, HTd, n, n, K, 1, nrm2));                                     // nmfsc.m:185
                TRY(scale_cols(st, HTd, n, K, nrm2, 1, 1));                                     // nmfsc.m:186
                TRY(scale_cols(st, Wd, m, K, nrm2, 1, 0));                                      // nmfsc.m:187
            }
        }
        if (!fixW) {
            double begobj;
            TRY(recon_obj(Wd, HTd, &begobj));                                                   // nmfsc.m:193,197
            if (sW > 0) {
                TRY(x_ht(Vp, Vh.as<float>(), NMFX_PRO_DIFF, G1.as<float>()));        // dW = (V_hat - V) * H'   nmfsc.m:194-200
                int tries = 0;
                for (;;) {
                    ++tries;
                    TRY(projfunc_cols(st, Wnew, m, Kv, L1a, 1.0, 1, nullptr, G1.as<float>(), -stepW, Wd));   // nmfsc.m:205-208
                    double newobj;
                    TRY(recon_obj(Wnew, HTd, &newobj));                                         // nmfsc.m:211-212
                    if (newobj <= begobj) break;                                                // nmfsc.m:215
                    stepW /= 2;
                    if (stepW < 1e-200) { early = true; break; }                                // nmfsc.m:221-225
                }
                if (r->tries_W) r->tries_W[nW] = tries;
                ++nW;
                if (early) { ncost = it; break; }
                stepW *= 1.2;                                                                   // nmfsc.m:228
                std::swap(Wd, Wnew);                                                            // nmfsc.m:229
            } else {
                TRY(x_ht(Vp, nullptr, NMFX_PRO_NONE, G1.as<float>()));               // nmfsc.m:194
                TRY(x_ht(Vh.as<float>(), nullptr, NMFX_PRO_NONE, G2.as<float>()));              // nmfsc.m:195
                TRY(mu_plain(st, Wd, G1.as<float>(), G2.as<float>(), (long)mK));                // nmfsc.m:232
            }
        }
        TRY(recon_obj(Wd, HTd, &r->cost[it]));                                                  // nmfsc.m:237-238
        if (p->tolerance >= 0 && it > 1 && r->cost[it] < r->cost[it - 1] && r->cost[it - 1] - r->cost[it] < p->tolerance) {   // nmfsc.m:241-244
            ncost = it + 1;
            break;
        }
    }
    r->cost_len = ncost;
    r->iters_run = ncost - 1;
    r->stepsize_H = stepH; r->stepsize_W = stepW;
    r->converged_early = early ? 1 : 0;
    if (r->tries_H) for (int i = nH; i < p->maxiter; ++i) r->tries_H[i] = 0;
    if (r->tries_W) for (int i = nW; i < p->maxiter; ++i) r->tries_W[i] = 0;
    TRY(transpose_f32(st, HTd, n, K, Hk.as<float>()));
    if (dev) {
        NMFX_HIP(hipMemcpyAsync(dev->W, Wd, mK * 4, hipMemcpyDeviceToDevice, st));
        NMFX_HIP(hipMemcpyAsync(dev->H, Hk.p, Kn * 4, hipMemcpyDeviceToDevice, st));
        NMFX_HIP(hipStreamSynchronize(st));
        return NMFX_OK;
    }
    TRY(download(st, Wd, p->dtype, r->W, mK, stage, STAGE_ELEMS));
    TRY(download(st, Hk.as<float>(), p->dtype, r->H, Kn, stage, STAGE_ELEMS));
    return NMFX_OK;
}

// cnmfsc.m:67-277 on the generic GEMM (materialised V_hat: its W branch updates V_hat incrementally, cnmfsc.m:262).
// The reference's quirks are mirrored, see oracle/nmf_oracle.py::cnmfsc.
nmfx_status run_cnmfsc(const nmfx_problem *p, nmfx_result *r) {
    TRY(validate_problem(p, r, true));
    const long m = p->m, n = p->n;
    const int K = p->K_total, T = p->T, KT = K * T;
    if (n < T) { set_error("cnmfsc: context_len exceeds the number of columns"); return NMFX_ERR_INVALID; }
    const size_t mn = (size_t)m * n, mK = (size_t)m * K, mKT = (size_t)m * KT, Kn = (size_t)K * n;
    double vmin = INFINITY, vmax = -INFINITY;   // cnmfsc.m:67-72
    if (p->dtype == NMFX_F64) { const double *v = static_cast<const double *>(p->V); for (size_t i = 0; i < mn; ++i) { if (v[i] < vmin) vmin = v[i]; if (v[i] > vmax) vmax = v[i]; } }
    else { const float *v = static_cast<const float *>(p->V); for (size_t i = 0; i < mn; ++i) { if (v[i] < vmin) vmin = v[i]; if (v[i] > vmax) vmax = v[i]; } }
    if (vmin < 0) { set_error("Negative values in data!"); return NMFX_ERR_NEGATIVE; }
    DeviceGuard dg_;
    TRY(check_device(p->device));
    hipStream_t st = nullptr;
    double sW = p->sc_W_sparsity, sH = p->sc_H_sparsity, L1a = 0, L1s = 0;
    if (sW > 0) { if (sW > 1) sW = 1; L1a = std::sqrt((double)m) - (std::sqrt((double)m) - 1) * sW; }   // cnmfsc.m:100-104
    if (sH > 0) { if (sH > 1) sH = 1; L1s = std::sqrt((double)n) - (std::sqrt((double)n) - 1) * sH; }   // cnmfsc.m:116-120
    const bool fixW = p->W_fixed && p->W_fixed[0], fixH = p->H_fixed && p->H_fixed[0];

    DevBuf V, Vh, W0b, Wb, Wnb, Hb, Hnb, HTb, HnT, G1, G2, stage, part, costd, scratch, rrs, g64, s64;
    // sparse-W gradients in fp64 where that is cheap (aux.hip::resid_xht64): m*n*K fp64 FMAs per slice
    const bool small64 = p->sc_W_sparsity > 0 && (double)p->m * (double)p->n * (double)p->K_total <= (double)(1 << 27);
    const int nch64 = small64 ? (int)std::min<long>(std::max<long>(1, 1024 / (((p->m + 255) / 256) * p->K_total)), (p->n + 63) / 64) : 1;
    if (small64) { TRY(g64.alloc(sizeof(double) * (size_t)p->m * p->K_total)); TRY(s64.alloc(sizeof(double) * (size_t)nch64 * p->m * p->K_total)); }
    DevBuf g64h;   // the same for the sparse-H gradient (aux.hip::resid_hgrad64): m*n*K*T fp64 FMAs
    const bool small64h = p->sc_H_sparsity > 0 && (double)p->m * (double)p->n * (double)p->K_total * (double)p->T <= (double)(1 << 27);
    if (small64h) TRY(g64h.alloc(sizeof(double) * (size_t)p->n * p->K_total));
    TRY(rrs.alloc(row_reduce_scratch_bytes(K)));
    TRY(HnT.alloc((size_t)p->K_total * p->n * 4));
    TRY(V.alloc(mn * 4)); TRY(Vh.alloc(mn * 4)); TRY(W0b.alloc(mKT * 4)); TRY(Wb.alloc(mKT * 4)); TRY(Wnb.alloc(mK * 4));
    TRY(Hb.alloc(Kn * 4)); TRY(Hnb.alloc(Kn * 4)); TRY(HTb.alloc(Kn * 4));
    const size_t gmax = std::max(Kn, mKT);
    TRY(G1.alloc(gmax * 4)); TRY(G2.alloc(gmax * 4)); TRY(stage.alloc(STAGE_ELEMS * 8));
    TRY(part.alloc(sizeof(double) * gemm_grid_blocks(m, n))); TRY(costd.alloc(64 + sizeof(double) * K));
    size_t sb = std::max(gemm_scratch_bytes(K, n, (long)T * m), gemm_scratch_bytes(m, K, n));
    TRY(scratch.alloc(sb));
    TRY(upload(st, p->V, p->dtype, V.as<float>(), mn, vmax, stage, STAGE_ELEMS));
    TRY(upload(st, p->W_init, p->dtype, W0b.as<float>(), mKT, 1.0, stage, STAGE_ELEMS));
    TRY(upload(st, p->H_init, p->dtype, Hb.as<float>(), Kn, 1.0, stage, STAGE_ELEMS));
    float *W0 = W0b.as<float>(), *W = Wb.as<float>(), *Wnew = Wnb.as<float>(), *H = Hb.as<float>(), *Hnew = Hnb.as<float>(), *HT = HTb.as<float>();
    NMFX_HIP(hipMemcpyAsync(W, W0, mKT * 4, hipMemcpyDeviceToDevice, st));                     // W = W0   cnmfsc.m:94
    if (sW > 0) TRY(projfunc_cols(st, W, m, KT, L1a, 1.0, 1, nullptr));                          // cnmfsc.m:105-109 (W only, not W0)
    auto project_rows = [&](float *Hx) -> nmfx_status {   // rows of H (K x n) through the transposed copy
        TRY(transpose_f32(st, Hx, K, n, HT));
        TRY(projfunc_cols(st, HT, n, K, L1s, 1.0, 1, nullptr));
        return transpose_f32(st, HT, n, K, Hx);
    };
    if (sH > 0) TRY(project_rows(H));                                                            // cnmfsc.m:121-123
    auto gemm = [&](GemmParams &g, double *obj) -> nmfx_status {
        g.splitk = 1;
        if (!obj) { g.epi = EPI_STORE; return gemm_auto(st, g, scratch.p, sb); }
        g.epi = EPI_COST; g.store_c = 1; g.cost_div = NMFX_DIV_EUCLIDEAN; g.Vref = V.as<float>(); g.ldv = m; g.cost_partials = part.as<double>();
        long blocks = 0;
        TRY(launch_gemm(st, g, &blocks));
        return read_obj(st, part.as<double>(), (int)blocks, costd.as<double>(), obj);
    };
    // V_hat = RFD(Wx (m x K x T), Hx) and 0.5*||V - V_hat||^2
    auto rfd = [&](const float *Wx, const float *Hx, double *obj) -> nmfx_status {
        GemmParams g; memset(&g, 0, sizeof(g));
        g.M = m; g.N = n; g.Kc = KT;
        g.A = OpView{Wx, nullptr, m, VIEW_RC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f};
        g.B = OpView{Hx, nullptr, (long)K, VIEW_HSTACK_KC, K, 0, 0, NMFX_PRO_NONE, 0.f, 0.f};
        g.C = Vh.as<float>(); g.ldc = m;
        return gemm(g, obj);
    };
    // out (K x n) = sum_t Wx_t' * lshift_t(X)
    // (X2 given: X is replaced by X2 - X element-wise while it is loaded -- the gradient in residual form, see run_nmfsc)
    auto hgrad = [&](const float *Wx, const float *X, float *out, const float *X2 = nullptr) -> nmfx_status {
        GemmParams g; memset(&g, 0, sizeof(g));
        g.M = K; g.N = n; g.Kc = (long)T * m;
        g.A = OpView{Wx, nullptr, m, VIEW_WSTACK_KC, (int)m, m * (long)K, 0, NMFX_PRO_NONE, 0.f, 0.f};
        g.B = OpView{X, X2, m, VIEW_XSHIFT_KC, (int)m, 0, (int)n, X2 ? NMFX_PRO_DIFF : NMFX_PRO_NONE, 0.f, 0.f};
        g.C = out; g.ldc = K;
        return gemm(g, nullptr);
    };
    // out (m x K) = X * rshift_t(H)'
    auto xht = [&](const float *X, const float *Hx, int t, float *out, const float *X2 = nullptr) -> nmfx_status {
        GemmParams g; memset(&g, 0, sizeof(g));
        g.M = m; g.N = K; g.Kc = n;
        g.A = OpView{X, X2, m, VIEW_RC, 0, 0, 0, X2 ? NMFX_PRO_DIFF : NMFX_PRO_NONE, 0.f, 0.f};
        g.B = OpView{Hx, nullptr, (long)K, VIEW_HSTACK_RC, K, 0, t * K, NMFX_PRO_NONE, 0.f, 0.f};
        g.C = out; g.ldc = m;
        return gemm(g, nullptr);
    };

    double stepH = 1.0;
    std::vector<double> stepW(T, 1.0);                                                           // cnmfsc.m:147-148
    TRY(rfd(W, H, &r->cost[0]));                                                                 // cnmfsc.m:152-153
    int ncost = p->maxiter + 1, nH = 0, nW = 0;
    bool early = false;
    double *nrm2 = costd.as<double>() + 8;
    for (int it = 1; it <= p->maxiter && !early; ++it) {
        if (!fixH) {
            if (sH > 0) {
                const double begobj = r->cost[it - 1];
                int tries = 0;
                TRY(transpose_f32(st, H, K, n, HT));                                                 // rows of H / dH contiguous: the projected vectors
                if (small64h) TRY(resid_hgrad64(st, V.as<float>(), Vh.as<float>(), m, n, W0, K, T, g64h.as<double>()));   // dH' in fp64 (small problems)
                else {
                    TRY(hgrad(W0, V.as<float>(), G2.as<float>(), Vh.as<float>()));              // dH = pos - neg = sum_t W0_t' * lshift_t(V_hat - V)   cnmfsc.m:160-168
                    TRY(transpose_f32(st, G2.as<float>(), K, n, G1.as<float>()));
                }
                for (;;) {
                    ++tries;
                    TRY(projfunc_cols(st, HnT.as<float>(), n, K, L1s, 1.0, 1, nullptr, small64h ? nullptr : G1.as<float>(), -stepH, HT, small64h ? g64h.as<double>() : nullptr));   // cnmfsc.m:174-177 (step formed in fp64 while loading)
                    TRY(transpose_f32(st, HnT.as<float>(), n, K, Hnew));
                    double newobj;
                    TRY(rfd(W0, Hnew, &newobj));                                                     // cnmfsc.m:180-181
                    if (newobj <= begobj) break;
                    stepH /= 2;
                    if (stepH < 1e-200) { early = true; break; }                                     // cnmfsc.m:190-194
                }
                if (r->tries_H) r->tries_H[nH] = tries;
                ++nH;
                if (early) { ncost = it; break; }
                stepH *= 1.2;
                std::swap(H, Hnew);
            } else {
                TRY(hgrad(W0, V.as<float>(), G1.as<float>()));                                   // cnmfsc.m:160-165
                TRY(hgrad(W0, Vh.as<float>(), G2.as<float>()));
                TRY(mu_plus_eps(st, H, G1.as<float>(), G2.as<float>(), (long)Kn));                   // H .* (neg ./ (pos + eps))   cnmfsc.m:202
                TRY(row_reduce(st, H, K, K, n, 1, nrm2, rrs.p));                                     // cnmfsc.m:205
                TRY(transpose_f32(st, H, K, n, HT));
                TRY(scale_cols(st, HT, n, K, nrm2, 1, 1));                                           // cnmfsc.m:206
                TRY(transpose_f32(st, HT, n, K, H));
                for (int t = 0; t < T; ++t) TRY(scale_cols(st, W0 + (size_t)t * mK, m, K, nrm2, 1, 0));   // cnmfsc.m:207-209
            }
        }
        if (!fixW) {
            double begobj;
            TRY(rfd(W0, H, &begobj));                                                            // cnmfsc.m:215
            for (int t = 0; t < T && !early; ++t) {
                float *W0t = W0 + (size_t)t * mK, *Wt = W + (size_t)t * mK;
                if (sW > 0) {
                    if (small64) TRY(resid_xht64(st, V.as<float>(), Vh.as<float>(), m, n, H, K, t, s64.as<double>(), nch64, g64.as<double>()));   // dW in fp64 (small problems)
                    else TRY(xht(V.as<float>(), H, t, G2.as<float>(), Vh.as<float>()));          // dW = pos - neg = (V_hat - V) * Hs'   cnmfsc.m:221-224
                    int tries = 0;
                    double newobj = 0;
                    for (;;) {
                        ++tries;
                        TRY(projfunc_cols(st, Wnew, m, K, L1a, 1.0, 1, nullptr, small64 ? nullptr : G2.as<float>(), -stepW[t], W0t, small64 ? g64.as<double>() : nullptr));   // cnmfsc.m:229-233 (step formed in fp64 while loading)
                        GemmParams g; memset(&g, 0, sizeof(g));                                      // RFD(Wnew, H) with a 2-D Wnew: plain Wnew*H  (cnmfsc.m:235)
                        g.M = m; g.N = n; g.Kc = K;
                        g.A = OpView{Wnew, nullptr, m, VIEW_RC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f};
                        g.B = OpView{H, nullptr, (long)K, VIEW_KC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f};
                        g.C = Vh.as<float>(); g.ldc = m;
                        TRY(gemm(g, &newobj));
                        if (newobj <= begobj) break;
                        stepW[t] /= 2;
                        if (stepW[t] < 1e-200) { early = true; break; }                              // cnmfsc.m:245-249
                    }
                    if (r->tries_W) r->tries_W[nW] = tries;
                    ++nW;
                    if (early) { ncost = it; break; }
                    stepW[t] *= 1.2;
                    NMFX_HIP(hipMemcpyAsync(Wt, Wnew, mK * 4, hipMemcpyDeviceToDevice, st));         // W(:,:,t) = Wnew
                    begobj = newobj;                                                                 // next t: 0.5*||V - V_hat||^2 of the V_hat left here
                } else {
                    TRY(xht(V.as<float>(), H, t, G1.as<float>()));                               // neg = V * Hs'
                    TRY(xht(Vh.as<float>(), H, t, G2.as<float>()));                              // pos = V_hat * Hs'
                    NMFX_HIP(hipMemcpyAsync(Wt, W0t, mK * 4, hipMemcpyDeviceToDevice, st));
                    TRY(mu_plain(st, Wt, G1.as<float>(), G2.as<float>(), (long)mK));                 // W_t = W0_t .* (neg ./ max(pos, eps))   cnmfsc.m:261
                    TRY(axpy_f32(st, (long)mK, -1.0f, W0t, Wt, Wnew));                               // dW = W_t - W0_t
                    GemmParams g; memset(&g, 0, sizeof(g));                                          // V_hat = max(V_hat + dW * rshift_t(H), 0)   cnmfsc.m:262
                    g.M = m; g.N = n; g.Kc = K;
                    g.A = OpView{Wnew, nullptr, m, VIEW_RC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f};
                    // rshift_t(H) as a view whose base lies t columns BEFORE H: every element it may touch (column r - t >= 0) is inside H, but the
                    // base itself is not -- chunks outside the view must load from inside the allocation (safe), or the launch faults when H
                    // happens to start a mapping (found by scripts/fuzz_campaign_sc.py)
                    g.B = OpView{H - (long)K * t, nullptr, (long)K, VIEW_HSTACK_KC, K, 0, -t, NMFX_PRO_NONE, 0.f, 0.f, 0, (long)K * t};
                    g.C = Vh.as<float>(); g.ldc = m; g.accumulate = 1; g.clamp0 = 1; g.epi = EPI_STORE; g.splitk = 1;
                    TRY(launch_gemm(st, g));
                }
            }
            if (early) break;
        }
        NMFX_HIP(hipMemcpyAsync(W0, W, mKT * 4, hipMemcpyDeviceToDevice, st));                   // W0 = W   cnmfsc.m:266
        TRY(rfd(W0, H, &r->cost[it]));                                                           // cnmfsc.m:269-270
        if (p->tolerance >= 0 && it > 1 && r->cost[it] < r->cost[it - 1] && r->cost[it - 1] - r->cost[it] < p->tolerance) {   // cnmfsc.m:273-276
            ncost = it + 1;
            break;
        }
    }
    r->cost_len = ncost;
    r->iters_run = ncost - 1;
    r->stepsize_H = stepH; r->stepsize_W = stepW[0];
    r->converged_early = early ? 1 : 0;
    if (r->tries_H) for (int i = nH; i < p->maxiter; ++i) r->tries_H[i] = 0;
    if (r->tries_W) for (int i = nW; i < p->maxiter * T; ++i) r->tries_W[i] = 0;
    TRY(download(st, W, p->dtype, r->W, mKT, stage, STAGE_ELEMS));
    TRY(download(st, H, p->dtype, r->H, Kn, stage, STAGE_ELEMS));
    return NMFX_OK;
}

}  // namespace

extern "C" {

nmfx_status nmfx_nmf(const nmfx_problem *p, nmfx_result *r) { return (p && p->n_gpus > 1) ? run_mu_multi(p, r, 0) : run_mu(p, r, 0); }
nmfx_status nmfx_cnmf(const nmfx_problem *p, nmfx_result *r) { return (p && p->n_gpus > 1) ? run_mu_multi(p, r, 1) : run_mu(p, r, 1); }
nmfx_status nmfx_lnmf(const nmfx_problem *p, nmfx_result *r) { return (p && p->n_gpus > 1) ? run_mu_multi(p, r, 2) : run_mu(p, r, 2); }
nmfx_status nmfx_constrainednmf(const nmfx_problem *p, const int64_t *segments, int64_t nz, const void *Z_init, nmfx_result *r, void *Z_out) {
    return run_mu(p, r, 3, segments, nz, Z_init, Z_out);
}
nmfx_status nmfx_nmfsc(const nmfx_problem *p, nmfx_result *r) { return run_nmfsc(p, r); }
nmfx_status nmfx_nmfsc_dev(const nmfx_problem *p, const float *V_dev, float *W_dev, float *H_dev, int64_t n_total, void *stream,
                           nmfx_allreduce_fn allreduce, void *allreduce_ctx, nmfx_result *r) {
    if (!p || !r || !V_dev || !W_dev || !H_dev || !r->cost) { set_error("nmfx_nmfsc_dev: null argument"); return NMFX_ERR_INVALID; }
    if (p->m <= 0 || p->n <= 0 || p->K_total <= 0 || p->maxiter <= 0 || n_total < p->n) { set_error("nmfx_nmfsc_dev: bad sizes"); return NMFX_ERR_INVALID; }
    DeviceGuard dg_;
    ScDev d{};
    d.V = V_dev; d.W = W_dev; d.H = H_dev; d.n_total = n_total; d.st = static_cast<hipStream_t>(stream);
    d.comm.fn = allreduce; d.comm.ctx = allreduce_ctx; d.comm.st = d.st;
    return run_nmfsc(p, r, &d);
}
nmfx_status nmfx_cnmfsc(const nmfx_problem *p, nmfx_result *r) { return run_cnmfsc(p, r); }

nmfx_status nmfx_reconstruct(int64_t m, int64_t n, int32_t K, int32_t T, int32_t dtype, const void *W, const void *H, void *V_hat,
                             int32_t device) {
    if (m <= 0 || n <= 0 || K <= 0 || T <= 0 || !W || !H || !V_hat) { set_error("nmfx_reconstruct: bad arguments"); return NMFX_ERR_INVALID; }
    DeviceGuard dg_;
    TRY(check_device(device));
    const size_t mn = (size_t)m * n, mKT = (size_t)m * K * T, Kn = (size_t)K * n;
    DevBuf Wd, Hd, Vd, stage;
    TRY(Wd.alloc(mKT * 4)); TRY(Hd.alloc(Kn * 4)); TRY(Vd.alloc(mn * 4)); TRY(stage.alloc(STAGE_ELEMS * 8));
    hipStream_t st = nullptr;
    TRY(upload(st, W, dtype, Wd.as<float>(), mKT, 1.0, stage, STAGE_ELEMS));
    TRY(upload(st, H, dtype, Hd.as<float>(), Kn, 1.0, stage, STAGE_ELEMS));
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.M = m; g.N = n; g.Kc = (long)K * T;
    g.A = OpView{Wd.as<float>(), nullptr, (long)m, VIEW_RC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f};
    if (T == 1) g.B = OpView{Hd.as<float>(), nullptr, (long)K, VIEW_KC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f};
    else g.B = OpView{Hd.as<float>(), nullptr, (long)K, VIEW_HSTACK_KC, K, 0, 0, NMFX_PRO_NONE, 0.f, 0.f};
    g.C = Vd.as<float>(); g.ldc = m; g.epi = EPI_STORE; g.splitk = 1;
    TRY(launch_gemm(st, g));
    return download(st, Vd.as<float>(), dtype, V_hat, mn, stage, STAGE_ELEMS);
}

// [W_sorted, H_sorted] = SortDictionary(W, H): basis columns by increasing centre of mass (SortDictionary.m:33-47), computed in the
// buffers' own dtype; H / H_sorted may be NULL (nargin < 2).  order_out[K] receives the 0-based permutation (`sorted` - 1).
nmfx_status nmfx_sort_dictionary(int64_t m, int32_t K, int64_t n, int32_t dtype, const void *W, const void *H, void *W_sorted, void *H_sorted,
                                 int32_t *order_out, int32_t device) {
    if (m <= 0 || K <= 0 || !W || !W_sorted || (H && (!H_sorted || n <= 0))) { set_error("nmfx_sort_dictionary: bad arguments"); return NMFX_ERR_INVALID; }
    if (dtype != NMFX_F32 && dtype != NMFX_F64) { set_error("dtype must be NMFX_F32 or NMFX_F64"); return NMFX_ERR_INVALID; }
    DeviceGuard dg_;
    TRY(check_device(device));
    const size_t es = dsize(dtype), wb = (size_t)m * K * es, hb = H ? (size_t)K * n * es : 0;
    DevBuf Wd, Ws, Hd, Hs, cog, ord;
    TRY(Wd.alloc(wb)); TRY(Ws.alloc(wb)); TRY(cog.alloc(sizeof(int) * K)); TRY(ord.alloc(sizeof(int) * K));
    hipStream_t st = nullptr;
    NMFX_HIP(hipMemcpyAsync(Wd.p, W, wb, hipMemcpyHostToDevice, st));
    TRY(center_of_gravity(st, Wd.p, dtype == NMFX_F64, m, K, cog.as<int>()));
    std::vector<int> cg(K), order(K);
    NMFX_HIP(hipMemcpyAsync(cg.data(), cog.p, sizeof(int) * K, hipMemcpyDeviceToHost, st));
    NMFX_HIP(hipStreamSynchronize(st));
    for (int k = 0; k < K; ++k) order[k] = k;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cg[a] < cg[b]; });   // MATLAB sort is stable (SortDictionary.m:43)
    NMFX_HIP(hipMemcpyAsync(ord.p, order.data(), sizeof(int) * K, hipMemcpyHostToDevice, st));
    TRY(permute(st, Wd.p, Ws.p, dtype == NMFX_F64, m, K, ord.as<int>(), 0));
    NMFX_HIP(hipMemcpyAsync(W_sorted, Ws.p, wb, hipMemcpyDeviceToHost, st));
    if (H) {
        TRY(Hd.alloc(hb)); TRY(Hs.alloc(hb));
        NMFX_HIP(hipMemcpyAsync(Hd.p, H, hb, hipMemcpyHostToDevice, st));
        TRY(permute(st, Hd.p, Hs.p, dtype == NMFX_F64, K, n, ord.as<int>(), 1));
        NMFX_HIP(hipMemcpyAsync(H_sorted, Hs.p, hb, hipMemcpyDeviceToHost, st));
    }
    NMFX_HIP(hipStreamSynchronize(st));
    if (order_out) for (int k = 0; k < K; ++k) order_out[k] = order[k];
    return NMFX_OK;
}

nmfx_status nmfx_projfunc_dev(void *stream, float *X_dev, int64_t N, int32_t count, double k1, double k2, int32_t nn, const float *src_dev,
                              const float *dir_dev, double mu, int32_t *usediters_dev) {
    if (N <= 0 || count <= 0 || !X_dev) { set_error("nmfx_projfunc_dev: bad arguments"); return NMFX_ERR_INVALID; }
    return projfunc_cols(static_cast<hipStream_t>(stream), X_dev, N, count, k1, k2, nn, usediters_dev, dir_dev, mu, src_dev);
}

nmfx_status nmfx_nmfsc_profile(int32_t enable) {
    g_sc_prof.enable(enable != 0);
    if (!enable) g_sc_prof.release();
    return NMFX_OK;
}
int32_t nmfx_nmfsc_profile_ntags(void) { return SC_COUNT; }
const char *nmfx_nmfsc_profile_tag_name(int32_t tag) { return (tag >= 0 && tag < SC_COUNT) ? kScTagNames[tag] : ""; }
nmfx_status nmfx_nmfsc_profile_read(double *ms_per_tag, int32_t *count_per_tag) { return g_sc_prof.read(SC_COUNT, ms_per_tag, count_per_tag); }

nmfx_status nmfx_projfunc(int64_t N, int32_t count, int32_t dtype, const void *s, double k1, double k2, int32_t nn, void *v,
                          int32_t *usediters, int32_t device) {
    if (N <= 0 || count <= 0 || !s || !v) { set_error("nmfx_projfunc: bad arguments"); return NMFX_ERR_INVALID; }
    DeviceGuard dg_;
    TRY(check_device(device));
    if (dtype != NMFX_F32 && dtype != NMFX_F64) { set_error("nmfx_projfunc: dtype must be NMFX_F32 or NMFX_F64"); return NMFX_ERR_INVALID; }
    const size_t tot = (size_t)N * count;
    DevBuf X, it;
    TRY(X.alloc(tot * dsize(dtype))); TRY(it.alloc(sizeof(int) * count));
    hipStream_t st = nullptr;
    // the vectors stay in the caller's precision: float64 input is projected in float64 end to end (projfunc.m computes in double)
    NMFX_HIP(hipMemcpyAsync(X.p, s, tot * dsize(dtype), hipMemcpyHostToDevice, st));
    if (dtype == NMFX_F64) TRY(projfunc_cols_f64(st, X.as<double>(), N, count, k1, k2, nn, it.as<int>()));
    else TRY(projfunc_cols(st, X.as<float>(), N, count, k1, k2, nn, it.as<int>()));
    if (usediters) NMFX_HIP(hipMemcpyAsync(usediters, it.p, sizeof(int) * count, hipMemcpyDeviceToHost, st));
    NMFX_HIP(hipMemcpyAsync(v, X.p, tot * dsize(dtype), hipMemcpyDeviceToHost, st));
    NMFX_HIP(hipStreamSynchronize(st));
    return NMFX_OK;
}

}  // extern "C"

// The device-resident engine behind the C ABI (include/nmfx.h): error plumbing and the phases of one multiplicative-update iteration
// (nmf.m:143-225, cnmf.m:175-258, lnmf.m:66-88, constrainednmf.m:183-258).  Kernels live in gemm.hip / fused*.hip / aux.hip.
#include "api_common.h"

namespace nmfx {

inline bool dual2_store() { return true; }   // IS / alpha-beta above K = 192: the W step as 4 + 2 instead of 4 + 4 m*n*K (the second map's values kept in the m x n scratch)
static thread_local char g_err[1024] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

nmfx_status check_device(int device) {
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

}  // namespace nmfx

using namespace nmfx;

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
    e->W64 = c.take<double>(mKT);
    e->H64 = e->algo == 3 ? nullptr : c.take<double>(Kn);
    e->P64 = nullptr;
    Layout L;
    if (e->fused) {
        // V_hat, Gn/Gp of the generic path are not needed: rewind and carve the fused buffers instead
        Carver f(ws);
        e->Vhat = nullptr;
        e->WT = f.take<float>(mKT);
        if (e->dual2 && !e->dualz && dual2_store()) e->Vhat = f.take<float>(mn);   // the second element map's values of the W step (1./S, S.^(a+b-1)): written by the first pass, contracted by the second
        // row-chunked W steps use more splits on fewer rows: rows*split per launch never exceeds max(nsplit_w, 2) * m / 2
        e->slabs = f.take<float>(std::max((size_t)std::max(e->nsplit_w, 2) * mKT, (size_t)e->isplit_h * Kn));
        e->slabs2 = e->dual ? f.take<float>(std::max((size_t)std::max(e->nsplit_w, 2) * mKT, (size_t)e->isplit_h * Kn)) : nullptr;
        e->Valpha = (e->dual && e->div == NMFX_DIV_AB && e->alpha != 1.0) ? f.take<float>((size_t)e->m * e->n) : nullptr;
        e->VT = e->use_vt ? f.take<float>((size_t)e->m * e->n) : nullptr;
        e->VTa = (e->use_vt && e->dual2 && e->div == NMFX_DIV_AB && e->alpha != 1.0) ? f.take<float>((size_t)e->m * e->n) : nullptr;   // (by the rule that gives Valpha, not by its pointer: the size query carves from a null base)
        e->Gn = f.take<float>(Kn);
        const bool euc = e->div == NMFX_DIV_EUCLIDEAN;
        e->Gp = (euc || e->dual) ? f.take<float>(Kn) : nullptr;
        e->P64 = euc ? f.take<double>(mKT) : nullptr;
        e->W64 = f.take<double>(mKT);
        e->H64 = e->algo == 3 ? nullptr : f.take<double>(Kn);
        e->GW = euc ? f.take<float>((size_t)e->K * e->K) : nullptr;
        size_t g1 = gemm_scratch_bytes(e->K, e->K, e->n), g2 = gemm_scratch_bytes(e->K, e->K, e->m), g3 = gemm_scratch_bytes(e->K, e->n, e->m);
        e->gemm_scratch_bytes = euc ? std::max(std::max(std::max(g1, g2), g3), sizeof(float) * (size_t)e->K * e->K * 128)   /* gram_fused: up to 128 column slabs of K x K */ : 0;
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
        e->sumVV = f.take<double>(2);
        e->dndp = f.take<double>(2 * (size_t)e->K);
        e->exact_flag = f.take<int>(16);
        L.total = f.off;
        L.packed_count = e->dual ? 2 * mKT : (euc ? mKT + (size_t)e->K * e->K : mKT + (size_t)e->KT);
        return L;
    }
    if (e->gram) {
        e->P64 = c.take<double>(mKT);
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
    if (e->p1gram) e->CC = c.take<float>((size_t)e->K * e->K);
    if (e->gram_cost) {   // (cnmf on the fused passes)
        e->sumVV = c.take<double>(2);
        e->dndp = c.take<double>(2 * (size_t)e->KT);
        e->exact_flag = c.take<int>(16);
        e->colV = c.take<double>(e->n);
    }
    if (e->qgemm) e->Qbuf = c.take<float>((size_t)e->KT * (e->n + e->hR));
    if (e->use_vtq) { e->VT = c.take<float>((size_t)e->m * e->n); e->WTf = c.take<float>(mKT); }
    if (e->fusedT_kl) {
        e->Hpad = c.take<float>((size_t)e->K * (e->n + e->hR + e->T - 1));
        e->slabsT = e->nsplit_T > 1 ? c.take<float>((size_t)e->nsplit_T * mKT) : nullptr;
        const int need = (int)((e->m + 127) / 128) * e->nsplit_T;
        if (need > e->n_cost_partials) { e->n_cost_partials = need; e->cost_partials = c.take<double>(need); }
        e->sumV_g = c.take<double>(1);
        e->colV_g = c.take<double>(e->n);
    }
    if (e->fusedT_dual) {
        e->Hpad = c.take<float>((size_t)e->K * (e->n + e->hR + e->T - 1));
        e->slabsT = e->nsplit_T > 1 ? c.take<float>((size_t)e->nsplit_T * mKT) : nullptr;
        const int need = (int)((e->m + 127) / 128) * e->nsplit_T;
        if (need > e->n_cost_partials) { e->n_cost_partials = need; e->cost_partials = c.take<double>(need); }
        e->Vhat2 = c.take<float>(mn);
        e->Valpha = (e->div == NMFX_DIV_AB && e->alpha != 1.0) ? c.take<float>(mn) : nullptr;
        e->sumVab = c.take<double>(1);
        e->colV = c.take<double>(e->n);
    }
    if (e->klw) {
        e->slabsT = e->nsplit_T > 1 ? c.take<float>((size_t)e->nsplit_T * e->m * 256) : nullptr;
        const int need = (int)((e->m + 127) / 128) * e->nsplit_T;
        if (need > e->n_cost_partials) { e->n_cost_partials = need; e->cost_partials = c.take<double>(need); }
        e->sumV = c.take<double>(1);
        e->colV = c.take<double>(e->n);
        if (e->klw_vt) {
            e->VT = c.take<float>((size_t)e->m * e->n);
            e->WT = c.take<float>(mKT);
            e->slabsH = e->klw_hsplit > 1 ? c.take<float>((size_t)e->klw_hsplit * Kn) : nullptr;
            const int needh = (int)((e->n + 127) / 128) * e->klw_hsplit;
            if (needh > e->n_cost_partials) { e->n_cost_partials = needh; e->cost_partials = c.take<double>(needh); }
        }
    }
    if (e->dualw) {
        e->slabsT = e->nsplit_T > 1 ? c.take<float>((size_t)e->nsplit_T * e->m * 256) : nullptr;
        const int need = (int)((e->m + 127) / 128) * e->nsplit_T;
        if (need > e->n_cost_partials) { e->n_cost_partials = need; e->cost_partials = c.take<double>(need); }
        e->Vhat2 = c.take<float>(mn);
        e->Valpha = (e->div == NMFX_DIV_AB && e->alpha != 1.0) ? c.take<float>(mn) : nullptr;
        e->sumVab = c.take<double>(1);
        e->colV = c.take<double>(e->n);
    }
    if (e->eucw) {
        e->slabsT = e->nsplit_T > 1 ? c.take<float>((size_t)e->nsplit_T * e->m * 256) : nullptr;
        e->VT = c.take<float>((size_t)e->m * e->n);
        e->WT = c.take<float>(mKT);
        e->slabsH = e->klw_hsplit > 1 ? c.take<float>((size_t)e->klw_hsplit * Kn) : nullptr;
        const int need = std::max((int)((e->m + 127) / 128) * e->nsplit_T, (int)((e->n + 127) / 128) * e->klw_hsplit);
        if (need > e->n_cost_partials) { e->n_cost_partials = need; e->cost_partials = c.take<double>(need); }
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
    // IS and alpha-beta (alpha ~= 0: the dual form has other equations) need two element maps per pass: with two accumulator sets in ONE pass up to K = 192
    // (registers), as two single-map passes above it (dual2, round 4)
    e->dual = (e->div == NMFX_DIV_IS || (e->div == NMFX_DIV_AB && e->alpha != 0)) && e->K <= 256;
    e->dual2 = e->dual && e->K > 192;
    // alpha == 0: the dual update equations (nmf.m:124-128).  Two passes per half-iteration at any K: numerators through S (functor 17), denominators without it
    e->dualz = e->div == NMFX_DIV_AB && e->alpha == 0 && e->K <= 256 && e->algo != 3;
    if (e->dualz) e->dual = e->dual2 = true;
    const bool eligible = (e->algo == 0 || e->algo == 2 || e->algo == 3) && e->T == 1 && (e->div == NMFX_DIV_KL || e->div == NMFX_DIV_EUCLIDEAN || e->dual) && fused_supported(e->K) &&
                          e->hL == 0 && e->hR == 0 && ((e->m >= 64 && e->n >= 64) || d->path == 2);   // ragged m / n: masked-edge kernels
    if (d->path == 2 && !eligible && e->algo != 1) {   // cnmf: see the fused shift-sum passes below
        set_error("nmfx_engine: fused path requested but the problem is not eligible (nmf / lnmf / constrainednmf rules, kl or euclidean, K a multiple of 32 up to 256)");
        return NMFX_ERR_UNSUPPORTED;
    }
    e->fused = eligible && d->path != 1;
    if (!e->fused) e->dual = e->dual2 = e->dualz = false;
    static const bool exact_cost_env = getenv("NMFX_EXACT_COST") != nullptr;   // dev switch (A/B runs): always the explicit residual inside the W-step pass
    e->gram_cost = e->fused && e->div == NMFX_DIV_EUCLIDEAN && !e->dual && !exact_cost_env;
    static const bool no_vt = getenv("NMFX_NO_VT") != nullptr;   // dev switch (A/B runs): H-step numerator on the pipelined GEMM, no transposed copy of V
    // the transposed copy of V (euclidean paths) is asked for unless the caller says no (nmfx_engine_desc.flags bit 0): the choice of
    // kernel -- and with it the summation order -- must not depend on how much memory happens to be free (run-to-run and rank-to-rank reproducibility).
    // A caller that cannot allocate the workspace with the copy retries with the flag set (the blocking API does).
    const bool room_vt = (d->flags & 1) == 0;
    e->use_vt = e->fused && (e->div == NMFX_DIV_EUCLIDEAN || (e->dual2 && !e->dualz && dual2_store())) && !no_vt && room_vt;
    // euclidean problems the register-stationary kernels do not take (cnmf; nmf / constrainednmf with K > 256 or tiny shapes) still never
    // materialise V_hat: denominators from Gram products, the cost from a store-less residual pass
    e->gram = !e->fused && (e->algo == 0 || e->algo == 1 || e->algo == 3) && (e->div == NMFX_DIV_EUCLIDEAN || e->div == NMFX_DIV_EUCLIDEAN_NOCOST) && d->path != 1;
    e->p1gram = !e->fused && !e->gram && (e->algo == 0 || e->algo == 3) && e->T == 1 && (e->div == NMFX_DIV_EUCLIDEAN || e->div == NMFX_DIV_EUCLIDEAN_NOCOST);
    e->fusedT = e->gram && e->T > 1 && fused_supported_T(e->K, e->T) && e->m >= 64 && e->n >= 64 && (e->hL == 0 || e->hL >= e->T - 1);
    // (column shards: the T-1 columns left of the shard are its halo -- or zeros on the first one --, and R = V./V_hat is also formed on the T-1 right-halo
    // columns, whose terms the shift-sum of the H step needs: cnmf.m:219)
    e->fusedT_kl = !e->fused && e->algo == 1 && e->div == NMFX_DIV_KL && e->T > 1 && fused_supported_T(e->K, e->T) && e->m >= 64 && e->n >= 64 &&
                   (e->hL == 0 || e->hL >= e->T - 1) && d->path != 1;
    // IS / alpha-beta (alpha != 0) cnmf, unsharded, the common (K, T) pairs: S pass with both element maps stored (functors 11 / 13 in the cost-only form),
    // numerator passes on either buffer, the H-step products as two well-shaped GEMMs on them (cnmf.m:179-194,227-231 without V_hat)
    e->fusedT_dual = !e->fused && e->algo == 1 && (e->div == NMFX_DIV_IS || (e->div == NMFX_DIV_AB && e->alpha != 0)) && e->T > 1 && fused_supported_T_dual(e->K, e->T) &&
                     e->m >= 64 && e->n >= 64 && e->K % 4 == 0 && e->m % 4 == 0 && (e->hL == 0 || e->hL >= e->T - 1) && d->path != 1;   // (column shards as for KL: halos)
    if (d->path == 2 && e->algo == 1 && !e->fusedT && !e->fusedT_kl && !e->fusedT_dual) {
        set_error("nmfx_engine: fused cnmf kernels requested but the problem is not eligible (euclidean or kl, T > 1, an instantiated (K, T) pair)");
        return NMFX_ERR_UNSUPPORTED;
    }
    if (e->fusedT || e->fusedT_kl || e->fusedT_dual) e->nsplit_T = fused_split((e->m + 127) / 128, e->n, e->KT, &e->cps_T);
    e->klw = !e->fused && e->algo != 1 && e->T == 1 && e->div == NMFX_DIV_KL && e->K > 256 && e->K % 32 == 0 && e->K <= 8 * 256 && e->hL == 0 && e->hR == 0 &&
             e->m >= 64 && e->n >= 64 && d->path != 1;
    e->eucw = e->gram && (e->algo == 0 || e->algo == 3) && e->T == 1 && e->div == NMFX_DIV_EUCLIDEAN && e->K > 256 && e->K % 32 == 0 && e->K <= 8 * 256 && e->hL == 0 && e->hR == 0 &&
              e->m >= 64 && e->n >= 64 && !no_vt && room_vt;
    if (e->eucw && !exact_cost_env) e->gram_cost = true;
    e->dualw = !e->fused && e->algo == 0 && e->T == 1 && (e->div == NMFX_DIV_IS || (e->div == NMFX_DIV_AB && e->alpha != 0)) && e->K > 256 && e->K % 32 == 0 && e->K <= 8 * 256 &&
               e->hL == 0 && e->hR == 0 && e->nvalid == e->n && e->m >= 64 && e->n >= 64 && d->path != 1;
    if (e->klw || e->eucw || e->dualw) {   // column blocks: as few as fit 256, as even as multiples of 32 allow (320 = 160 + 160, 288 = 160 + 128, 512 = 256 + 256)
        const int units = e->K / 32;
        e->klw_nb = (e->K + 255) / 256;
        for (int b = 0, k0 = 0; b < e->klw_nb; ++b) {
            const int u = units / e->klw_nb + (b < units % e->klw_nb ? 1 : 0);
            e->klw_k0[b] = k0; e->klw_kb[b] = 32 * u;
            k0 += 32 * u;
        }
        e->nsplit_T = fused_split((e->m + 127) / 128, e->n, 256, &e->cps_T);
        e->klw_vt = e->klw && !no_vt && room_vt;
        e->klw_hsplit = fused_split((e->n + 127) / 128, e->m, 256, &e->klw_hcps);
    }
    e->lagram = e->fusedT && e->hL == 0 && e->hR == 0 && e->nvalid == e->n && e->n >= 2L * e->T;
    // cnmf on the fused passes, unsharded: the same Gram-form cost (its explicit residual pass is a third of the iteration)
    if (e->fusedT && e->div == NMFX_DIV_EUCLIDEAN && e->hL == 0 && e->hR == 0 && e->nvalid == e->n && !exact_cost_env) e->gram_cost = true;
    e->qgemm = !e->fused && e->algo == 1 && e->T > 1 && e->K % 4 == 0 && e->m % 4 == 0 && d->path != 1;
    {   // euclidean cnmf on the fused passes, unsharded: the Q product of the H step on a transposed copy of V (see nmfx_engine_hstep)
        static const bool no_vt = getenv("NMFX_NO_VT") != nullptr;
        e->vtq_block = 128;   // C4 (K*T = 512): four 128-wide blocks, two workgroups per CU, 0.526 ms; two 256-wide blocks 0.549; the two-operand GEMM 0.585
        e->use_vtq = e->fusedT && e->qgemm && e->hL == 0 && e->hR == 0 && !no_vt && room_vt && e->KT % e->vtq_block == 0 && fused_supported(e->vtq_block);
    }
    e->nsplit_w = e->isplit_h = 1;
    if (e->fused) {
        e->nsplit_w = fused_split((e->m + 127) / 128, e->n, e->K, &e->cps_w);
        e->isplit_h = fused_split((e->n + 127) / 128, e->m, e->K, &e->cps_h);
    }
    return NMFX_OK;
}

// X*X' (K x K) for X = K x len with contiguous columns, on the W-step form of the stationary kernel: D = X plays V (K "rows"), the columns of X
// are streamed through LDS by DMA exactly like the columns of H in the W step -- out(k, r) = sum_c X(r, c) X(k, c)
nmfx_status gram_fused(nmfx_engine *e, const float *X, long len, float *G) {
    const int K = e->K;
    const long blocks = (K + 127) / 128, tiles = (len + 63) / 64;
    long s = 1;
    const long gs_cap = 256;
    while (blocks * s * 2 <= gs_cap && s * 4 <= tiles) s *= 2;          // one workgroup per CU, at least two tiles each
    const long per = (tiles + s - 1) / s;
    const int split = (int)((tiles + per - 1) / per);
    if (split > 1 && sizeof(float) * (size_t)split * K * K > e->gemm_scratch_bytes) { set_error("gram_fused: scratch too small"); return NMFX_ERR_INVALID; }
    FusedParams g;
    memset(&g, 0, sizeof(g));
    g.Y = X; g.D = X; g.ldd = K; g.R = K; g.Cn = len; g.K = K; g.c_per_split = per * 64;
    g.out = split == 1 ? G : e->gemm_scratch; g.slab_stride = (long)K * K; g.os_r = K; g.os_k = 1;
    TRY(launch_fused(e->st, g, split, true, 0, true, 0));
    if (split > 1) TRY(reduce_slabs(e->st, e->gemm_scratch, split, g.slab_stride, g.slab_stride, G, 0));
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
    if ((e->fused && e->dual) || e->fusedT_dual || e->dualw) {
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
    // fused KL: the partials hold the complete divergence of the workgroups' elements (NMFX_KL_MODE 1 / 2).  Mode 0: sum V.*log(V./V_hat) only, and
    // sum(V_hat) - sum(V) = sum_k colsum(W)_k * rowsum(H_local)_k - sum(V_local) is added here
    const bool tail = e->fused && kl_closed_form && e->tail_with_cost;
    return finish_cost(e->st, e->cost_partials, nparts, scale, useW ? e->l1W : nullptr, e->KT, e->lamW, useH ? e->l1H : nullptr, e->K,
                       e->lamH, e->cost, (kl_closed_form && !KL_CONSISTENT_COST) ? e->Gpvec : nullptr, e->rowsum, e->K, e->sumV, nullptr, 0.0, 0.0,
                       e->cost_dst2, tail ? e->rowsum : nullptr, tail ? e->packed + (size_t)e->m * e->KT : nullptr, e->K);
}


}  // namespace

namespace nmfx {
// grid.y of a fused pass over `blocks` 128-row blocks: enough workgroups for 256 CUs while every slice keeps whole 64-column tiles
int fused_split(long blocks, long extent, int K, long *c_per_split) {
    const long target = K <= 128 ? 512 : 256;   // K <= 128 kernels fit two workgroups per CU
    const long tiles = (extent + 63) / 64;
    long s = 1;
    while (blocks * s < target && s * 2 <= tiles) s *= 2;   // every split keeps at least one 64-wide tile
    // ... and no fp32 accumulation chain of the second product runs over more than chain_max streamed indices: a workgroup's accumulators are a sequential
    // fmaf chain over its slice of the streamed dimension, whose rounding error grows with the square root of its length; the slabs are summed in double
    static const long chain_max = getenv("NMFX_CHAIN_MAX") ? atol(getenv("NMFX_CHAIN_MAX")) : 0;
    if (chain_max >= 64) { const long need = (tiles * 64 + chain_max - 1) / chain_max; if (need > s) s = need < tiles ? need : tiles; }
    const long per = (tiles + s - 1) / s;                   // tiles per split; trailing splits that would be empty are dropped
    *c_per_split = per * 64;
    return (int)((tiles + per - 1) / per);
}
}  // namespace nmfx

namespace {

constexpr double GRAM_COST_RATIO_MIN = 0.05;   // cost / (0.5*||V||^2) below which the explicit residual pass takes over (error bound there: 3e-9*2/0.05 = 1.2e-7 relative)
// is THIS W step in Gram-cost mode?  Host-side and the same on every rank AND in every run: capability, not latched to classic, (on shards) the global norm
// known -- and the flag as it stood after the decision of TWO W updates ago.  gram_decide publishes the state of the flag after every decision with a
// sequence stamp; W step j waits for the stamp of decision j-2 (the work of a whole iteration is queued behind it, so the wait never drains the device) and
// latches on exactly that value.  Decision j-1 may or may not have run by then -- it is never looked at, so neither host timing nor the rank can change where
// the engine switches kernels (round 3 read "whatever the flag is now": ranks could switch at different iterations and mix cost partials of two modes).
inline bool gram_active(nmfx_engine *e) {
    if (!e->gram_cost || e->classic) return false;
    if (e->dist_seen && !e->sumvv_global_set) return false;
    if (e->decide_seq >= 2) {
        const unsigned want = e->decide_seq - 2;
        const int stamp = (int)((want + 1) & 0x3fffffffu);
        int *slot = e->exact_flag_host + (want & 7u);
        int v = __atomic_load_n(slot, __ATOMIC_ACQUIRE);
        for (unsigned long spins = 0; (v >> 1) != stamp; ++spins) {
            if ((spins & 0xfffffu) == 0xfffffu && hipStreamQuery(e->st) != hipErrorNotReady) {   // the stream is idle (or dead) and the stamp is not there: do not hang on it
                v = __atomic_load_n(slot, __ATOMIC_ACQUIRE);
                if ((v >> 1) != stamp) { (void)hipGetLastError(); v = 0; }
                break;
            }
            __builtin_ia32_pause();
            v = __atomic_load_n(slot, __ATOMIC_ACQUIRE);
        }
        if (v & 1) { e->classic = true; return false; }   // from now on the one-pass kernel with the cost inside
    }
    return true;
}

// fused W-step pass (K2) or cost-only pass over rows [row0, row0 + rows) of the local shard.  N of those rows goes to `out`
// as a contiguous rows x K block; cost partials are appended at e->chunk_parts.
// no_cost: euclidean numerators only (R = V, no first product): the cost comes in Gram form from the W update that follows.
// run_if: conditional cost-only launch (see FusedParams.run_if)
nmfx_status fused_wpass_rows(nmfx_engine *e, bool do_g2, long row0, long rows, float *out, bool no_cost = false, const int *run_if = nullptr) {
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
    f.run_if = run_if;
    int func = e->div == NMFX_DIV_KL ? 3 : (no_cost ? 0 : 1);
    float *out2 = nullptr;
    if (e->dual) {   // IS / alpha-beta: the denominators come out of the same pass, into the second half of `packed`
        if (rows != e->m) { set_error("fused IS / alpha-beta W step: row chunks are not supported"); return NMFX_ERR_UNSUPPORTED; }
        func = mdiv(e) == NMFX_DIV_IS ? 4 : 5;
        out2 = out + (size_t)e->m * e->K;
        f.out2 = split == 1 ? out2 : e->slabs2;
        f.ab_alpha = (float)e->alpha; f.ab_beta = (float)e->beta; f.inv_exp = 1.0f;
        if (e->Valpha) f.D = e->Valpha + row0;
    }
    if (e->dual2) {
        // K > 192: numerators (+ the cost terms) and denominators in two passes; the second one writes where the dual-map kernel's second accumulator set
        // would have gone.  With the m x n scratch (e->Vhat) the first pass also leaves the second map's values there -- by-products of the first map -- and the
        // second pass contracts them WITHOUT forming S again (functor 0 on that buffer): 4 + 2 = 6*m*n*K instead of 8
        func = e->dualz ? 17 : (mdiv(e) == NMFX_DIV_IS ? 11 : 13);
        const bool stb = do_g2 && !e->dualz && e->Vhat != nullptr && rows == e->m;
        f.out2 = nullptr;
        if (stb) f.Rout = e->Vhat;
        if (e->dualz) f.D = e->V + row0;   // (the Valpha slot holds V.^(beta-1), the operand of the SECOND pass)
        if (e->dualz && !do_g2) {
            // the dual form has no cost to reduce (the reference divides by alpha*beta = 0: the finisher reproduces its +-Inf from the element count alone)
            NMFX_HIP(hipMemsetAsync(e->cost_partials + e->chunk_parts, 0, sizeof(double) * (size_t)(blocks * split), e->st));
        } else {
            Scope s(e, do_g2 ? TAG_FUSED_W : TAG_FUSED_COST);
            TRY(launch_fused(e->st, f, split, true, stb ? (func == 11 ? 15 : 16) : func, do_g2, 0));
        }
        if (do_g2) {
            FusedParams g = f;
            g.out = split == 1 ? out2 : e->slabs2;
            g.cost_partials = nullptr; g.Rout = nullptr;
            if (stb) g.D = e->Vhat;
            if (e->dualz) g.D = e->Valpha + row0;
            Scope s(e, TAG_FUSED_W);
            TRY(launch_fused(e->st, g, split, true, (stb || e->dualz) ? 0 : func + 1, true, 0));
        }
    } else {
        Scope s(e, run_if ? TAG_SMALL : (do_g2 ? TAG_FUSED_W : TAG_FUSED_COST));   // (a conditional launch is a no-op most of the time: not worth an event pair)
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
    e->wstep_gram = do_g2 && gram_active(e);
    if (e->wstep_gram) {         // numerators only; the cost of the state this step starts from follows in wstep_finish (Gram form)
        e->cost_valid = false;
        return fused_wpass_rows(e, true, 0, e->m, e->packed, true);
    }
    TRY(fused_wpass_rows(e, do_g2, 0, e->m, e->packed));
    return fused_wpass_finish(e);
}

// cnmf fused passes (fused_kernel TT > 1) over the local columns: do_g2 -> N_all = V * H_stack' into `out` (m x KT), else the residual cost
// partials of the CURRENT (W, H).  H's T-1 columns to the left of the shard are its halo, or zeros (Hpad) on the first / only shard.
enum FusedTMode { FT_NUM = 0, FT_COST_EUC = 1, FT_S_KL = 2, FT_COST_KL = 3, FT_S_DUAL = 4, FT_COST_DUAL = 5 };   // 4 / 5: IS / alpha-beta, both element maps stored / cost only
nmfx_status ensure_hpad(nmfx_engine *e) {   // Hpad = [T-1 zero columns | H | T-1 zero columns (lag-form Gram products only)]
    if (e->hpad_valid) return NMFX_OK;      // H changed since the last pass (init, H step)
    Scope s(e, TAG_SMALL);
    TRY(pad_left(e->st, e->H, e->K, e->n + ((e->fusedT_kl || e->fusedT_dual) ? e->hR : 0), e->T - 1, e->Hpad, e->lagram ? e->T - 1 : 0));   // (KL shards: the S pass also runs over the right-halo columns)
    e->hpad_valid = true;
    return NMFX_OK;
}
nmfx_status fusedT_pass(nmfx_engine *e, int mode, float *out, const int *run_if = nullptr, const float *Dnum = nullptr) {   // Dnum: the numerator pass's data operand (default: V, KL: R)
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
    if (mode == FT_NUM && Dnum) f.D = Dnum;
    if (mode == FT_S_KL) f.Rout = e->Vhat;
    if (mode == FT_S_DUAL || mode == FT_COST_DUAL) {   // A = V./S.^2 | V.^a.*S.^(b-1) (+ the cost terms), B = 1./S | S.^(a+b-1)
        f.ab_alpha = (float)e->alpha; f.ab_beta = (float)e->beta; f.inv_exp = 1.0f;
        if (e->Valpha) f.D = e->Valpha;
        if (mode == FT_S_DUAL) { f.Rout = e->Vhat; f.Rout2 = e->Vhat2; }
    }
    f.c_per_split = e->cps_T;
    const long mKT = e->m * (long)e->KT;
    f.out = e->nsplit_T == 1 ? out : e->slabsT;
    f.slab_stride = mKT; f.os_r = 1; f.os_k = e->m; f.os_t = e->m * (long)e->K;
    f.cost_partials = do_g2 ? nullptr : e->cost_partials;
    f.run_if = run_if;
    const int func = mode == FT_NUM ? 0 : (mode == FT_COST_EUC ? 1 : ((mode == FT_S_DUAL || mode == FT_COST_DUAL) ? (e->div == NMFX_DIV_IS ? 11 : 13) : 3));
    {
        Scope s(e, run_if ? TAG_SMALL : (do_g2 ? TAG_FUSED_W : TAG_FUSED_COST));
        TRY(launch_fused(e->st, f, e->nsplit_T, true, func, do_g2, 0));
    }
    if (do_g2 && e->nsplit_T > 1) {
        Scope s(e, TAG_SMALL);
        TRY(reduce_slabs(e->st, e->slabsT, e->nsplit_T, mKT, mKT, out, 0));
    }
    if (!do_g2) e->n_cost_used = (int)((e->m + 127) / 128) * e->nsplit_T;
    if (mode == FT_S_DUAL && e->hR > 0) {   // column shard: both element maps' values on the T-1 right-halo columns too, as for KL below
        FusedParams h = f;
        h.Y = Hy + (size_t)e->K * e->n; h.D = (e->Valpha ? e->Valpha : e->V) + (size_t)e->m * e->n;
        h.Rout = e->Vhat + (size_t)e->m * e->n; h.Rout2 = e->Vhat2 + (size_t)e->m * e->n;
        h.Cn = e->hR; h.c_per_split = 64; h.cost_partials = nullptr; h.out = nullptr;
        Scope s(e, TAG_SMALL);
        TRY(launch_fused(e->st, h, 1, true, func, false, 0));
    }
    if (mode == FT_S_KL && e->hR > 0) {
        // column shard: R = V./V_hat on the T-1 right-halo columns too (Q((t,k), j+t) of the last local columns reads them, cnmf.m:219); they belong to
        // the neighbour's cost, so this second, tiny launch carries none
        FusedParams h = f;
        h.Y = Hy + (size_t)e->K * e->n; h.D = e->V + (size_t)e->m * e->n; h.Rout = e->Vhat + (size_t)e->m * e->n;
        h.Cn = e->hR; h.c_per_split = 64; h.cost_partials = nullptr; h.out = nullptr;
        Scope s(e, TAG_SMALL);
        TRY(launch_fused(e->st, h, 1, true, 3, false, 0));
    }
    return NMFX_OK;
}
// KL cnmf on the fused passes: the cost of the CURRENT (W, H) from the S pass's partials (cnmf.m:243, every term from the pass's own S; NMFX_KL_MODE 0 summed
// V.*log(V./V_hat) only and added sum(V_hat) - sum(V) = sum_{t,k} colsum(W_t)_k * sum_{j < n-t} H(k, j) - sum(V) in closed form)
nmfx_status fusedT_kl_cost(nmfx_engine *e) {
    Scope s(e, TAG_SMALL);
    if (!KL_CONSISTENT_COST) {
        TRY(row_reduce(e->st, e->H, e->K, e->K, e->n, 0, e->rowsum, e->rr_scratch));
        TRY(kl_pvec(e->st, e->rowsum, e->H, e->K, e->n, e->T, e->Pvec, e->hL));   // sum over the shard's own columns j of H(k, j - t): reaches into the left halo
        TRY(col_reduce(e->st, e->W, e->m, e->m, e->KT, 0, e->colsum));
    }
    const bool useW = e->any_lamW && e->rank0, useH = e->any_lamH;
    if (useW) TRY(col_reduce(e->st, e->W, e->m, e->m, e->KT, 2, e->l1W));
    if (useH) TRY(row_reduce(e->st, e->H, e->K, e->K, e->n, 2, e->l1H, e->rr_scratch));
    TRY(finish_cost(e->st, e->cost_partials, e->n_cost_used, 1.0, useW ? e->l1W : nullptr, e->KT, e->lamW, useH ? e->l1H : nullptr, e->K, e->lamH, e->cost,
                    KL_CONSISTENT_COST ? nullptr : e->colsum, e->Pvec, e->KT, e->sumV_g, nullptr, 0.0, 0.0, e->cost_dst2));
    e->cost_valid = true;
    return NMFX_OK;
}

// KL with K > 256 (e->klw).  S pass: S = W*H accumulated block by block in the V_hat buffer; the last block maps it to R = V./S (kept there for the
// numerator passes when store_R) and reduces sum(V.*log(V./S)) when with_cost
nmfx_status klw_s_pass(nmfx_engine *e, bool store_R, bool with_cost) {
    Scope s(e, TAG_FUSED_COST);
    for (int b = 0; b < e->klw_nb; ++b) {
        const bool last = b + 1 == e->klw_nb;
        const int kb = e->klw_kb[b];
        long cps = 0;
        const int split = fused_split((e->m + 127) / 128, e->n, kb, &cps);
        if ((long)((e->m + 127) / 128) * split > e->n_cost_partials) { set_error("klw_s_pass: cost partials too small"); return NMFX_ERR_INVALID; }
        FusedParams f;
        memset(&f, 0, sizeof(f));
        f.X = e->W + (size_t)e->m * e->klw_k0[b]; f.xs_r = 1; f.xs_k = e->m;
        f.Y = e->H + e->klw_k0[b]; f.y_stride = e->K;
        f.D = e->V; f.ldd = e->m; f.R = e->m; f.Cn = e->n; f.K = kb; f.c_per_split = cps;
        f.Sin = b > 0 ? e->Vhat : nullptr;
        f.Rout = (!last || store_R) ? e->Vhat : nullptr;
        f.cost_partials = (last && with_cost) ? e->cost_partials : nullptr;
        int func_last = 8;
        if (e->dualw) {   // IS / alpha-beta: the last block runs map 11 / 13 on the accumulated S and stores both maps' values (A over the partial sums, B next to them)
            func_last = e->div == NMFX_DIV_IS ? 19 : 20;
            f.ab_alpha = (float)e->alpha; f.ab_beta = (float)e->beta; f.inv_exp = 1.0f;
            if (last && e->Valpha) f.D = e->Valpha;
            if (last && store_R) f.Rout2 = e->Vhat2;
        }
        TRY(launch_fused(e->st, f, split, true, last ? func_last : 7, false, 0));
        if (last) e->n_cost_used = (int)((e->m + 127) / 128) * split;
    }
    return NMFX_OK;
}
// N(:, block) = R * H(block, :)' for every column block, R = V./S in the V_hat buffer: the W-step form without a first product
nmfx_status klw_num_pass(nmfx_engine *e, float *out, const float *D) {
    for (int b = 0; b < e->klw_nb; ++b) {
        const int kb = e->klw_kb[b];
        long cps = 0;
        int split = fused_split((e->m + 127) / 128, e->n, kb, &cps);
        if (split > e->nsplit_T) { split = e->nsplit_T; cps = e->cps_T; }
        FusedParams f;
        memset(&f, 0, sizeof(f));
        f.Y = e->H + e->klw_k0[b]; f.y_stride = e->K;
        f.D = D; f.ldd = e->m; f.R = e->m; f.Cn = e->n; f.K = kb; f.c_per_split = cps;
        float *dst = out + (size_t)e->m * e->klw_k0[b];
        f.out = split == 1 ? dst : e->slabsT; f.slab_stride = e->m * (long)kb; f.os_r = 1; f.os_k = e->m;
        {
            Scope s(e, TAG_FUSED_W);
            TRY(launch_fused(e->st, f, split, true, 0, true, 0));
        }
        if (split > 1) { Scope s(e, TAG_SMALL); TRY(reduce_slabs(e->st, e->slabsT, split, f.slab_stride, f.slab_stride, dst, 0)); }
    }
    return NMFX_OK;
}
// euclidean, K > 256 (e->eucw): the explicit residual sum (V - W*H).^2 behind the Gram-form cost -- the S chain above with functor 10 closing it; run_if: the
// device-side flag of the conditional launch
nmfx_status eucw_cost_pass(nmfx_engine *e, const int *run_if) {
    Scope s(e, run_if ? TAG_SMALL : TAG_FUSED_COST);
    for (int b = 0; b < e->klw_nb; ++b) {
        const bool last = b + 1 == e->klw_nb;
        long cps = 0;
        const int split = fused_split((e->m + 127) / 128, e->n, e->klw_kb[b], &cps);
        FusedParams f;
        memset(&f, 0, sizeof(f));
        f.X = e->W + (size_t)e->m * e->klw_k0[b]; f.xs_r = 1; f.xs_k = e->m;
        f.Y = e->H + e->klw_k0[b]; f.y_stride = e->K;
        f.D = e->V; f.ldd = e->m; f.R = e->m; f.Cn = e->n; f.K = e->klw_kb[b]; f.c_per_split = cps;
        f.Sin = b > 0 ? e->Vhat : nullptr;
        f.Rout = last ? nullptr : e->Vhat;
        f.cost_partials = last ? e->cost_partials : nullptr;
        f.run_if = run_if;
        TRY(launch_fused(e->st, f, split, true, last ? 10 : 7, false, 0));
        if (last) e->n_cost_used = (int)((e->m + 127) / 128) * split;
    }
    return NMFX_OK;
}
// ... and its H-step numerator Gn(block, :) = (V' * W(:, block))' on the transposed copy of V (as the K <= 256 path does)
nmfx_status eucw_hnum(nmfx_engine *e) {
    {
        Scope s(e, TAG_SMALL);
        TRY(transpose_f32(e->st, e->W, e->m, e->K, e->WT));
    }
    const long Kn = (long)e->K * e->n;
    const int split = e->klw_hsplit;
    for (int b = 0; b < e->klw_nb; ++b) {
        FusedParams g;
        memset(&g, 0, sizeof(g));
        g.Y = e->WT + e->klw_k0[b]; g.y_stride = e->K;
        g.D = e->VT; g.ldd = e->n; g.R = e->n; g.Cn = e->m; g.K = e->klw_kb[b]; g.c_per_split = e->klw_hcps;
        g.out = (split == 1 ? e->Gn : e->slabsH) + e->klw_k0[b]; g.slab_stride = Kn; g.os_r = e->K; g.os_k = 1;
        Scope s(e, TAG_HNUM);
        TRY(launch_fused(e->st, g, split, true, 0, true, 0));
    }
    if (split > 1) { Scope s(e, TAG_SMALL); TRY(reduce_slabs(e->st, e->slabsH, split, Kn, Kn, e->Gn, 0)); }
    return NMFX_OK;
}
// H step on the transposed copy of V: rows of V' (columns j of V) stationary, rows of W (the W' copy) streamed.  R' = V'./(H'*W') block by block into the V_hat
// buffer (as n x m), then Gn(block, :) = (R'*W(:, block))' on the same kernel without a first product
nmfx_status klw_hstep_vt(nmfx_engine *e) {
    {
        Scope s(e, TAG_SMALL);
        TRY(transpose_f32(e->st, e->W, e->m, e->K, e->WT));
    }
    {
        Scope s(e, TAG_FUSED_COST);
        for (int b = 0; b < e->klw_nb; ++b) {
            const bool last = b + 1 == e->klw_nb;
            long cps = 0;
            const int split = fused_split((e->n + 127) / 128, e->m, e->klw_kb[b], &cps);
            FusedParams f;
            memset(&f, 0, sizeof(f));
            f.X = e->H + e->klw_k0[b]; f.xs_r = e->K; f.xs_k = 1;
            f.Y = e->WT + e->klw_k0[b]; f.y_stride = e->K;
            f.D = e->VT; f.ldd = e->n; f.R = e->n; f.Cn = e->m; f.K = e->klw_kb[b]; f.c_per_split = cps;
            f.Sin = b > 0 ? e->Vhat : nullptr;
            f.Rout = e->Vhat;
            TRY(launch_fused(e->st, f, split, true, last ? 8 : 7, false, 0));
        }
    }
    const long Kn = (long)e->K * e->n;
    int split = e->klw_hsplit;
    for (int b = 0; b < e->klw_nb; ++b) {
        FusedParams g;
        memset(&g, 0, sizeof(g));
        g.Y = e->WT + e->klw_k0[b]; g.y_stride = e->K;
        g.D = e->Vhat; g.ldd = e->n; g.R = e->n; g.Cn = e->m; g.K = e->klw_kb[b]; g.c_per_split = e->klw_hcps;
        g.out = (split == 1 ? e->Gn : e->slabsH) + e->klw_k0[b]; g.slab_stride = Kn; g.os_r = e->K; g.os_k = 1;
        Scope s(e, TAG_HNUM);
        TRY(launch_fused(e->st, g, split, true, 0, true, 0));
    }
    if (split > 1) { Scope s(e, TAG_SMALL); TRY(reduce_slabs(e->st, e->slabsH, split, Kn, Kn, e->Gn, 0)); }
    return NMFX_OK;
}
// ... and the cost of the CURRENT (W, H) from the S pass's partials (closed-form sum(V_hat) - sum(V), like the fused path)
nmfx_status klw_cost(nmfx_engine *e) {
    Scope s(e, TAG_SMALL);
    TRY(row_reduce(e->st, e->H, e->K, e->K, e->n, 0, e->rowsum, e->rr_scratch));
    TRY(col_reduce(e->st, e->W, e->m, e->m, e->K, 0, e->Gpvec));
    TRY(cost_from_partials(e, e->n_cost_used, true));
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
void nmfx_abi_sizes(int32_t *problem_bytes, int32_t *result_bytes, int32_t *engine_desc_bytes) {
    if (problem_bytes) *problem_bytes = (int32_t)sizeof(nmfx_problem);
    if (result_bytes) *result_bytes = (int32_t)sizeof(nmfx_result);
    if (engine_desc_bytes) *engine_desc_bytes = (int32_t)sizeof(nmfx_engine_desc);
}
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
    if (e->eucw && layout(e, nullptr).total > workspace_bytes) { e->eucw = false; e->gram_cost = false; }   // (sized without V': the plain Gram path)
    if ((e->use_vt || e->use_vtq || e->klw_vt) && layout(e, nullptr).total > workspace_bytes) e->use_vt = e->use_vtq = e->klw_vt = false;
    else if (!e->use_vt && !e->use_vtq && !e->klw_vt) {   // ... and the other way round: memory looked tight now, but the workspace was sized with the copy
        nmfx_engine probe = *e;
        probe.use_vt = probe.fused && (probe.div == NMFX_DIV_EUCLIDEAN || (probe.dual2 && !probe.dualz && dual2_store())) && getenv("NMFX_NO_VT") == nullptr;
        probe.use_vtq = probe.fusedT && probe.qgemm && probe.hL == 0 && probe.hR == 0 && getenv("NMFX_NO_VT") == nullptr && probe.KT % probe.vtq_block == 0 && fused_supported(probe.vtq_block);
        probe.klw_vt = probe.klw && getenv("NMFX_NO_VT") == nullptr;
        if ((probe.use_vt || probe.use_vtq || probe.klw_vt) && layout(&probe, nullptr).total <= workspace_bytes) { e->use_vt = probe.use_vt; e->use_vtq = probe.use_vtq; e->klw_vt = probe.klw_vt; }
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
    if (he != hipSuccess) {
        set_error("nmfx_engine_create: %s", hipGetErrorString(he));
        if (hipStreamSynchronize(e->st) != hipSuccess) (void)hipGetLastError();   // (a copy queued before the one that failed may still be reading the vectors)
        delete e;
        return NMFX_ERR_HIP;
    }
    if (e->all_fixW) e->gram_cost = false;   // no W update, no column statistics
    if (e->gram_cost) {
        he = hipHostMalloc(reinterpret_cast<void **>(&e->exact_flag_host), 64, hipHostMallocMapped | hipHostMallocPortable);
        if (he != hipSuccess) { (void)hipGetLastError(); e->exact_flag_host = nullptr; e->gram_cost = false; }
        else memset(e->exact_flag_host, 0, 64);
    }
    *out = e;
    return NMFX_OK;
}

void nmfx_engine_destroy(nmfx_engine *e) {
    if (!e) return;
    if (e->seg_dev) (void)hipFree(e->seg_dev);
    if (e->exact_flag_host) (void)hipHostFree(e->exact_flag_host);
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
    StreamDrain drain_(e->st);
    NMFX_HIP(hipMalloc(&e->seg_dev, sizeof(long) * (nz + 1)));
    NMFX_HIP(hipMemcpyAsync(e->seg_dev, sg.data(), sizeof(long) * (nz + 1), hipMemcpyHostToDevice, e->st));
    NMFX_HIP(hipStreamSynchronize(e->st));
    e->Z = Z_dev; e->nz = nz;
    return NMFX_OK;
}

nmfx_status nmfx_engine_set_rank0(nmfx_engine *e, int32_t is_rank0) { e->rank0 = is_rank0; e->dist_seen = true; return NMFX_OK; }   // (only sharded callers say which rank they are)
// Column shards + Gram-form cost: the mode decision needs the GLOBAL ||V||^2 and must be identical on every rank.  After nmfx_engine_init,
// nmfx_engine_sumvv_local copies this shard's ||V_local||^2 (fp64) to dst_dev (0.0 when the engine has no such mode); the caller sums over the
// ranks (one 8-byte all-reduce, once) and hands the result back with nmfx_engine_sumvv_set_global.  Until then a sharded engine keeps the explicit pass.
nmfx_status nmfx_engine_sumvv_local(nmfx_engine *e, double *dst_dev) {
    if (e->gram_cost) NMFX_HIP(hipMemcpyAsync(dst_dev, e->sumVV, sizeof(double), hipMemcpyDeviceToDevice, e->st));
    else NMFX_HIP(hipMemsetAsync(dst_dev, 0, sizeof(double), e->st));
    return NMFX_OK;
}
nmfx_status nmfx_engine_sumvv_set_global(nmfx_engine *e, const double *src_dev) {
    if (e->gram_cost) NMFX_HIP(hipMemcpyAsync(e->sumVV + 1, src_dev, sizeof(double), hipMemcpyDeviceToDevice, e->st));
    e->sumvv_global_set = true;
    return NMFX_OK;
}
// W64 <- W, H64 <- H (own columns): the masters follow the fp32 arrays (init; a caller that rewrote W / H itself)
nmfx_status nmfx_engine_sync_master(nmfx_engine *e) {
    DeviceGuard dg_;
    NMFX_HIP(hipSetDevice(e->device));
    TRY(cvt_to_f64(e->st, e->W, e->W64, (long)e->m * e->KT));
    if (e->H64) TRY(cvt_to_f64(e->st, e->H, e->H64, (long)e->K * e->n));
    return NMFX_OK;
}
nmfx_status nmfx_engine_master_ptrs(nmfx_engine *e, double **W64_dev, double **H64_dev) {
    if (W64_dev) *W64_dev = e->W64;
    if (H64_dev) *H64_dev = e->H64;
    return NMFX_OK;
}
// 0: the cost of iteration i is ready after hstep(i); 1: after wstep_partial(i+1); 2: after wstep_finish(i+1) (read it there; engines of kind 2
// may also deliver it at point 1 -- reading at point 2 is always right for them)
int32_t nmfx_engine_cost_lag(nmfx_engine *e) { return e->gram_cost ? 2 : ((e->fused || e->fusedT_kl || e->fusedT_dual || e->klw || e->dualw) ? 1 : 0); }

// nmf.m:130-139 / cnmf.m:155-171: normalise W (all sources, fixed or not), cnmf also rescales H; then V_hat.
// The normalisation runs on the float64 masters: one fp32 rounding of the INITIAL state is a perturbation the iteration carries to the end, and problems that
// amplify it (one factor fixed, planted data: x300) left the contract through exactly that door (scripts/diag_wstep.py: 1.7e-5 from an fp32-normalised W0,
// 5e-6 from fl32(W0) normalised in double, 1.5e-6 from the caller's float64 W0) -- hence also nmfx_engine_init_f64
static nmfx_status engine_init_impl(nmfx_engine *e, const double *W0, const double *H0) {
    DeviceGuard dg_;
    NMFX_HIP(hipSetDevice(e->device));
    e->hpad_valid = false;
    if (e->algo == 3) {
        if (!e->Z) { set_error("nmfx_engine_init: constrainednmf needs nmfx_engine_set_constraint first"); return NMFX_ERR_INVALID; }
        TRY(z_update(e->st, e->Z, e->H, nullptr, nullptr, nullptr, e->K, e->nz, e->seg_dev, nullptr, nullptr, 1.0f, 1));   // H = Z*A, constrainednmf.m:177
    }
    {
        Scope s(e, TAG_SMALL);
        const size_t mKT = (size_t)e->m * e->KT, Kn = (size_t)e->K * e->n;
        if (W0) NMFX_HIP(hipMemcpyAsync(e->W64, W0, sizeof(double) * mKT, hipMemcpyDeviceToDevice, e->st));
        else TRY(cvt_to_f64(e->st, e->W, e->W64, (long)mKT));
        if (e->H64) {
            if (H0) { NMFX_HIP(hipMemcpyAsync(e->H64, H0, sizeof(double) * Kn, hipMemcpyDeviceToDevice, e->st)); TRY(cvt_f64_to_f32(e->st, e->H64, e->H, (long)Kn)); }
            else TRY(cvt_to_f64(e->st, e->H, e->H64, (long)Kn));
        }
        TRY(col_reduce64(e->st, e->W64, e->m, e->m, e->KT, e->algo == 2 ? 0 : 1, e->sumsq));   // lnmf.m:59: L1 sums
        TRY(w_normalize(e->st, e->W, e->m, e->K, e->T, e->sumsq, nullptr, norm_mode(e), e->f_out, e->K_valid, e->W64));
        if (e->algo == 1) {   // cnmf.m:165; the halos too (fp32 copies of the neighbours' columns: every rank applies the same factors)
            TRY(scale_rows64(e->st, e->H64, e->H, e->K, e->n, e->f_out));
            if (e->hL) TRY(scale_rows(e->st, e->Hext, e->K, e->hL, e->f_out));
            if (e->hR) TRY(scale_rows(e->st, e->H + (size_t)e->K * e->n, e->K, e->hR, e->f_out));
        }
        if (e->fused) {
            e->cost_valid = false;
            if (e->VT) TRY(transpose_f32(e->st, e->V, e->m, e->n, e->VT));   // once: V is constant over the iterations
            if (e->div == NMFX_DIV_KL) {   // sum(V_local), once
                TRY(col_reduce(e->st, e->V, e->m, e->m, (int)e->n, 0, e->colV));
                TRY(sum_vec(e->st, e->colV, e->n, e->sumV));
            }
            if (e->gram_cost) {            // ||V_local||^2, once; the global norm defaults to it (one shard)
                TRY(col_reduce(e->st, e->V, e->m, e->m, (int)e->n, 1, e->colV));
                TRY(sum_vec(e->st, e->colV, e->n, e->sumVV));
                NMFX_HIP(hipMemcpyAsync(e->sumVV + 1, e->sumVV, sizeof(double), hipMemcpyDeviceToDevice, e->st));
                NMFX_HIP(hipMemsetAsync(e->exact_flag, 0, 64, e->st));
                NMFX_HIP(hipStreamSynchronize(e->st));   // decisions of an earlier run of this engine may still be queued: their late writes into the stamped slots would
                memset(e->exact_flag_host, 0, 64);       // satisfy the NEW run's wait for a decision of the same number (stamps restart at 1)
                e->classic = false; e->decide_seq = 0;
            }
            if (e->dual && e->div == NMFX_DIV_AB) {   // sum(V.^(alpha+beta)) for the cost, V.^alpha as the kernels' data operand; once
                TRY(col_reduce_pow(e->st, e->V, e->m, e->m, (int)e->n, (float)(e->alpha + e->beta), e->colV));
                TRY(sum_vec(e->st, e->colV, e->n, e->sumVab));
                if (e->Valpha) TRY(pow_map(e->st, e->V, e->Valpha, (long)e->m * e->n, e->dualz ? (float)(e->beta - 1.0) : (float)e->alpha));   // (dual form: the denominator operand V.^(alpha+beta-1))
                if (e->VTa) TRY(transpose_f32(e->st, e->Valpha, e->m, e->n, e->VTa));
            }
            return refresh_w_derived(e);
        }
    }
    if (e->gram) {                 // no V_hat state on the Gram path
        if (e->use_vtq || e->eucw) TRY(transpose_f32(e->st, e->V, e->m, e->n, e->VT));   // once: V is constant over the iterations
        if (e->gram_cost) {        // ||V||^2, once
            Scope s(e, TAG_SMALL);
            TRY(col_reduce(e->st, e->V, e->m, e->m, (int)e->n, 1, e->colV));
            TRY(sum_vec(e->st, e->colV, e->n, e->sumVV));
            NMFX_HIP(hipMemcpyAsync(e->sumVV + 1, e->sumVV, sizeof(double), hipMemcpyDeviceToDevice, e->st));
            NMFX_HIP(hipMemsetAsync(e->exact_flag, 0, 64, e->st));
            NMFX_HIP(hipStreamSynchronize(e->st));   // (as above: no decision of an earlier run may land in the slots after this)
            memset(e->exact_flag_host, 0, 64);
            e->classic = false; e->decide_seq = 0;
            e->cost_valid = false;
        }
        return NMFX_OK;
    }
    if (e->klw) {                  // nor here
        Scope s(e, TAG_SMALL);
        e->cost_valid = false;
        TRY(col_reduce(e->st, e->V, e->m, e->m, (int)e->n, 0, e->colV));
        if (e->klw_vt) TRY(transpose_f32(e->st, e->V, e->m, e->n, e->VT));   // once: V is constant over the iterations
        return sum_vec(e->st, e->colV, e->n, e->sumV);
    }
    if (e->fusedT_dual || e->dualw) {   // nor here: the constants of the alpha-beta cost and V.^alpha, the S pass's data operand, once
        Scope s(e, TAG_SMALL);
        e->cost_valid = false;
        if (e->div == NMFX_DIV_AB) {
            TRY(col_reduce_pow(e->st, e->V, e->m, e->m, (int)e->n, (float)(e->alpha + e->beta), e->colV));
            TRY(sum_vec(e->st, e->colV, e->n, e->sumVab));
            if (e->Valpha) TRY(pow_map(e->st, e->V, e->Valpha, (long)e->m * (e->n + e->hR), (float)e->alpha));   // (the right-halo columns of a cnmf shard too)
        }
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
nmfx_status nmfx_engine_init(nmfx_engine *e) { return engine_init_impl(e, nullptr, nullptr); }
nmfx_status nmfx_engine_init_f64(nmfx_engine *e, const double *W_init64_dev, const double *H_init64_dev) { return engine_init_impl(e, W_init64_dev, H_init64_dev); }

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
        TRY(gram_fused(e, e->H, e->n, e->packed + mKT));
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
    if (chunk == 0) { e->chunk_parts = 0; e->w_chunks = nchunks; e->cost_valid = false; e->wstep_gram = false; }
    if (chunk == 0 && e->gram_cost && !e->classic) {   // row chunks carry their cost inside: the Gram-form mode is switched off for good (every rank chunks alike)
        static const int one = 1;
        NMFX_HIP(hipMemcpyAsync(e->exact_flag, &one, sizeof(int), hipMemcpyHostToDevice, e->st));
        e->classic = true;
    }
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
    if (e->fusedT_dual) {   // S pass: both element maps' values into the two m x n buffers + the (lagged) cost of the state this iteration starts from
        TRY(fusedT_pass(e, e->all_fixW ? FT_COST_DUAL : FT_S_DUAL, nullptr));
        Scope s(e, TAG_SMALL);
        TRY(cost_from_partials(e, e->n_cost_used));
        e->cost_valid = true;
    }
    if (e->fusedT_kl) {   // S pass: R = V./V_hat into the V_hat buffer + the (lagged) cost of the state this iteration starts from
        TRY(fusedT_pass(e, e->all_fixW ? FT_COST_KL : FT_S_KL, nullptr));
        TRY(fusedT_kl_cost(e));
    }
    if (e->dualw) {       // IS / alpha-beta nmf with K > 256: S in column blocks, both element maps' values out of the last one, the (lagged) cost
        TRY(klw_s_pass(e, !e->all_fixW, true));
        Scope s(e, TAG_SMALL);
        TRY(cost_from_partials(e, e->n_cost_used));
        e->cost_valid = true;
    }
    if (e->klw) {         // the same for nmf with K > 256
        TRY(klw_s_pass(e, !e->all_fixW, true));
        TRY(klw_cost(e));
    }
    if (e->eucw && e->gram_cost && e->dist_seen && !e->sumvv_global_set) {
        set_error("nmfx_engine: column shards in Gram-form cost mode need nmfx_engine_sumvv_set_global first");
        return NMFX_ERR_INVALID;
    }
    e->wstep_gram = (e->fusedT || e->eucw) && e->gram_cost;   // the cost of the state this step starts from follows in wstep_finish (no host latch here: once the device flag
                                                 // is set the conditional residual pass simply runs every time -- the same work the un-lagged cost pass was)
    if (e->all_fixW) return NMFX_OK;
    OpView a{}, b{};
    num_view(e, a);
    if (e->fusedT_dual) {   // [N | P] = [A | B] * H_stack': two numerator passes, no first product
        TRY(fusedT_pass(e, FT_NUM, e->packed, nullptr, e->Vhat));
        return fusedT_pass(e, FT_NUM, e->packed + mKT, nullptr, e->Vhat2);
    }
    if (e->fusedT || e->fusedT_kl) TRY(fusedT_pass(e, FT_NUM, e->packed));   // all T numerators in one pass over V (KL: over R), the shifted H tile in LDS
    else if (e->dualw) {   // [N | P] = [A | B] * H', column block by column block
        TRY(klw_num_pass(e, e->packed, e->Vhat));
        return klw_num_pass(e, e->packed + mKT, e->Vhat2);
    }
    else if (e->klw) TRY(klw_num_pass(e, e->packed, e->Vhat));
    else if (e->eucw) TRY(klw_num_pass(e, e->packed, e->V));
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
    } else if (e->p1gram) {
        // path 1 keeps V_hat for the cost and the H step, but V_hat*H' (nmf.m:150) as an fp32 product of the fp32 V_hat is what put over-complete problems with
        // H fixed at 1e-5 ... 3e-5 on W (profiles/r5_05_fuzz_campaign_fixed_factor.log): the shard's share of it is formed as W*(H*H') in float64 from the
        // master copy of W, like every other euclidean path, and rounded ONCE into the fp32 slot that travels in the all-reduce
        Scope s(e, TAG_WDEN);
        OpView hv{e->H, nullptr, (long)e->K, VIEW_RC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f};
        TRY(small_gemm(e, e->K, e->K, e->n, hv, hv, e->CC, e->K));
        TRY(gemm64(e->st, e->m, e->K, e->K, e->W64, nullptr, e->m, nullptr, e->CC, e->K, nullptr, e->packed + mKT, e->m));
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
            Scope s(e, TAG_GRAM);   // P = W * (H*H'), float64 accumulation from the master copy of W (gemm64.hip)
            TRY(gemm64(e->st, e->m, e->K, e->K, e->W64, nullptr, e->m, nullptr, e->packed + mK, e->K, e->P64, nullptr, e->m));
            p.P64 = e->P64;
        }
        p.W64 = e->W64;
        p.rule = e->algo == 2 ? 1 : 0;
        // update, column normalisation (nmf.m:169 / lnmf.m:70) and, for KL, the column sums of the final W (H-step denominator) in ONE launch
        p.fuse_norm = norm_mode(e) == 2 ? 2 : 1;
        p.colsum_out = e->div == NMFX_DIV_KL ? e->Gpvec : nullptr;
        if (e->wstep_gram) {
            // Gram-form cost of the state this W step started from (W, H untouched so far).  Column statistics first, then the decision, then -- only
            // once the residual has become too small for fp32 to resolve it this way -- the explicit residual pass, and only then the update.
            const bool useW = e->any_lamW && e->rank0, useH = e->any_lamH;
            {
                Scope s(e, TAG_SMALL);
                if (useW) TRY(col_reduce(e->st, e->W, e->m, e->m, e->K, 2, e->l1W));
                if (useH) {
                    if (e->algo == 3) TRY(row_reduce(e->st, e->Z, e->K, e->K, e->nz, 2, e->l1H, e->rr_scratch));
                    else TRY(row_reduce(e->st, e->H, e->K, e->K, e->n, 2, e->l1H, e->rr_scratch));
                }
                p.dndp = e->dndp; p.stats_only = 1;
                TRY(w_update(e->st, p));
                TRY(gram_decide(e->st, e->dndp, e->K, e->sumVV, GRAM_COST_RATIO_MIN, e->exact_flag, e->exact_flag_host + (e->decide_seq & 7u), (int)((e->decide_seq + 1) & 0x3fffffffu)));
                e->decide_seq++;
            }
            e->chunk_parts = 0;
            TRY(fused_wpass_rows(e, false, 0, e->m, nullptr, false, e->exact_flag));   // returns at once while the flag is clear
            Scope s(e, TAG_SMALL);
            // the cost finisher rides in the update launch (its last workgroup): statistics, flag and residual partials are complete by now
            p.fin_on = 1; p.fin_nc = e->K; p.fin_sumVV = e->sumVV; p.fin_rank0 = e->rank0; p.fin_exact_flag = e->exact_flag; p.fin_partials = e->cost_partials;
            p.fin_nparts = e->chunk_parts; p.fin_l1W = useW ? e->l1W : nullptr; p.fin_nW = e->K; p.fin_lamW = e->lamW; p.fin_l1H = useH ? e->l1H : nullptr; p.fin_K = e->K;
            p.fin_lamH = e->lamH; p.fin_out = e->cost; p.fin_out2 = e->cost_dst2;
            if (e->cost_dst2) e->cost_dst2_done = true;
            p.stats_only = 0; p.stats_in = 1;
            TRY(w_update(e->st, p));
            e->cost_valid = false;
            return refresh_w_derived(e, true);
        }
        Scope s(e, TAG_SMALL);
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
        p.W64 = e->W64;
        if (e->gram) {   // P_all = W_flat * (Hs*Hs'), float64 accumulation from the master copy of W (gemm64.hip)
            TRY(gemm64(e->st, e->m, e->KT, e->KT, e->W64, nullptr, e->m, nullptr, e->packed + mKT, e->KT, e->P64, nullptr, e->m));
            p.P64 = e->P64;
        } else if (div_has_matrix_den(e->div)) p.P = e->packed + mKT;
        else {
            TRY(f2d(e->st, e->packed + mKT, e->Pvec, e->KT));
            p.Pvec = e->Pvec;
        }
        p.rule = e->algo == 2 ? 1 : 0;
        if (e->wstep_gram) {   // Gram-form cost of the state this W step started from: statistics, decision, (conditional) residual pass, cost -- then the update
            const bool useW = e->any_lamW && e->rank0, useH = e->any_lamH;
            if (useW) TRY(col_reduce(e->st, e->W, e->m, e->m, e->KT, 2, e->l1W));
            if (useH) TRY(row_reduce(e->st, e->H, e->K, e->K, e->n, 2, e->l1H, e->rr_scratch));
            p.dndp = e->dndp; p.stats_only = 1;
            TRY(w_update(e->st, p));
            TRY(gram_decide(e->st, e->dndp, e->KT, e->sumVV, GRAM_COST_RATIO_MIN, e->exact_flag, e->exact_flag_host + (e->decide_seq & 7u), (int)((e->decide_seq + 1) & 0x3fffffffu)));
            e->decide_seq++;
            if (e->eucw) TRY(eucw_cost_pass(e, e->exact_flag));
            else TRY(fusedT_pass(e, FT_COST_EUC, nullptr, e->exact_flag));
            // the cost finisher rides in the update launch (its last workgroup)
            p.fin_on = 1; p.fin_nc = e->KT; p.fin_sumVV = e->sumVV; p.fin_rank0 = e->rank0; p.fin_exact_flag = e->exact_flag; p.fin_partials = e->cost_partials;
            p.fin_nparts = e->n_cost_used; p.fin_l1W = useW ? e->l1W : nullptr; p.fin_nW = e->KT; p.fin_lamW = e->lamW; p.fin_l1H = useH ? e->l1H : nullptr; p.fin_K = e->K;
            p.fin_lamH = e->lamH; p.fin_out = e->cost; p.fin_out2 = e->cost_dst2;
            if (e->cost_dst2) e->cost_dst2_done = true;
            p.stats_only = 0; p.stats_in = 1;
        }
        TRY(w_update(e->st, p));
        TRY(w_normalize(e->st, e->W, e->m, e->K, e->T, e->sumsq, e->fixW, norm_mode(e), nullptr, 0, e->W64));
        e->cost_valid = false;
    }
    if (e->gram || e->fusedT_kl || e->fusedT_dual || e->klw || e->dualw) return NMFX_OK;
    return recon(e, false);
}

nmfx_status nmfx_engine_hstep_finish(nmfx_engine *e);
// H step + V_hat refresh + local cost partial: nmf.m:176-218 / cnmf.m:207-251
nmfx_status nmfx_engine_hstep(nmfx_engine *e) {
    DeviceGuard dg_;
    NMFX_HIP(hipSetDevice(e->device));
    if (e->fused) {
        if (e->all_fixH) return NMFX_OK;
        // euclidean: W'*V_hat = (W'*W)*H.  W'*W from the transposed copy of W (rows of W contiguous: every MFMA operand one coalesced load);
        // the product with H is folded into the H update (small_mm.hip), except for constrainednmf, whose update sums over label segments first
        const bool hug = e->div == NMFX_DIV_EUCLIDEAN && e->algo != 3 && h_update_gram_supported(e->K);
        if (e->div == NMFX_DIV_EUCLIDEAN) {
            Scope s(e, TAG_GRAM);
            TRY(gram_fused(e, e->WT, e->m, e->GW));
            if (!hug)
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
        if (func == 0 && e->VT) {
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
            if (hug) {
                TRY(h_update_gram(e->st, e->H, e->GW, e->isplit_h == 1 ? e->Gn : e->slabs, e->isplit_h, g.slab_stride, e->K, e->n, e->lamH, e->fixH, e->H64));
            } else {
            if (e->isplit_h > 1 && !fuse_sum) TRY(reduce_slabs(e->st, e->slabs, e->isplit_h, g.slab_stride, g.slab_stride, e->Gn, 0));
            if (fuse_sum) TRY(h_update(e->st, e->H, e->slabs, e->Gp, nullptr, e->K, e->n, e->lamH, e->fixH, 1.0f, e->isplit_h, g.slab_stride, e->H64));
            else if (e->algo == 3) TRY(z_update(e->st, e->Z, e->H, e->Gn, e->Gp, e->Gpvec, e->K, e->nz, e->seg_dev, e->lamH, e->fixH, 1.0f, 0));
            else TRY(h_update(e->st, e->H, e->Gn, e->Gp, nullptr, e->K, e->n, e->lamH, e->fixH, 1.0f, 1, 0, e->H64));
            }
        } else if (func == 0 && e->K % 64 == 0) {   // K % 64 != 0 would drop the GEMM to its unaligned (general) kernel
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
            if (hug) { TRY(h_update_gram(e->st, e->H, e->GW, e->Gn, 1, 0, e->K, e->n, e->lamH, e->fixH, e->H64)); }
            else if (e->algo == 3) TRY(z_update(e->st, e->Z, e->H, e->Gn, e->Gp, e->Gpvec, e->K, e->nz, e->seg_dev, e->lamH, e->fixH, 1.0f, 0));
            else TRY(h_update(e->st, e->H, e->Gn, e->Gp, nullptr, e->K, e->n, e->lamH, e->fixH, 1.0f, 1, 0, e->H64));
        } else if (e->isplit_h == 1 && e->algo != 3 && !hug && !e->dual2) {
            f.Hio = e->H; f.H64 = e->H64; f.den = kl ? nullptr : e->Gp; f.denvec = kl ? e->Gpvec : nullptr; f.lam = e->lamH; f.fix = e->fixH;
            f.sqrt_rule = e->algo == 2;
            Scope s(e, TAG_FUSED_H);
            TRY(launch_fused(e->st, f, 1, false, func, true, 1));
        } else if (e->dual) {   // split over the rows of W, or constrainednmf, or K > 192: numerator and denominator slabs, then the generic update
            f.out = e->isplit_h == 1 ? e->Gn : e->slabs; f.out2 = e->isplit_h == 1 ? e->Gp : e->slabs2;
            f.slab_stride = (long)e->K * e->n; f.os_r = e->K; f.os_k = 1;
            if (e->dual2 && e->Vhat && e->VT && (!e->Valpha || e->VTa)) {
                // ... as 4 + 2 m*n*K on the transposed copy of V, the W step's scheme with the roles swapped: rows of V' (columns j) stationary, rows of W (the W' copy)
                // streamed, B' = (1./S)' left in the m x n scratch as n x m, then (B'*W)' without forming S again.  out(k, j) straight into the K x n arrays
                FusedParams a;
                memset(&a, 0, sizeof(a));
                a.X = e->H; a.xs_r = e->K; a.xs_k = 1;
                a.Y = e->WT; a.D = e->VTa ? e->VTa : e->VT; a.ldd = e->n; a.R = e->n; a.Cn = e->m; a.K = e->K; a.c_per_split = e->cps_h;
                a.out = f.out; a.slab_stride = f.slab_stride; a.os_r = e->K; a.os_k = 1;
                a.ab_alpha = (float)e->alpha; a.ab_beta = (float)e->beta; a.inv_exp = 1.0f;
                a.Rout = e->Vhat;
                FusedParams b = a;
                b.D = e->Vhat; b.Rout = nullptr; b.out = f.out2;
                Scope s(e, TAG_FUSED_H);
                TRY(launch_fused(e->st, a, e->isplit_h, true, mdiv(e) == NMFX_DIV_IS ? 15 : 16, true, 0));
                TRY(launch_fused(e->st, b, e->isplit_h, true, 0, true, 0));
            } else if (e->dual2) {   // one element map per pass: W'*A, then W'*B
                const int fa = e->dualz ? 17 : (mdiv(e) == NMFX_DIV_IS ? 11 : 13);
                FusedParams g = f;
                g.out = f.out2; g.out2 = nullptr;
                f.out2 = nullptr;
                if (e->dualz) { f.D = e->V; g.D = e->Valpha; }   // numerators from V through S, denominators from V.^(beta-1) without it
                Scope s(e, TAG_FUSED_H);
                TRY(launch_fused(e->st, f, e->isplit_h, false, fa, true, 0));
                TRY(launch_fused(e->st, g, e->isplit_h, false, e->dualz ? 0 : fa + 1, true, 0));
            } else {
                Scope s(e, TAG_FUSED_H);
                TRY(launch_fused(e->st, f, e->isplit_h, false, func, true, 0));
            }
            Scope s(e, TAG_SMALL);
            if (e->isplit_h > 1) {
                TRY(reduce_slabs(e->st, e->slabs, e->isplit_h, f.slab_stride, f.slab_stride, e->Gn, 0));
                TRY(reduce_slabs(e->st, e->slabs2, e->isplit_h, f.slab_stride, f.slab_stride, e->Gp, 0));
            }
            if (e->algo == 3) TRY(z_update(e->st, e->Z, e->H, e->Gn, e->Gp, e->Gpvec, e->K, e->nz, e->seg_dev, e->lamH, e->fixH, outer_exp(e), 0));
            else TRY(h_update(e->st, e->H, e->Gn, e->Gp, nullptr, e->K, e->n, e->lamH, e->fixH, outer_exp(e), 1, 0, e->H64));
        } else {
            f.out = e->isplit_h == 1 ? e->Gn : e->slabs; f.slab_stride = (long)e->K * e->n; f.os_r = e->K; f.os_k = 1;
            {
                Scope s(e, TAG_FUSED_H);
                TRY(launch_fused(e->st, f, e->isplit_h, false, func, true, 0));
            }
            Scope s(e, TAG_SMALL);
            const bool fuse_sum = e->isplit_h > 1 && e->algo != 3;   // h_update sums the slabs on the fly
            if (hug) {
                TRY(h_update_gram(e->st, e->H, e->GW, e->isplit_h == 1 ? e->Gn : e->slabs, e->isplit_h, f.slab_stride, e->K, e->n, e->lamH, e->fixH, e->H64));
            } else {
                if (e->isplit_h > 1 && !fuse_sum) TRY(reduce_slabs(e->st, e->slabs, e->isplit_h, f.slab_stride, f.slab_stride, e->Gn, 0));
                if (fuse_sum) TRY(h_update(e->st, e->H, e->slabs, kl ? nullptr : e->Gp, kl ? e->Gpvec : nullptr, e->K, e->n, e->lamH, e->fixH, e->algo == 2 ? -2.0f : 1.0f,
                                           e->isplit_h, f.slab_stride, e->H64));
                else if (e->algo == 3) TRY(z_update(e->st, e->Z, e->H, e->Gn, kl ? nullptr : e->Gp, e->Gpvec, e->K, e->nz, e->seg_dev, e->lamH, e->fixH, 1.0f, 0));
                else TRY(h_update(e->st, e->H, e->Gn, kl ? nullptr : e->Gp, kl ? e->Gpvec : nullptr, e->K, e->n, e->lamH, e->fixH, e->algo == 2 ? -2.0f : 1.0f, 1, 0, e->H64));
            }
        }
        e->cost_valid = false;
        return NMFX_OK;
    }
    if (!e->all_fixH) {
        OpView a{}, b{};
        num_view(e, a);
        bool fuse_hupd = false;
        if (e->lagram) TRY(ensure_hpad(e));   // the denominator below reads the padded copy of the CURRENT H
        if (e->fusedT_dual) {   // both element maps' values with the W just updated
            TRY(fusedT_pass(e, FT_S_DUAL, nullptr));
            a = OpView{e->Vhat, nullptr, e->m, VIEW_KC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f};
        }
        if (e->fusedT_kl) {   // R = V./V_hat with the W just updated
            TRY(fusedT_pass(e, FT_S_KL, nullptr));
            a = OpView{e->Vhat, nullptr, e->m, VIEW_KC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f};
        }
        if (e->dualw) {                     // both element maps' values with the W just updated; W'*A and W'*B below are plain two-operand products (K x n x m)
            TRY(klw_s_pass(e, true, false));
            a = OpView{e->Vhat, nullptr, e->m, VIEW_KC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f};
        }
        if (e->klw && !e->klw_vt) {         // likewise
            TRY(klw_s_pass(e, true, false));
            a = OpView{e->Vhat, nullptr, e->m, VIEW_KC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f};
        }
        e->hpad_valid = false;
        if (e->klw_vt) TRY(klw_hstep_vt(e));
        else if (e->eucw) TRY(eucw_hnum(e));
        else if (e->qgemm) {
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
            // euclidean Gram path, one GPU: the shift-sum over t rides in the H update below (h_update_shift), which also rewrites the padded copy of H
            fuse_hupd = e->gram && e->algo == 1 && mdiv(e) == NMFX_DIV_EUCLIDEAN && outer_exp(e) == 1.0f && e->hL == 0 && e->hR == 0 && e->Hpad && !e->fusedT_kl;
            if (!fuse_hupd) {
                Scope s(e, TAG_SMALL);
                TRY(shift_sum(e->st, e->Qbuf, e->K, e->T, e->n, e->nvalid, e->Gn));
            }
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
        } else if (e->fusedT_dual) {   // sum_t W_t' * lshift_t(B) as Q = W_flat' * B + a shift-sum, like the numerator above
            {
                Scope s(e, TAG_HDEN);
                GemmParams g;
                memset(&g, 0, sizeof(g));
                g.M = e->KT; g.N = e->nvalid; g.Kc = e->m;
                g.A = OpView{e->W, nullptr, e->m, VIEW_KC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f};
                g.B = OpView{e->Vhat2, nullptr, e->m, VIEW_KC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f};
                g.C = e->Qbuf; g.ldc = e->KT; g.epi = EPI_STORE; g.splitk = 1;
                TRY(gemm_auto(e->st, g, e->gemm_scratch, e->gemm_scratch_bytes));
            }
            Scope s(e, TAG_SMALL);
            TRY(shift_sum(e->st, e->Qbuf, e->K, e->T, e->n, e->nvalid, e->Gp));
        } else if (div_has_matrix_den(e->div)) {
            den_view(e, b);
            if (e->dualw) b = OpView{e->Vhat2, nullptr, e->m, VIEW_KC, 0, 0, 0, NMFX_PRO_NONE, 0.f, 0.f};
            TRY(wt_times_x(e, b, e->Gp, TAG_HDEN));
        }
        Scope s(e, TAG_SMALL);
        if (!div_has_matrix_den(e->div)) {
            TRY(col_reduce(e->st, e->W, e->m, e->m, e->KT, 0, e->colsum));
            TRY(sum_over_t(e->st, e->colsum, e->K, e->T, e->Gpvec));
        }
        if (e->algo == 3) TRY(z_update(e->st, e->Z, e->H, e->Gn, e->Gp, e->Gpvec, e->K, e->nz, e->seg_dev, e->lamH, e->fixH, outer_exp(e), 0));
        else if (fuse_hupd) {
            TRY(h_update_shift(e->st, e->H, e->Qbuf, e->Gp, e->K, e->T, e->n, e->nvalid, e->lamH, e->fixH, e->Hpad, e->T - 1, e->lagram ? e->T - 1 : 0, e->H64));
            e->hpad_valid = true;   // (the layout ensure_hpad would produce)
        } else TRY(h_update(e->st, e->H, e->Gn, e->Gp, div_has_matrix_den(e->div) ? nullptr : e->Gpvec, e->K, e->n, e->lamH, e->fixH, e->algo == 2 ? -2.0f : outer_exp(e), 1, 0, e->H64));
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
    if (e->fusedT_kl || e->fusedT_dual || e->klw || e->dualw) { e->cost_valid = false; return NMFX_OK; }   // the cost lags: by-product of the next S pass, or nmfx_engine_cost_pass
    if ((e->fusedT || e->eucw) && e->gram_cost) { e->cost_valid = false; return NMFX_OK; }   // the cost lags: Gram form out of the next W update, or nmfx_engine_cost_pass
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
    if (e->fusedT_dual) {
        TRY(fusedT_pass(e, FT_COST_DUAL, nullptr));
        Scope s(e, TAG_SMALL);
        e->cost_valid = true;
        return cost_from_partials(e, e->n_cost_used);
    }
    if (e->klw) { TRY(klw_s_pass(e, false, true)); return klw_cost(e); }
    if (e->dualw) {
        TRY(klw_s_pass(e, false, true));
        Scope s(e, TAG_SMALL);
        e->cost_valid = true;
        return cost_from_partials(e, e->n_cost_used);
    }
    if ((e->fusedT || e->eucw) && e->gram_cost) {
        if (e->eucw) TRY(eucw_cost_pass(e, nullptr));
        else
        TRY(fusedT_pass(e, FT_COST_EUC, nullptr));
        Scope s(e, TAG_SMALL);
        e->cost_valid = true;
        return cost_from_partials(e, e->n_cost_used);
    }
    set_error("nmfx_engine_cost_pass: no cost available yet (call hstep first)");
    return NMFX_ERR_INVALID;
}
int32_t nmfx_engine_is_fused(nmfx_engine *e) { return e->fused ? 1 : ((e->fusedT_kl || e->fusedT_dual) ? 4 : ((e->klw || e->dualw) ? 5 : (e->fusedT ? 3 : (e->eucw ? 6 : (e->gram ? 2 : 0))))); }   // 6 euclidean with K > 256 in column blocks, 1 fused kernels, 3 fused cnmf passes + Gram denominators, 4 KL cnmf on the fused passes, 5 KL with K > 256 in column blocks, 2 Gram form on the GEMM, 0 materialised V_hat

nmfx_status nmfx_engine_cost_ptr(nmfx_engine *e, double **dev_cost) { *dev_cost = e->cost; return NMFX_OK; }
nmfx_status nmfx_engine_copy_cost(nmfx_engine *e, double *dst_dev) {
    NMFX_HIP(hipMemcpyAsync(dst_dev, e->cost, sizeof(double), hipMemcpyDeviceToDevice, e->st));
    return NMFX_OK;
}

// the whole stretch between two all-reduces of a column-sharded run as ONE call: replicated W update, local H step, and (unless
// `last`) the next iteration's W-step partial.  Not for cnmf shards, whose H step is split around the halo exchange.
nmfx_status nmfx_engine_between_allreduces_cost(nmfx_engine *e, int32_t last, double *lag2_cost_dst_dev) {
    if (e->hL || e->hR) { set_error("nmfx_engine_between_allreduces: not for shards with halos"); return NMFX_ERR_UNSUPPORTED; }
    TRY(nmfx_engine_wstep_finish(e));
    // engines of cost lag 2: the cost of the PREVIOUS iteration is in e->cost exactly here -- written by this wstep_finish (Gram form) or, once the engine has
    // latched to the one-pass kernel, by the wstep_partial before it -- and the next wstep_partial below may overwrite it: hand it out now
    if (lag2_cost_dst_dev) NMFX_HIP(hipMemcpyAsync(lag2_cost_dst_dev, e->cost, sizeof(double), hipMemcpyDeviceToDevice, e->st));
    TRY(nmfx_engine_hstep(e));
    if (!last) TRY(nmfx_engine_wstep_partial(e));
    return NMFX_OK;
}
nmfx_status nmfx_engine_between_allreduces(nmfx_engine *e, int32_t last) { return nmfx_engine_between_allreduces_cost(e, last, nullptr); }

nmfx_status nmfx_engine_iterate(nmfx_engine *e, int32_t iters, double *dev_cost_out) {
    const bool lag = nmfx_engine_cost_lag(e) != 0;   // the cost of iteration i is a by-product of iteration i+1's W step
    for (int it = 0; it < iters; ++it) {
        // fused path: the W-step pass also produces the cost of the state it starts from, i.e. of iteration it-1; its finisher writes it
        // straight into the caller's vector (no separate 8-byte copy)
        e->cost_dst2 = (lag && it > 0 && dev_cost_out) ? dev_cost_out + it - 1 : nullptr;
        nmfx_status ws_ = nmfx_engine_wstep_partial(e);
        if (!e->wstep_gram) e->cost_dst2 = nullptr;   // Gram-form cost: the finisher runs inside wstep_finish
        TRY(ws_);
        ws_ = nmfx_engine_wstep_finish(e);
        e->cost_dst2 = nullptr;
        TRY(ws_);
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
    e->prof.skip_mask = enable == 2 ? ((1u << TAG_SMALL) | (1u << TAG_GRAM)) : 0u;   // level 2: the big passes only
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
        if (e->fusedT || e->fusedT_kl || e->fusedT_dual || e->klw || e->eucw) { *flops = f; *bytes = 4.0 * (m * n + m * KT + e->K * n); return NMFX_OK; }   // cnmf numerator pass: one contraction (klw: the launches of all column blocks together)
        if (e->dualz) { *flops = 1.5 * f; *bytes = 4.0 * (m * n + 2.0 * m * KT + e->K * n); return NMFX_OK; }   // alpha == 0: functor 17 (S + one contraction) and functor 0 (one contraction), averaged over the two launches
        if (e->dual2 && e->Vhat) { *flops = 1.5 * f; *bytes = 4.0 * (2.0 * m * n + 2.0 * m * KT + e->K * n); return NMFX_OK; }   // per launch, averaged over the two of a W step: S + one contraction (+ the m x n store), then one contraction
        if (e->dual2) { *flops = 2.0 * f; *bytes = 4.0 * (m * n + 2.0 * m * KT + e->K * n); return NMFX_OK; }   // per launch (two per W step): S + one contraction
        if (e->dual) { *flops = 3.0 * f; *bytes = 4.0 * (m * n + 3.0 * m * KT + e->K * n); return NMFX_OK; }   // S + two contractions
        if (e->wstep_gram) { *flops = f; *bytes = 4.0 * (m * n + m * KT + e->K * n); return NMFX_OK; }   // numerators only: one contraction
        *flops = 2.0 * f / ch; *bytes = 4.0 * (m * n / ch + 2.0 * m * KT / ch + e->K * n); return NMFX_OK;
    }
    case TAG_FUSED_H: *flops = (e->dualz ? 3.0 : (e->dual2 && e->Vhat && e->VT) ? 3.0 : e->dual2 ? 4.0 : e->dual ? 3.0 : (mdiv(e) == NMFX_DIV_KL ? 2.0 : 1.0)) * f; *bytes = 4.0 * (m * n + m * KT + 2.0 * e->K * n); return NMFX_OK;
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

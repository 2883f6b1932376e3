// Internal declarations shared by the HIP translation units of libnmfx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "nmfx.h"

#define NMFX_EPS_F 2.220446049250313e-16f /* MATLAB eps = 2^-52, representable in fp32 */

namespace nmfx {

void set_error(const char *fmt, ...);
#define NMFX_HIP(call)                                                                    \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            nmfx::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return NMFX_ERR_HIP;                                                          \
        }                                                                                 \
    } while (0)

#define TRY(x) do { nmfx_status s_ = (x); if (s_ != NMFX_OK) return s_; } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel AND device: the attribute belongs to the device's copy of the code object, and one process may drive
// several GPUs (the blocking multi-GPU call).  One static instance per launcher instantiation; a lost race sets it twice; a larger request than the last one sets it again.
struct LdsAttrOnce {
    int set_bytes[64] = {};   // per device: the largest size asked for so far
    nmfx_status set(const void *fn, int bytes) {
        int dev = 0;
        NMFX_HIP(hipGetDevice(&dev));
        const bool tracked = dev >= 0 && dev < 64;
        if (tracked && set_bytes[dev] >= bytes) return NMFX_OK;
        NMFX_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        if (tracked) set_bytes[dev] = bytes;
        return NMFX_OK;
    }
};

// Declared AFTER the host vectors / device buffers an entry point copies to or from asynchronously: whichever way the function is left -- an NMFX_HIP / TRY
// early return included -- the stream is drained before that memory goes away, so no DMA is ever in flight into (or out of) a dead std::vector or a freed buffer
struct StreamDrain {
    hipStream_t st;
    explicit StreamDrain(hipStream_t s) : st(s) {}
    ~StreamDrain() { if (hipStreamSynchronize(st) != hipSuccess) (void)hipGetLastError(); }
};

// Collectives are the CALLER's: libnmfx never links RCCL.  A Comm wraps the all-reduce callback handed to nmfx_nmfsc_dev
// (torch.distributed in the Python driver, ncclAllReduce in a MEX shim); inactive on one GPU.
struct Comm {
    nmfx_allreduce_fn fn = nullptr;
    void *ctx = nullptr;
    hipStream_t st = nullptr;
    bool active() const { return fn != nullptr; }
    nmfx_status allreduce(void *dev_ptr, long count, int dtype, int op) const {
        if (!fn) return NMFX_OK;
        if (fn(ctx, dev_ptr, (int64_t)count, dtype, op, (void *)st) != 0) { set_error("all-reduce callback failed"); return NMFX_ERR_INVALID; }
        return NMFX_OK;
    }
};

// ---- operand views of the general MFMA GEMM (gemm.hip) ------------------------------------
// An operand element is addressed by (r, kc): r = its non-contracted index (row of op(A) /
// column of op(B)), kc = the contraction index.  RC = memory-contiguous along r, KC = along kc.
enum ViewMode {
    VIEW_RC = 0,         // p[r + ld*kc]
    VIEW_KC = 1,         // p[kc + ld*r]
    VIEW_HSTACK_KC = 2,  // kc=(t,k): H[k + K*(r - t)],  r>=t      : B of  V_hat = W_flat * H_stack   (RFD.m:36-38)
    VIEW_HSTACK_RC = 3,  // r=(t,k):  H[k + K*(kc - t)], kc>=t     : B of  N_all = X * H_stack'       (cnmf.m:191)
    VIEW_WSTACK_KC = 4,  // kc=(t,i): W[i + m*r + m*K*t]           : A of  G = sum_t W_t' * lshift(X) (cnmf.m:225)
    VIEW_XSHIFT_KC = 5   // kc=(t,i): X[i + m*(r + t)], r+t<lim    : B of the same product            (cnmf.m:219)
};

struct OpView {
    const float *p;   // primary input
    const float *p2;  // second input of the prologue (same addressing) or nullptr
    long ld;          // leading dimension (elements)
    int mode;         // ViewMode
    int blk;          // length of the inner index of a stacked (t, inner) index
    long tstride;     // VIEW_WSTACK_KC: m*K;  VIEW_HSTACK_KC: if > 0, rows r >= tstride read as zero
    int lim;          // VIEW_XSHIFT_KC: n;  VIEW_HSTACK_RC: row offset (select stacked block t: lim = t*blk);  VIEW_HSTACK_KC: extra column shift g (element valid iff r + g >= t)
    int func;         // nmfx_prologue
    float e1, e2;     // NMFX_PRO_POWPROD exponents (MATLAB .^ semantics: x.^0 == 1, x.^1 == x)
    int goff;         // HSTACK views: columns j >= -goff exist (left halo of a column shard); element valid iff j - t >= -goff
    long safe;        // element offset from p that is ALWAYS readable: the pipelined kernel loads it for chunks outside the view (and zeroes them).
                      // 0 unless p itself lies before the allocation (a view shifted by pointer arithmetic, cnmfsc's V_hat correction)
};

enum EpiMode {
    EPI_STORE = 0,       // C = acc  (or C += acc)
    EPI_COST = 1,        // per-block fp64 partial of the divergence between Vref and acc (+ optional store of acc)
};

struct GemmParams {
    OpView A, B;
    long M, N, Kc;
    float *C;
    long ldc;
    int accumulate;      // C += acc
    int clamp0;          // EPI_STORE: C = max(C, 0) after the accumulate
    int store_c;         // EPI_COST: also store acc to C
    int epi;             // EpiMode
    int cost_div;        // nmfx_divergence for EPI_COST
    float cost_alpha, cost_beta;   // NMFX_DIV_AB (nmf.m:214)
    const float *Vref;   // EPI_COST: reference matrix (M x N, ld = ldv)
    long ldv;
    double *cost_partials;  // EPI_COST: one fp64 partial per block [gridDim.x*gridDim.y]
    long cost_ncols;     // EPI_COST: only columns j < cost_ncols enter the cost (0 = all): halo columns of a shard are not this rank's
    int splitk;          // >1: partial sums written to slabs C + z*slab_stride, reduced by reduce_slabs
    long slab_stride;
    long kc_per_split;   // multiple of BK
    // z-batched launch (pipelined kernel only): blockIdx.z = t runs the SAME contraction on shifted operands and writes slab t,
    //   A.p += t*zA_off, B.p += t*zB_off, B.lim += t*zB_lim, B.tstride += t*zB_tstride      (cnmf Gram H step: one launch for all t)
    int zbatch;
    long zA_off, zB_off;
    int zB_lim, zB_tstride;
};
bool gemm_pipe_eligible(const GemmParams &p);   // would launch_gemm run the pipelined kernel on p? (required for zbatch)

nmfx_status launch_gemm(hipStream_t st, const GemmParams &p, long *blocks_out = nullptr);
// picks split-K, runs the GEMM and the deterministic slab reduction; scratch >= gemm_scratch_bytes(M,N,Kc)
nmfx_status gemm_auto(hipStream_t st, GemmParams p, void *scratch, size_t scratch_bytes);
int gemm_pick_split(long M, long N, long Kc);
long gemm_grid_blocks(long M, long N);   // upper bound of the (x,y) blocks (== cost partials) of an EPI_COST launch
size_t gemm_scratch_bytes(long M, long N, long Kc);

// ---- fused two-stage kernel (fused.hip) -------------------------------------------------------------
// Build-time switches (A/B builds: nmf_toolbox_amd/build.py --variant; the defaults are what ships).
// NMFX_KL_MODE: the KL element map of functors 3 / 8 (R = V./S with the cost terms, nmf.m:152,210)
//   0  round 1-5: q = V*rcp(S), the kernel sums V.*log(q) only and the caller adds sum(S) - sum(V) in closed form from the factors.  Two first-order errors reach the
//      cost that way: the bias of the hardware reciprocal (-2e-8 relative in q: -2e-8*sum(V)) and the rounding of the fp32 S the log sees against the EXACT sum(S) of
//      the closed form.  Both are of the size of sum(V), so a well-fitting factorisation (cost << sum(V)) showed them as 1e-6 ... 2e-6 of its cost
//   1  the divergence term of every element is formed from ONE S and ONE q:  V.*log(q) + (S - q.*S)  [= V.*log(V./S) - V + S with V replaced by q.*S, which it
//      equals up to the rounding of q].  d/dq of that expression vanishes at q = V/S, so the error of q enters squared, and the S the log sees is the S that is
//      summed.  One fma + one add more per element (NU = 6)
//   2  the same on PACKED fp32 VALU instructions (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 on register pairs; rcp and log stay scalar): 8 instructions per
//      PAIR of elements -- the instruction count of mode 0 with the terms of mode 1.  Functor 2 (no cost) drops to 3 per pair
#ifndef NMFX_KL_MODE
#define NMFX_KL_MODE 2
#endif
// NMFX_G2_VEC: the streamed operand of the SECOND product read from LDS as b128 / b64 / b96 instead of K/32 ds_read_b32 per step.  The k index an MFMA's
//   A-row carries is free (it only names the output row the epilogue writes), so lane i takes the 4 CONSECUTIVE floats 4*i .. 4*i+3 of a 128-float chunk for
//   four MFMAs instead of floats i, 32+i, 64+i, 96+i: out(k = 128*c + 4*i + j, r) <-> acc[4*c + j].  No change to the LDS layout or to the first product
#ifndef NMFX_G2_VEC
#define NMFX_G2_VEC 1
#endif
// NMFX_G1_ASM: the first product's MFMAs as inline asm with the S tile in VGPRs and the stationary operand in AGPRs (fused_kernel.h)
#ifndef NMFX_G1_ASM
#define NMFX_G1_ASM 1
#endif
constexpr bool KL_CONSISTENT_COST = NMFX_KL_MODE != 0;   // engine.hip: the finishers add the closed form only when the kernel does not sum it
struct FusedParams {
    const float *X;       // stationary factor: X(r, k) = X[r*xs_r + k*xs_k]   (W step: W; H step: H^T i.e. H with xs_r = K, xs_k = 1)
    long xs_r, xs_k;
    int T;                // cnmf (W-step form only): K = Kh*T contraction indices (t, k); X / out slice t at xs_t / os_t; Y = H (Kh x n),
    long xs_t, os_t;      //   preceded by T-1 readable columns (zeros or a shard's left halo).  0 or 1: plain nmf
    const float *Y;       // streamed factor: row c = K contiguous floats at Y + c*y_stride  (W step: columns of H; H step: rows of W = W^T copy)
    long y_stride;        // 0 = K.  > K: the K floats are a column block of longer rows (H-step numerator of a wide factor in blocks of <= 256)
    int nz;               // grid.z (0 / 1: none): block z streams the K floats at Y + z*yz_stride of every row and writes at out + z*oz_stride (T <= 1 only)
    long yz_stride, oz_stride;
    const float *D;       // V (m x n, ld = ldd); W step reads V(r, c) = D[r + ldd*c], H step V(c, r) = D[c + ldd*r]
    long ldd;
    long R, Cn;           // stationary rows (multiple of 128), streamed rows (multiple of 64)
    long c_per_split;     // streamed rows per grid.y slice (multiple of 64)
    int K;
    float *out;           // EPI 0: O(k, r) -> out[split*slab_stride + r*os_r + k*os_k]
    float *Rout2;         // func 11 / 13, cost-only pass (cnmf, IS / alpha-beta): the SECOND element map's values (1./S, S.^(a+b-1)) next to the first one's in Rout
    float *Rout;          // func 2 / 3, cost-only pass, W-step form: also store R = V./S (m x n, ld = ldd); nullptr = don't.  func 7: the partial S; func 8: R
    const float *Sin;     // func 7 / 8: partial S (m x n, ld = ldd) of the column blocks contracted by earlier launches, or nullptr (first block)
    float *out2;          // func 4 / 5 (dual-map divergences), EPI 0: the second contraction (denominators), same indexing
    float ab_alpha, ab_beta;   // func 5
    float inv_exp;        // func 4 / 5, EPI 1: outer exponent 1/alpha of nmf.m:193-194 (1 = none); set it to 1 for func 4
    long slab_stride, os_r, os_k;
    double *cost_partials;  // [gridDim.x*gridDim.y] or nullptr.  Euclidean: sum (V-S)^2.  KL: the complete sum(V.*log(V./S) - V + S) of the workgroup's
                            // elements, every term from the same fp32 S (NMFX_KL_MODE above; mode 0: sum V.*log(V./S) only, the caller adds the rest in closed form)
    float *Hio;           // EPI 1: H updated in place
    const float *den;     // EPI 1: K x n denominator matrix, or nullptr -> denvec
    const double *denvec; // EPI 1: [K]
    const float *lam;     // [K] or nullptr
    const uint8_t *fix;   // [K] or nullptr
    int sqrt_rule;        // EPI 1: H <- sqrt(H .* G)   (lnmf.m:76) instead of the ratio update
    double *H64;          // EPI 1: float64 master copy of H (K x n) or nullptr: the epilogue reads it, updates in double and writes both arrays
    const int *run_if;    // when set: every workgroup returns at once unless *run_if != 0 (a device-side decision, no host round trip)
};
bool fused_supported(int K);
bool fused_supported_T_dual(int Kh, int T);   // ... of the IS / alpha-beta S pass (both element maps stored)
bool fused_supported_T(int Kh, int T);   // cnmf: instantiated (Kh, T) pairs of the W-step-form kernels (numerator pass, cost pass)
// func: 0 R=V (no S) | 1 R=V + euclidean cost from S | 2 R=V./S (KL) | 3 R=V./S + KL cost | 4 IS | 5 alpha-beta (K <= 192) | 6 R=S-V + euclidean cost (do_g2, slabs out)
//       | 7 / 8 (do_g2=false): S over column blocks of a wide factor, see fused_kernel.h;  do_g2=false: cost-only pass
nmfx_status launch_fused(hipStream_t st, const FusedParams &p, int nsplit, bool d_rc, int func, bool do_g2, int epi);

// ---- small products of the Gram form (small_mm.hip) ----------------------------------------------------------------------
bool h_update_gram_supported(int K);
nmfx_status h_update_gram(hipStream_t st, float *H, const float *G, const float *Gn, int n_slabs, long slab_stride, int K, long n, const float *lam,
                          const uint8_t *fix, double *H64 = nullptr);   // H64: the float64 master of H (K x n), see h_update

// ---- small kernels (aux.hip) ----------------------------------------------------------------
nmfx_status reduce_slabs(hipStream_t st, const float *slabs, int nslab, long slab_stride, long count, float *out,
                         int accumulate);
// column reductions in fp64: out[c] = sum_i f(X[i + ld*c]); mode 0 sum, 1 sum of squares, 2 sum of |x|
nmfx_status col_reduce(hipStream_t st, const float *X, long rows, long ld, int ncols, int mode, double *out);
// row reductions: out[k] = sum_j f(X[k + ld*j]); scratch >= row_reduce_scratch_bytes(rows)
nmfx_status row_reduce(hipStream_t st, const float *X, int rows, long ld, long ncols, int mode, double *out, void *scratch);
size_t row_reduce_scratch_bytes(int rows);
// W update of one W step for all (k,t) columns: nmf.m:168 / cnmf.m:193 with the diag terms as column sums
struct WUpdateParams {
    float *W;            // m x (K*T)
    const float *N;      // m x (K*T) numerator GEMM result
    const float *P;      // m x (K*T) denominator GEMM result, or nullptr when Pvec is used
    const double *Pvec;  // [K*T] broadcast denominator (KL: rowsum of (shifted) H), or nullptr
    const float *Pvecf;  // the same as fp32 (the all-reduced tail of `packed`), used when Pvec is nullptr
    const float *lamW;   // [K] device or nullptr
    const uint8_t *fixW; // [K] device or nullptr
    long m;
    int K, T;
    double *sumsq;       // out [K*T]: sum of squares of the updated (un-normalised) columns
    float inv_exp;       // outer exponent 1/alpha (1 = none)
    int rule;            // 0: nmf/cnmf (diag terms, sum of squares out); 1: lnmf (plain ratio, column sum out)
    int n_chunks;        // <= 1: N is m x K column-major.  c > 1: N = c contiguous (m/c x K) row blocks (row-chunked W-step partial)
    int fuse_norm;       // 0: leave the columns un-normalised (w_normalize follows).  1 / 2 (T == 1 only): also apply nmf.m:169 (L2) / lnmf.m:70 (L1)
                         // to the column in a third sweep of the same workgroup -- one launch and one pass over W fewer
    double *colsum_out;  // with fuse_norm != 0: [K] column sums of the FINAL W, fixed columns included (KL H-step denominator, nmf.m:184), or nullptr
    double *dndp;        // [2*K*T] or nullptr: dn[c] = sum_i W.*P and dp[c] = sum_i W.*N of column c (fixed columns included).  stats_only: the kernel
    int stats_only;      // writes them and returns (w_stats); stats_in: it reads them instead of summing (the update half of a split W update)
    int stats_in;
    // Gram-form cost (see gram_cost_finish): when fin_on, the LAST workgroup of the launch first finishes the cost of the state this W step started from --
    // the statistics, the flag and the (conditional) residual partials are all complete by then -- and only then updates its own column: one launch less
    int fin_on, fin_nc, fin_rank0, fin_nparts, fin_nW, fin_K;
    const double *fin_sumVV, *fin_partials, *fin_l1W, *fin_l1H;
    const int *fin_exact_flag;
    const float *fin_lamW, *fin_lamH;
    double *fin_out, *fin_out2;
    // float64 master copy of W (m x K*T, or nullptr): the column is read from it, updated in double and written to BOTH arrays (nmf.m:168-169 run in
    // double; W is what the MFMA passes contract).  P64: the denominator product in float64 (gemm64), instead of P
    double *W64;
    const double *P64;
};
nmfx_status w_update(hipStream_t st, const WUpdateParams &p);
// Euclidean cost in Gram form, from the column statistics of the W update:  0.5*||V - W*H||^2 = 0.5*sumVV - sum(dp) + 0.5*sum(dn).
// gram_decide: sets *exact_flag (sticky) once that value drops below ratio_min * 0.5*sumVV[1] -- below it fp32 products no longer resolve the cost to the
// contract and the explicit residual pass takes over (launched with run_if = exact_flag).  The state of the flag after THIS decision goes to the host-mapped
// *host_slot as (stamp << 1) | flag (stamp > 0 numbers the decisions of an engine; the host reads the slot of a fixed earlier decision, engine.hip::gram_active).
nmfx_status gram_decide(hipStream_t st, const double *dndp, int nc, const double *sumVV, double ratio_min, int *exact_flag, int *host_slot, int stamp);
// cost = (*exact_flag ? 0.5*sum(partials) : 0.5*sumVV[0] + (rank0 ? 0.5*sum(dn) - sum(dp) : 0)) + lambda terms     -> out, out2
nmfx_status gram_cost_finish(hipStream_t st, const double *dndp, int nc, const double *sumVV, int rank0, const int *exact_flag, const double *partials, int nparts,
                             const double *l1W, int nW, const float *lamW, const double *l1H, int K, const float *lamH, double *out, double *out2);
nmfx_status w_normalize(hipStream_t st, float *W, long m, int K, int T, const double *sumsq, const uint8_t *fix, int cnmf_rule,
                        double *f_out, int kvalid = 0, double *W64 = nullptr);   // W64: the float64 master (read, scaled in double, both arrays written)
// C (M x N) = A * B accumulated in float64 on the fp64 matrix core (gemm64.hip); A(i, k) = A[i + lda*k], B(k, j) = B[k + ldb*j], each fp32 or float64
nmfx_status gemm64(hipStream_t st, long M, long N, long Kc, const double *A64, const float *A32, long lda, const double *B64, const float *B32, long ldb,
                   double *C64, float *C32, long ldc);
constexpr int NMFX_MAX_GPUS = 16;
struct PeerPtrs { float *p[NMFX_MAX_GPUS]; };
nmfx_status peer_reduce(hipStream_t st, const PeerPtrs &bufs, int ndev, int self, long off, long count);
nmfx_status shift_sum(hipStream_t st, const float *Q, int K, int T, long n, long nvalid, float *Gn);
nmfx_status pad_left(hipStream_t st, const float *src, int K, long n, int pad, float *dst, int pad_right = 0);
// cnmf Gram products by lag (aux.hip): KT x KT Gram of the stacked H from the T lag Grams; lag sums of CC; exact last T-1 columns of the H-step denominator
nmfx_status gram_from_lags(hipStream_t st, const float *L, const float *H, int K, int T, long n, float *G);
nmfx_status lag_sum(hipStream_t st, const float *CC, int K, int T, float *E);
nmfx_status gp_tail(hipStream_t st, const float *CC, const float *H, int K, int T, long n, float *Gp);
nmfx_status repack_rows(hipStream_t st, const float *src, int rs, float *dst, int rd, long cols);
nmfx_status scale_rows(hipStream_t st, float *H, int K, long n, const double *s);
nmfx_status scale_rows64(hipStream_t st, double *H64, float *H, int K, long n, const double *s);   // the float64 master and its fp32 image
nmfx_status repack_rows64(hipStream_t st, const double *src, int rs, double *dst, int rd, long cols);
nmfx_status col_reduce64(hipStream_t st, const double *X, long rows, long ld, int ncols, int mode, double *out);
nmfx_status cvt_f64_to_f32(hipStream_t st, const double *in, float *out, long count);
nmfx_status scale_cols(hipStream_t st, float *X, long rows, int ncols, const double *s, int use_sqrt, int divide);
// cnmf, euclidean Gram path: Gn(k, j) = sum_t Q((t,k), j + t) (cnmf.m:217-226, the shift-sum of the Q product), the update of cnmf.m:231 and the
// zero-padded copy Hpad = [padL zero columns | H | padR zero columns] the next passes stream -- shift_sum + h_update + pad_left in ONE launch
nmfx_status h_update_shift(hipStream_t st, float *H, const float *Q, const float *Gp, int K, int T, long n, long nvalid, const float *lamH, const uint8_t *fixH,
                           float *Hpad, int padL, int padR, double *H64 = nullptr);
// H64 (or nullptr): the float64 master copy of H -- the update reads it, runs nmf.m:199 in double and writes both arrays
nmfx_status h_update(hipStream_t st, float *H, const float *Gn, const float *Gp, const double *Gpvec, int K, long n,
                     const float *lamH, const uint8_t *fixH, float inv_exp, int n_slabs = 1, long slab_stride = 0, double *H64 = nullptr);   // n_slabs > 1: Gn = sum of slabs
nmfx_status z_update(hipStream_t st, float *Z, float *H, const float *Gn, const float *Gp, const double *Gpvec, int K, long nz, const long *seg,
                     const float *lamZ, const uint8_t *fixZ, float inv_exp, int gather_only);
nmfx_status center_of_gravity(hipStream_t st, const void *W, int is_f64, long m, int K, int *cog);
nmfx_status permute(hipStream_t st, const void *in, void *out, int is_f64, long rows, long cols, const int *order, int by_rows);
nmfx_status finish_cost(hipStream_t st, const double *partials, int count, double scale, const double *l1W, int nW, const float *lamW,
                        const double *l1H, int K, const float *lamH, double *out, const double *dotA = nullptr, const double *dotB = nullptr,
                        int ndot = 0, const double *minus = nullptr,    // + sum_k dotA[k]*dotB[k] - *minus
                        const double *pre_c = nullptr, double pre_a = 0.0, double pre_b = 0.0,    // scale * (sum(partials) + pre_a * *pre_c + pre_b)
                        double *out2 = nullptr,                                                   // second destination of the cost (the caller's cost vector)
                        const double *cvt_src = nullptr, float *cvt_dst = nullptr, int ncvt = 0);  // + cvt_dst[i] = (float)cvt_src[i]  (rowsum(H) into the tail of `packed`)
// nmfsc line searches: partials[b] = 2*<grad, Xc - X> + sum_i (Xc - X)(i,:) * G * (Xc - X)(i,:)' over the rows of block b (aux.hip)
int quad_rows_blocks(long R, int K);
// the same on the K x n layout (columns are the K-vectors; small_mm.hip, G*D on the MFMA): partials[b] over 32 columns
int dot_2a_b_blocks(long count);
nmfx_status dot_2a_b(hipStream_t st, const float *d, const float *a, const float *b, long count, double *partials);   // partials[block] = sum d .* (2a + b)
int quad_cols_blocks(long n);
nmfx_status quad_cols(hipStream_t st, const float *H, const float *Hc, const float *grad, const float *G, int K, long n, double *partials);
bool quad_rows_supported(int K);
nmfx_status quad_rows(hipStream_t st, const float *X, const float *Xc, const float *grad, const float *G, long R, int K, double *partials);
nmfx_status publish_obj(hipStream_t st, const double *partials, int count, double scale, const double *src, double *out, double *slot, unsigned long long seq);
nmfx_status col_reduce_pow(hipStream_t st, const float *X, long rows, long ld, int ncols, float e, double *out);
nmfx_status pow_map(hipStream_t st, const float *in, float *out, long count, float e);
nmfx_status sum_vec(hipStream_t st, const double *v, long count, double *out);
// nmfsc with K <= smallk_max(): residual-form gradients and objective in fp64 (aux.hip)
// out (m x K doubles) = (Vh - V) * rshift_t(H)' in fp64; slabs: nch * m * K doubles (nch column chunks)
// R64 (m x n float64, or nullptr): the residual itself, used instead of Vh - V (recon_resid64 forms it: sum_{t < Tn} W_t * rshift_t(H) - V in float64)
nmfx_status recon_resid64(hipStream_t st, const float *V, const float *W, long m, long n, int K, int Tn, const float *H, double *R64);
nmfx_status resid_xht64(hipStream_t st, const float *V, const float *Vh, const double *R64, long m, long n, const float *H, int K, int t, double *slabs, int nch, double *out);
// outT (n x K doubles) = (sum_t W_t' * lshift_t(Vh - V))' in fp64
nmfx_status resid_hgrad64(hipStream_t st, const float *V, const float *Vh, const double *R64, long m, long n, const float *W, int K, int T, double *outT);
// nmfsc on small problems, any K: R64 = W*H - V (m x n doubles, or nullptr: objective only) + sum of squares per workgroup; then dH' = (W'*R64)' and dW = R64*H'
int resid64_blocks(long m, long n);
nmfx_status resid64(hipStream_t st, const float *V, long m, long n, const float *W, const float *H, int K, int ldh, double *R64, double *partials, int *nparts);
nmfx_status r64_wt(hipStream_t st, const double *R64, long m, long n, const float *W, int K, double *outT);
nmfx_status r64_ht(hipStream_t st, const double *R64, long m, long n, const float *H, int K, int ldh, double *slabs, int nch, double *out);
int smallk_max();
int smallk_dw_chunks(long m, long n);
int smallk_partials(long m, long n);   // upper bound of *nparts
nmfx_status smallk_grad(hipStream_t st, int Kv, const float *V, long m, long n, const float *W, const float *H, int ldh, double *dHT, double *dW, double *slabs,
                        double *partials, int *nparts);
nmfx_status fill_f32(hipStream_t st, float *p, long count, float v);
nmfx_status axpy_f32(hipStream_t st, long count, float a, const float *x, const float *y, float *out);  // out = y + a*x
nmfx_status mu_plain(hipStream_t st, float *X, const float *neg, const float *pos, long count);  // X .* (neg ./ max(pos, eps))
nmfx_status mu_plain_diff(hipStream_t st, const float *X0, const float *neg, const float *pos, long count, float *Xnew, float *dX);   // Xnew = X0 .* (neg ./ max(pos, eps)), dX = Xnew - X0
nmfx_status minmax_dev(hipStream_t st, const float *X_dev, long count, double *out_dev);   // [max, -min] (aux.hip)
nmfx_status scale_div(hipStream_t st, const float *X_dev, long count, double divide_by, float *out_dev);
nmfx_status cnmfsc_w_slices(hipStream_t st, const float *W0, const float *Nn, const float *G, long m, int K, int T, float *W);   // cnmfsc.m:257-263 from N = V*H_stack' and G = Hs*Hs' (aux.hip)
nmfx_status mu_plus_eps(hipStream_t st, float *X, const float *neg, const float *pos, long count);  // X .* (neg ./ (pos + eps))
nmfx_status transpose_f32(hipStream_t st, const float *in, long rows, long cols, float *out);    // out (cols x rows)
nmfx_status kl_pvec(hipStream_t st, const double *rowsum, const float *H, int K, long n, int T, double *Pvec, int halo_left = 0);
nmfx_status sum_over_t(hipStream_t st, const double *colsum, int K, int T, double *out);
nmfx_status d2f(hipStream_t st, const double *in, float *out, int count);
nmfx_status f2d(hipStream_t st, const float *in, double *out, int count);
// in: DEVICE staging buffer of `dtype`; out = (float)(in / divide_by)
nmfx_status cvt_to_f32(hipStream_t st, const void *in, int dtype, float *out, long count, double divide_by);
nmfx_status cvt_to_f64(hipStream_t st, const float *in, double *out, long count);

// ---- Hoyer projection (projfunc.hip): vectors are the COLUMNS of X (len x count), in place ----
// dir (or dir64, the direction as doubles) != nullptr: the vectors projected are src + mu*dir, formed in fp64 (the line-search step of
// nmfsc.m:154 / 205 fused into the load); src == nullptr means X itself.  X receives the result.
nmfx_status projfunc_cols(hipStream_t st, float *X, long len, int count, double k1, double k2, int nn, int *usediters_dev,
                          const float *dir = nullptr, double mu = 0.0, const float *src = nullptr, const double *dir64 = nullptr);
nmfx_status projfunc_cols_f64(hipStream_t st, double *X, long len, int count, double k1, double k2, int nn, int *usediters_dev);
// the same projection when every vector is split over the ranks of `comm` (len = local part, N_total = whole length);
// v_scratch: len*count doubles, flags: len*count bytes, red: 6*count doubles
nmfx_status projfunc_cols_dist(hipStream_t st, float *X, long len, int count, long N_total, double k1, double k2, int nn, const Comm &comm,
                               double *v_scratch, unsigned char *flags, double *red, const float *dir = nullptr, double mu = 0.0,
                               const float *src = nullptr, const double *dir64 = nullptr);

}  // namespace nmfx

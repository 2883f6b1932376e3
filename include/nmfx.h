/* nmfx -- MI355X-native NMF multiplicative-update engine: C ABI (libnmfx.so).
 *
 * The reference (colinvaz/nmf-toolbox) is pure MATLAB and has no FFI; the boundary a maintainer
 * would bind is therefore the MATLAB call surface itself.  Each entry point below states the
 * reference interface it replaces (file:line under the reference tree).  The MEX gateway and the
 * `.m` wrappers that keep the toolbox signatures are shown in INTEGRATION.md.
 *
 * Conventions
 *   - every matrix is COLUMN-MAJOR (MATLAB order): V[i + m*j], W[i + m*k + m*K*t], H[k + K*j]
 *   - no torch / C++ types cross this boundary: plain pointers, sizes, enums
 *   - every function returns nmfx_status; on failure nmfx_last_error() (thread-local) holds the
 *     message a wrapper turns into MATLAB error() / a Python exception
 *   - the library never frees or retains caller memory; one call = one blocking computation,
 *     except the nmfx_engine_* phase API, which is asynchronous on the caller's HIP stream
 *   - there is NO CPU fallback: without a usable gfx950 device compute calls fail with
 *     NMFX_ERR_NO_DEVICE
 *   - device arithmetic is fp32 (MFMA v_mfma_f32_32x32x2_f32) with fp64 scalar/vector reductions;
 *     eps is MATLAB's 2^-52, not FLT_EPSILON
 */
#ifndef NMFX_H
#define NMFX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version: bumped whenever a struct below grows or an entry point changes (600: nmfx_problem.multi_backend, the RCCL / exchange hooks, nmfx_abi_sizes).
 * A client compares nmfx_version() with the NMFX_VERSION it was BUILT against and nmfx_abi_sizes() with its own sizeof()s before the first call:
 * the library reads every field of the structs it is handed, so a client built against an older, shorter nmfx_problem must not call in
 * (matlab/nmfx_mex.c and nmf_toolbox_amd/_lib.py both refuse to). */
#define NMFX_VERSION 600

typedef enum {
    NMFX_OK = 0,
    NMFX_ERR_INVALID = 1,     /* bad argument (wrapper -> error()) */
    NMFX_ERR_NO_DEVICE = 2,   /* no HIP device / kernels not loadable: never a silent CPU path */
    NMFX_ERR_HIP = 3,         /* a HIP runtime call failed */
    NMFX_ERR_UNSUPPORTED = 4, /* valid in the reference, not implemented here yet */
    NMFX_ERR_NOMEM = 5,
    NMFX_ERR_NEGATIVE = 6     /* nmfsc.m:57-59 "Negative values in data!" */
} nmfx_status;

/* config.divergence strings of nmf.m:147-167 / cnmf.m:137-147 */
typedef enum {
    NMFX_DIV_EUCLIDEAN = 0,        /* 'euclidean' */
    NMFX_DIV_KL = 1,               /* 'kl_divergence', 'kl' */
    NMFX_DIV_IS = 2,               /* 'is_divergence', 'is' */
    NMFX_DIV_AB = 3,               /* 'ab_divergence', 'ab' (alpha, beta) */
    NMFX_DIV_EUCLIDEAN_NOCOST = 4  /* cnmf only: 'frobenius' or any unrecognised string -- euclidean
                                      updates, cost stays 0 (cnmf.m:137-147 vs 239-248) */
} nmfx_divergence;

typedef enum { NMFX_F32 = 0, NMFX_F64 = 1 } nmfx_dtype;

/* ------------------------------------------------------------------------------------------
 * Problem / result of one blocking factorisation with HOST buffers (what the MEX gateway binds).
 * Multi-source problems (cell-array arguments of nmf.m:114-117) are passed concatenated:
 * W = [W_1 ... W_S] (K_total columns), H = [H_1; ...; H_S], with per-source arrays of length
 * num_sources.  Random defaults (nmf.m:277,298) stay in the wrapper: W_init/H_init are required.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int64_t m, n;             /* size(V) */
    int32_t K_total;          /* sum of num_basis_elems */
    int32_t T;                /* context_len (cnmf.m:1); 1 for nmf / nmfsc */
    int32_t dtype;            /* nmfx_dtype of V, W_init, H_init and of the result W, H */
    const void *V;            /* m x n */
    const void *W_init;       /* m x K_total x T */
    const void *H_init;       /* K_total x n */
    int32_t divergence;       /* nmfx_divergence */
    double alpha, beta;       /* used only for NMFX_DIV_AB */
    int32_t num_sources;      /* S >= 1 */
    const int32_t *K_s;       /* [S] basis elements per source; NULL iff S == 1 */
    const double *W_sparsity; /* [S] additive lambda (nmf.m:168), already clamped >= 0; NULL = 0 */
    const double *H_sparsity; /* [S] (nmf.m:199); NULL = 0 */
    const uint8_t *W_fixed;   /* [S] (nmf.m:146); NULL = all false */
    const uint8_t *H_fixed;   /* [S] (nmf.m:177); NULL = all false */
    int32_t maxiter;          /* > 0 (wrapper applies the <=0 -> 100 rule, nmf.m:404-406) */
    double tolerance;         /* > 0 (wrapper applies the <=0 -> 1e-3 rule, nmf.m:409-411);
                                 NMFX extension: a NEGATIVE value disables the stop rule (benchmarks) */
    int32_t device;           /* HIP device ordinal */
    /* nmfsc only (nmfsc.m:87-110): Hoyer sparseness targets in [0,1]; <= 0 selects the MU branch */
    double sc_W_sparsity, sc_H_sparsity;
    int32_t path;             /* NMFX extension: 0 = auto, 1 = generic kernels only, 2 = require the fused kernels */
    /* NMFX extensions; all-zero = the reference's behaviour on p->device */
    double sc_stepsize_H0, sc_stepsize_W0; /* nmfsc / nmfsc_dev: initial line-search step sizes; <= 0 = 1 (nmfsc.m:133-134).  With
                                              result.stepsize_H / stepsize_W of a previous call this resumes a run exactly */
    int32_t sc_resume;        /* nmfx_nmfsc_dev only: W / H are the state a previous call left (skip the initial projections, nmfsc.m:94-110) */
    int32_t n_gpus;           /* nmfx_nmf / nmfx_cnmf / nmfx_lnmf / nmfx_nmfsc: 0 or 1 = one GPU (p->device); N > 1 = V and H column-sharded
                                 over N GPUs of this process, W replicated, ONE all-reduce of the packed W-step sums per iteration
                                 (SURVEY 8(e)); cnmf adds T-1 halo columns of H copied between neighbouring devices per iteration, nmfsc
                                 runs one host thread per shard with the sums of its line searches / projfunc reduced over the peers */
    const int32_t *device_ids;/* [n_gpus] HIP ordinals, or NULL = 0 .. n_gpus-1 */
    int32_t multi_backend;    /* the per-iteration exchange of the packed W-step sums behind nmfx_nmf / nmfx_cnmf / nmfx_lnmf with n_gpus >= 1:
                                 0 = auto (RCCL -- ncclCommInitAll over device_ids, ONE ncclAllReduce per device and iteration on its stream -- when the
                                 devices are distinct and librccl can be dlopen'ed, else the peer exchange; NMFX_MULTI_BACKEND=peer|rccl overrides auto),
                                 1 = peer reduce-scatter + all-gather over xGMI peer mappings (no library; the only one that takes one device named twice),
                                 2 = RCCL (an error when it cannot run).  With a value != 0, n_gpus == 1 also takes the sharded driver (one shard) */
} nmfx_problem;

typedef struct {
    void *W;                  /* caller-allocated m*K_total*T elements of problem.dtype */
    void *H;                  /* caller-allocated K_total*n */
    double *cost;             /* caller-allocated: maxiter entries (nmf, cnmf), maxiter+1 (nmfsc) */
    int32_t cost_len;         /* out: valid entries of cost (== iterations run; nmfsc: see nmfsc.m:137-139,238) */
    int32_t iters_run;        /* out */
    int32_t *tries_H;         /* nmfsc, optional [maxiter]: line-search tries per outer iteration (nmfsc.m:152-175) */
    int32_t *tries_W;         /* nmfsc, optional [maxiter] (nmfsc.m:203-226) */
    double stepsize_H, stepsize_W; /* out (nmfsc.m:133-134,178,228) */
    int32_t converged_early;  /* out: nmfsc step-size underflow return (nmfsc.m:170-174) */
} nmfx_result;

/* [W,H,cost] = nmf(V, num_basis_elems, config)            -- replaces nmf.m:1 (hot loop nmf.m:143-225) */
nmfx_status nmfx_nmf(const nmfx_problem *p, nmfx_result *r);
/* [W,H,cost] = cnmf(V, num_basis_elems, context_len, config) -- replaces cnmf.m:1 (hot loop cnmf.m:175-258) */
nmfx_status nmfx_cnmf(const nmfx_problem *p, nmfx_result *r);
/* [W,H,cost] = lnmf(V, num_basis_elems, config)           -- replaces lnmf.m:1 (loop lnmf.m:66-88).
 * divergence must be NMFX_DIV_KL, one source; cost keeps maxiter entries (the reference does not trim on break). */
nmfx_status nmfx_lnmf(const nmfx_problem *p, nmfx_result *r);
/* [W,H,Z,A,cost] = constrainednmf(V, labels, num_basis_elems, config) -- replaces constrainednmf.m:1 (loop constrainednmf.m:183-258;
 * SURVEY 8(f) row f4).  The caller has done the label bookkeeping of constrainednmf.m:147-170 (host control logic): V's columns
 * are already sorted by processed label (unlabelled samples first), and segments[0..nz] gives, for every column c of the cluster
 * matrix Z (K x nz), the range [segments[c], segments[c+1]) of sorted samples it owns (length-1 ranges for unlabelled samples, one
 * range per class after them), i.e. the non-zeros of row c of A.  H_init is ignored (H = Z*A); H_sparsity[0] / H_fixed[0] carry
 * config.Z_sparsity / config.Z_fixed; result.H = Z*A in SORTED sample order; Z_out receives Z (same dtype as the problem).
 * Divergences: euclidean, kl, is, and ab with alpha == 0 only -- the alpha ~= 0 expression at constrainednmf.m:229 is ill-formed
 * in the reference (element-wise product of a K x n and an m x n matrix) and is refused with NMFX_ERR_UNSUPPORTED. */
nmfx_status nmfx_constrainednmf(const nmfx_problem *p, const int64_t *segments, int64_t nz, const void *Z_init,
                                nmfx_result *r, void *Z_out);
/* [W,H,cost] = nmfsc(V, num_basis_elems, config)          -- replaces nmfsc.m:1 (hot loop nmfsc.m:141-245) */
nmfx_status nmfx_nmfsc(const nmfx_problem *p, nmfx_result *r);
/* [W,H,cost] = cnmfsc(V, num_basis_elems, context_len, config) -- replaces cnmfsc.m:1 (hot loop cnmfsc.m:155-277).
 * Uses sc_W_sparsity / sc_H_sparsity, T = context_len, W_fixed[0] / H_fixed[0]; result.tries_W needs maxiter*T entries. */
nmfx_status nmfx_cnmfsc(const nmfx_problem *p, nmfx_result *r);
/* nmfsc on DEVICE buffers, optionally on a column shard per rank ("multi-GPU nmfsc, distributed projfunc
 * reductions").  libnmfx does not link RCCL: every cross-rank sum goes through the caller's all-reduce callback, which must
 * reduce `count` elements of `dtype` (NMFX_F32 / NMFX_F64) at dev_ptr IN PLACE over all ranks, ordered after the work already
 * queued on `stream`, and return 0.  The same sequence of calls is made on every rank.  allreduce == NULL means one GPU.
 *   V  m x n_local fp32, already divided by the GLOBAL max(V(:)) (nmfsc.m:62 -- one MAX all-reduce the caller does itself)
 *   W  m x K (replicated, identical on every rank), H  K x n_local: updated in place
 *   p->n = n_local; n_total = global column count (the H sparseness target of nmfsc.m:102-106 is defined on whole rows);
 *   p->V / W_init / H_init / dtype are ignored; result.W / result.H are ignored, cost / tries_* / stepsize_* are filled.
 * Per outer iteration: ONE large all-reduce of [V*H' | H*H'] (m*K + K*K floats), an 8-byte sum per objective evaluation
 * (nmfsc.m:161,212,238) and, when H is projected, 4*K doubles per reduction of projfunc.m:22-53.  Column shards need the
 * fused kernels: any K <= 256 (padded internally to a multiple of 32 with zero components), m and n_local >= 64
 * (NMFX_ERR_UNSUPPORTED otherwise). */
typedef enum { NMFX_REDUCE_SUM = 0, NMFX_REDUCE_MAX = 1 } nmfx_reduce_op;
typedef int32_t (*nmfx_allreduce_fn)(void *ctx, void *dev_ptr, int64_t count, int32_t dtype, int32_t op, void *stream);
nmfx_status nmfx_nmfsc_dev(const nmfx_problem *p, const float *V_dev, float *W_dev, float *H_dev, int64_t n_total, void *stream,
                           nmfx_allreduce_fn allreduce, void *allreduce_ctx, nmfx_result *r);
/* Measurement hooks for nmfx_nmfsc / nmfx_nmfsc_dev on the calling thread (bench.py --workload c5): hipEvent pairs around every
 * launch group on the stream the kernels run on; read total ms / launch count per tag after the call returned. */
nmfx_status nmfx_nmfsc_profile(int32_t enable);
int32_t nmfx_nmfsc_profile_ntags(void);
const char *nmfx_nmfsc_profile_tag_name(int32_t tag);
nmfx_status nmfx_nmfsc_profile_read(double *ms_per_tag, int32_t *count_per_tag);
/* measurement hook (bench.py --workload c4sc): completion time, in seconds from the start of the iterations, of every outer iteration of the last
 * nmfx_cnmfsc call on this thread; returns their number */
int32_t nmfx_sc_iteration_seconds(double *out, int32_t capacity);
/* V_hat = ReconstructFromDecomposition(W, H)              -- replaces ReconstructFromDecomposition.m:1 */
nmfx_status nmfx_reconstruct(int64_t m, int64_t n, int32_t K, int32_t T, int32_t dtype, const void *W,
                             const void *H, void *V_hat, int32_t device);
/* [W_sorted,H_sorted] = SortDictionary(W, H)               -- replaces SortDictionary.m:1.  H and H_sorted may
 * be NULL; order_out (optional, [K]) receives the 0-based column permutation. */
nmfx_status nmfx_sort_dictionary(int64_t m, int32_t K, int64_t n, int32_t dtype, const void *W, const void *H,
                                 void *W_sorted, void *H_sorted, int32_t *order_out, int32_t device);
/* [v,usediters] = projfunc(s, k1, k2, nn) applied to `count` vectors of length N (stride N) -- replaces projfunc.m:1 */
nmfx_status nmfx_projfunc(int64_t N, int32_t count, int32_t dtype, const void *s, double k1, double k2,
                          int32_t nn, void *v, int32_t *usediters, int32_t device);

/* the same projection on DEVICE buffers (fp32), asynchronous on `stream`: X (N x count, column-major) receives the projection of
 * src + mu*dir formed in fp64 (src NULL = X itself, dir NULL = no step: the line-search candidates of nmfsc.m:154-157 in one kernel) */
nmfx_status nmfx_projfunc_dev(void *stream, float *X_dev, int64_t N, int32_t count, double k1, double k2, int32_t nn, const float *src_dev,
                              const float *dir_dev, double mu, int32_t *usediters_dev);

/* nmfsc.m:57-62 on a DEVICE-resident column shard, for callers of nmfx_nmfsc_dev (which wants V already divided by the GLOBAL max):
 *   nmfx_minmax_dev   out_dev[0] = max(X(:)), out_dev[1] = -min(X(:)) as doubles -- both "larger is more extreme", so ONE MAX all-reduce of the two over
 *                     the ranks gives the global pair; the caller raises "Negative values in data!" when the second is > 0 (nmfsc.m:57-59);
 *   nmfx_scale_dev    out = (float)((double)X / divide_by), element-wise (nmfsc.m:62 with the global max); out may be X.
 * Asynchronous on `stream`; fixed reduction order (run-to-run deterministic). */
nmfx_status nmfx_minmax_dev(void *stream, const float *X_dev, int64_t count, double *out_dev);
nmfx_status nmfx_scale_dev(void *stream, const float *X_dev, int64_t count, double divide_by, float *out_dev);

/* Measurement hook for the blocking calls (bench.py --api blocking): wall seconds the last nmfx_nmf / nmfx_cnmf / nmfx_lnmf /
 * nmfx_constrainednmf on the calling thread spent moving the host arrays in (host-side fp64 -> fp32 conversion on threads + DMA through
 * two pinned buffers), iterating, and moving the results out; and the bytes of host arrays read / written.  Any pointer may be NULL. */
nmfx_status nmfx_last_call_timing(double *ingest_s, double *iterate_s, double *egress_s, double *host_bytes_in, double *host_bytes_out);

/* the packed exchange of the last blocking multi-GPU call on this thread: mean milliseconds device 0's stream spent in it (hipEvent pairs around the first <= 32
 * exchanges), how many were timed, and the backend that ran (1 peer, 2 RCCL; 0: no such call) */
nmfx_status nmfx_last_call_exchange(double *ms_per_exchange, int32_t *exchanges_timed, int32_t *backend);
/* the RCCL library the backend loads ("" when none can be loaded) and its version code */
const char *nmfx_rccl_library(int32_t *version);
const char *nmfx_last_error(void);
int32_t nmfx_device_count(void);   /* 0 when no HIP device is usable */
int32_t nmfx_version(void);
/* sizeof(nmfx_problem), sizeof(nmfx_result), sizeof(nmfx_engine_desc) as THIS library was compiled (any pointer may be NULL) */
void nmfx_abi_sizes(int32_t *problem_bytes, int32_t *result_bytes, int32_t *engine_desc_bytes);

/* ------------------------------------------------------------------------------------------
 * Phase API on DEVICE buffers (fp32), asynchronous on the caller's HIP stream.  This is what
 * bench.py and the one-process-per-GPU driver use: V is column-sharded, W replicated, and the
 * caller all-reduces the packed W-step partials between wstep_partial and wstep_finish
 * (SURVEY.md 8(e)).  nmfx_nmf()/nmfx_cnmf() are thin loops over exactly these calls.
 * ------------------------------------------------------------------------------------------ */
typedef struct nmfx_engine nmfx_engine;

typedef struct {
    int64_t m, n_local;       /* local column shard of V */
    int32_t K_total, T;
    int32_t divergence;
    double alpha, beta;
    const float *lamW_col;    /* host [K_total] per-column lambda (source value repeated); NULL = 0 */
    const float *lamH_row;    /* host [K_total]; NULL = 0 */
    const uint8_t *fixW_col;  /* host [K_total]; NULL = 0 */
    const uint8_t *fixH_row;  /* host [K_total]; NULL = 0 */
    int32_t device;
    void *stream;             /* hipStream_t of the caller (NULL = default stream) */
    int64_t col_offset;       /* global index of the first local column (cnmf halo logic; 0 on 1 GPU) */
    int32_t path;             /* 0 = auto; 1 = force generic (materialised V_hat) path; 2 = force fused path */
    /* cnmf on a column shard: H carries halo_left + n_local + halo_right columns and V n_local + halo_right
     * (the convolution reaches T-1 columns into the neighbours); n_valid = how many of V's columns exist globally
     * (= n_local + halo_right except on the last rank).  The caller refreshes H's halos after every hstep.  All 0 on one GPU. */
    int32_t halo_left, halo_right;
    int64_t n_valid;
    int32_t algorithm;        /* 3 = constrainednmf rules (nmf's W step; the H step updates Z and sets H = Z*A,
                                 constrainednmf.m:213-237; needs nmfx_engine_set_constraint; one GPU only);
                                 2 = lnmf rules (lnmf.m:59,69-70,76: L1 columns, plain ratio, sqrt H update);
                                 0 = nmf rules (nmf.m:130-134,169: unit-L2 columns); 1 = cnmf rules
                                 (cnmf.m:157-166,196-199: slab Frobenius norm T, H rescaled at init only) */
    int32_t K_valid;          /* 0 = all K_total components are real.  Otherwise components k >= K_valid are zero padding (zero
                                 columns of W / rows of H, marked fixed by the caller) that only rounds K up to a kernel-friendly
                                 size: they contribute exact zeros everywhere and are skipped by the initial normalisation */
    int32_t flags;            /* bit 0: no transposed copy of V (euclidean paths keep V' = n x m next to V and run their H-step numerators on it; set the
                                 bit to give those m*n*4 bytes back at ~10 % of the iteration rate).  Every rank of a sharded run must pass the same
                                 flags: the kernel path, and with it the summation order, follows from the descriptor alone */
} nmfx_engine_desc;

/* bytes of device scratch the engine needs (caller allocates: torch tensor / hipMalloc) */
nmfx_status nmfx_engine_workspace_bytes(const nmfx_engine_desc *d, size_t *bytes);
/* number of fp32 elements of the packed all-reduce buffer [N | P-or-rowsum | ...] */
nmfx_status nmfx_engine_packed_count(const nmfx_engine_desc *d, size_t *count);
/* V, W, H, workspace, packed: DEVICE pointers owned by the caller; W and H are updated in place */
nmfx_status nmfx_engine_create(const nmfx_engine_desc *d, const float *V, float *W, float *H, void *workspace,
                               size_t workspace_bytes, float *packed, nmfx_engine **out);
void nmfx_engine_destroy(nmfx_engine *e);
/* nmf.m:130-139 / cnmf.m:155-171: normalise W (cnmf: and rescale H), form the initial V_hat state */
nmfx_status nmfx_engine_init(nmfx_engine *e);
/* W step, local part: fills packed[] with this shard's numerator/denominator sums (nmf.m:149-164, cnmf.m:187-192) */
nmfx_status nmfx_engine_wstep_partial(nmfx_engine *e);
/* W step, replicated part after the all-reduce: ratio update + normalisation (nmf.m:168-169, cnmf.m:193-199) */
nmfx_status nmfx_engine_wstep_finish(nmfx_engine *e);
/* H step (column-local, nmf.m:176-203 / cnmf.m:207-236).  On the generic path this also refreshes V_hat and leaves the
 * shard's cost of the iteration in the engine; on the fused path V_hat is never formed and the cost of iteration i is a
 * by-product of the W-step pass of iteration i+1 (same W, H), or of nmfx_engine_cost_pass(). */
nmfx_status nmfx_engine_hstep(nmfx_engine *e);
/* Column-sharded cnmf: with defer = 1, nmfx_engine_hstep stops after the H update so that the caller can refresh H's halo
 * columns from the neighbouring ranks; nmfx_engine_hstep_finish then refreshes V_hat / the cost with the new H. */
nmfx_status nmfx_engine_defer_hstep_finish(nmfx_engine *e, int32_t defer);
nmfx_status nmfx_engine_hstep_finish(nmfx_engine *e);
/* Row-chunked form of wstep_partial (fused path): chunk c of nchunks computes rows [c*m/nchunks, (c+1)*m/nchunks) of the W-step
 * sums into a contiguous block of `packed`; nmfx_engine_packed_chunk gives the element range that is final after chunk c (the last
 * one carries the small tail).  The multi-GPU driver all-reduces chunk c while chunk c+1 computes; wstep_finish reads the chunked
 * layout.  m must be a multiple of 128*nchunks.  nmfx_engine_wstep_partial == one chunk. */
nmfx_status nmfx_engine_wstep_partial_chunk(nmfx_engine *e, int32_t chunk, int32_t nchunks);
nmfx_status nmfx_engine_packed_chunk(nmfx_engine *e, int32_t chunk, int32_t nchunks, size_t *offset, size_t *count);
/* make the engine's cost refer to the CURRENT (W, H): no-op when it already does, else one fused S = W*H pass */
nmfx_status nmfx_engine_cost_pass(nmfx_engine *e);
int32_t nmfx_engine_is_fused(nmfx_engine *e);   /* 1 = fused kernels (cost lags one pass), 3 = cnmf on the fused shift-sum passes (Gram denominators),
                                                   4 = KL cnmf on the fused passes (cost lags one pass, R = V./V_hat in HBM),
                                                   2 = Gram form on the GEMM (no V_hat in HBM), 0 = materialised V_hat */
/* device pointer to the fp64 cost of the last hstep for the local shard: data-fit partial + lambda*L1 terms
 * (the W term is included only when the engine is rank 0, see nmfx_engine_set_rank0); sum over ranks = nmf.m:206-218 */
nmfx_status nmfx_engine_cost_ptr(nmfx_engine *e, double **dev_cost);
/* enqueue an 8-byte device-to-device copy of that cost into dst_dev on the engine's stream */
nmfx_status nmfx_engine_copy_cost(nmfx_engine *e, double *dst_dev);
nmfx_status nmfx_engine_set_rank0(nmfx_engine *e, int32_t is_rank0);
/* Where the cost of iteration i becomes available: 0 after hstep(i); 1 after wstep_partial(i+1) (fused KL passes: a by-product of the next
 * W-step pass); 2 after wstep_finish(i+1) (euclidean fused path: the cost in Gram form, 0.5*||V||^2 - <W, V*H'> + 0.5*<W, W*(H*H')>, out of the
 * column sums the W update forms anyway -- nmf.m:149-150's diagonal terms -- so the W-step pass needs no W*H product; when the residual gets too
 * small for fp32 to resolve that difference (cost < 5 % of 0.5*||V||^2) a device-side flag switches the explicit residual pass back on).
 * An engine of lag 2 may, from some iteration on, deliver the cost at point 1 already (it goes back to the one-pass kernel two W updates after the flag
 * was set -- a fixed distance, so the switch falls on the same iteration on every rank and in every run); nmfx_engine_cost_lag keeps returning 2 and
 * reading at point 2 -- after wstep_finish(i+1), BEFORE the next wstep_partial -- is right in both regimes. */
int32_t nmfx_engine_cost_lag(nmfx_engine *e);
/* Column shards + Gram-form cost: the mode decision needs the GLOBAL ||V||^2 and must be identical on every rank.  After nmfx_engine_init,
 * nmfx_engine_sumvv_local copies this shard's ||V_local||^2 (fp64) to dst_dev (0.0 when the engine has no such mode); the caller sums it over the
 * ranks (one 8-byte all-reduce, once) and hands the result back with nmfx_engine_sumvv_set_global.  Until then a sharded engine -- one that was told
 * its rank with nmfx_engine_set_rank0 -- keeps the explicit cost pass. */
nmfx_status nmfx_engine_sumvv_local(nmfx_engine *e, double *dst_dev);
nmfx_status nmfx_engine_sumvv_set_global(nmfx_engine *e, const double *src_dev);
/* algorithm 3 only, before nmfx_engine_init: host segments[0..nz] (see nmfx_constrainednmf) and the DEVICE cluster matrix Z (K x nz) */
nmfx_status nmfx_engine_set_constraint(nmfx_engine *e, const int64_t *segments_host, int64_t nz, float *Z_dev);
/* column shards without halos: everything between two all-reduces of `packed` in one call -- wstep_finish, hstep and, unless
 * `last`, the next iteration's wstep_partial (one host call per iteration next to the collective) */
nmfx_status nmfx_engine_between_allreduces(nmfx_engine *e, int32_t last);
/* the same with the read point of a lag-2 engine inside: right after its wstep_finish the cost of the previous iteration is copied (8 bytes, device to
 * device, on the engine's stream) to lag2_cost_dst_dev when that is not NULL -- the wstep_partial that follows may overwrite the engine's cost */
nmfx_status nmfx_engine_between_allreduces_cost(nmfx_engine *e, int32_t last, double *lag2_cost_dst_dev);
/* The engine keeps float64 MASTER copies of W and H in its workspace (nmf.m:168-169,199 run in double: the reference's state between iterations is
 * float64).  Every update reads the master, computes in double and writes both the master and the fp32 array the MFMA passes contract; the fp32 arrays the
 * caller handed over stay the results.  nmfx_engine_init derives the masters from W / H; a caller that rewrites W or H itself afterwards (a restore after
 * a stop, a halo refresh is NOT one: halos are read-only operands) calls nmfx_engine_sync_master.  The pointers (W64: m x K*T, H64: K x n_local, both
 * column-major, device) are NULL for constrainednmf's H, which is a gather of Z. */
nmfx_status nmfx_engine_sync_master(nmfx_engine *e);
/* nmfx_engine_init with the caller's float64 initial factors (DEVICE arrays: W_init64 m x K*T, H_init64 K x n_local -- the shard's own columns; either may be
 * NULL = take the fp32 array).  The masters start from them exactly and the fp32 arrays are rewritten as their images: what the blocking calls do with
 * float64 host buffers, so that MATLAB's doubles are not rounded on the way in (nmf.m:130-139 normalises in double). */
nmfx_status nmfx_engine_init_f64(nmfx_engine *e, const double *W_init64_dev, const double *H_init64_dev);
nmfx_status nmfx_engine_master_ptrs(nmfx_engine *e, double **W64_dev, double **H64_dev);
/* convenience for one GPU: `iters` full iterations, costs written to the DEVICE array dev_cost_out[iters] (may be NULL) */
nmfx_status nmfx_engine_iterate(nmfx_engine *e, int32_t iters, double *dev_cost_out);

/* Measurement hooks (bench.py): when enabled, every launch group of an iteration is bracketed by a hipEvent
 * pair on the engine's stream; after synchronising, read total ms and launch count per tag.
 * enable: 0 off | 1 every launch group | 2 only the big passes (fused passes, m*n*K GEMMs) -- the small-kernel groups and the K x K
 * products of an iteration are left unbracketed (an event pair costs ~5 us of stream time). */
nmfx_status nmfx_engine_profile(nmfx_engine *e, int32_t enable);
int32_t nmfx_engine_profile_ntags(void);
const char *nmfx_engine_profile_tag_name(int32_t tag);
nmfx_status nmfx_engine_profile_read(nmfx_engine *e, double *ms_per_tag, int32_t *count_per_tag);
/* algorithmic work of ONE launch behind `tag`: flops = 2*M*N*Kc of the contraction(s) it issues (by formula),
 * bytes = compulsory HBM traffic of its operands */
nmfx_status nmfx_engine_tag_work(nmfx_engine *e, int32_t tag, double *flops, double *bytes);

/* ------------------------------------------------------------------------------------------
 * Kernel-level entry points on device buffers (used by the parity tests to check each kernel
 * against the oracle in isolation).  C (M x N) = op(A) * op(B), all column-major fp32.
 * ------------------------------------------------------------------------------------------ */
typedef enum { NMFX_OP_N = 0, NMFX_OP_T = 1 } nmfx_op;
typedef enum {
    NMFX_PRO_NONE = 0,        /* x */
    NMFX_PRO_RATIO = 1,       /* x ./ x2          (V ./ V_hat,      nmf.m:152) */
    NMFX_PRO_RATIO_SQ = 2,    /* x ./ x2.^2       (V ./ V_hat.^2,   nmf.m:155) */
    NMFX_PRO_RECIP2 = 3,      /* 1 ./ x2          (1 ./ V_hat,      nmf.m:156) */
    NMFX_PRO_DIFF = 4,        /* x2 - x           (V_hat - V: dH = W'*V_hat - W'*V, nmfsc.m:148) */
    NMFX_PRO_POWPROD = 5      /* x.^e1 .* x2.^e2  (alpha-beta divergence: V.^alpha .* V_hat.^(beta-1), nmf.m:162) */
} nmfx_prologue;
nmfx_status nmfx_gemm_f32(void *stream, int32_t opA, int32_t opB, int64_t M, int64_t N, int64_t Kc,
                          const float *A, const float *A2, int64_t lda, int32_t proA, const float *B,
                          const float *B2, int64_t ldb, int32_t proB, float *C, int64_t ldc, int32_t accumulate,
                          void *workspace, size_t workspace_bytes);
/* C (M x N) = A (M x Kc) * B (Kc x N) accumulated in float64 on the fp64 matrix core: A(i, k) = A[i + lda*k] given as float64 (A64) or fp32 (A32),
 * B(k, j) = B[k + ldb*j] likewise, C[i + ldc*j] written as float64 and / or fp32 (either may be NULL).  What the engine runs nmf.m:150's V_hat*H' with, in
 * the Gram form W*(H*H'), from the float64 master copy of W. */
nmfx_status nmfx_gemm64(void *stream, int64_t M, int64_t N, int64_t Kc, const double *A64, const float *A32, int64_t lda, const double *B64,
                        const float *B32, int64_t ldb, double *C64, float *C32, int64_t ldc);

#ifdef __cplusplus
}
#endif
#endif /* NMFX_H */

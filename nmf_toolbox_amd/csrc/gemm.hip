// General fp32 MFMA GEMM with operand views, fused element-wise prologues and a fused divergence
// epilogue (gfx950).  C (M x N, column-major) = op(A) * op(B).
//
// This is the "materialised V_hat" building block of the engine: every contraction of
// nmf.m:149-203, cnmf.m:187-236, nmfsc.m:144-238 and ReconstructFromDecomposition.m:31-38 is one
// launch of it -- the convolutive shift-sums are expressed as stacked/shifted operand VIEWS
// (nmfx_internal.h), never as padded copies, and V./V_hat style element maps are applied while
// the tile is staged into LDS.
//
// Structure: 256 threads = 4 waves (2x2), block tile BM x BN x 32, wave tile (BM/2)x(BN/2) built from
// v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD).  Operands are staged through LDS as
// S[k][r] (r contiguous) so each MFMA operand register is one conflict-free ds_read_b32; the next
// tile's global loads are issued before the current tile's MFMAs and written to the other LDS
// buffer after them (one barrier per k-tile).  The MFMA roles are swapped (B tile feeds the "A"
// port) so that lanes 0-31 of an accumulator register hold 32 CONSECUTIVE ROWS i of one column of
// C: stores to C and loads of V in the epilogue are 128-byte contiguous segments.
#include <cstdlib>
#include <type_traits>

#include <algorithm>
#include <cstdlib>
#include "gemm_common.h"

namespace nmfx {

// one thread's share of a BR x BK operand tile: NCH chunks of 4 elements along the contiguous direction
template <int BR, bool KC, bool FAST, bool HEAVY>
struct Loader {
    static constexpr int NCH = BR * BK / 4 / NTHREADS;
    static constexpr int LDS_STRIDE = BR + (KC ? 1 : 0);
    // RC: chunk = rows 4c..4c+3 of k-row (kq + p*KSTEP);  KC: chunk = k 4c..4c+3 of row (rq + p*RSTEP)
    static constexpr int CPR = KC ? (BK / 4) : (BR / 4);  // chunks per line
    static constexpr int LSTEP = NTHREADS / CPR;          // lines covered per pass
    float4 reg[NCH];
    long offr[KC ? NCH : 1];
    int gr[KC ? NCH : 1];
    int c, q;

    __device__ __forceinline__ void init(const OpView &v, int tid, int r_tile0, long R) {
        c = tid % CPR;
        q = tid / CPR;
        if (FAST) {
            if (KC) {
#pragma unroll
                for (int p = 0; p < NCH; ++p) dec_r(v, r_tile0 + q + p * LSTEP, offr[p], gr[p]);
            } else {
                dec_r(v, r_tile0 + 4 * c, offr[0], gr[0]);
            }
        }
    }
    __device__ __forceinline__ float elem(const OpView &v, int r, int kc, long R, long Kend) {
        if (r >= R || kc >= Kend) return 0.0f;
        long o1, o2; int g1, g2;
        dec_r(v, r, o1, g1);
        dec_k(v, kc, o2, g2);
        if (g1 + g2 < 0) return 0.0f;
        float x = v.p[o1 + o2];
        float y = v.p2 ? v.p2[o1 + o2] : 1.0f;
        return pro1<HEAVY>(v.func, x, y, v.e1, v.e2);
    }
    __device__ __forceinline__ void load(const OpView &v, int r_tile0, int k0, long R, long Kend) {
        if (FAST) {
            if (KC) {
                long ok; int gk;
                dec_k(v, k0 + 4 * c, ok, gk);
#pragma unroll
                for (int p = 0; p < NCH; ++p) {
                    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (gr[p] + gk >= 0) {
                        x = *reinterpret_cast<const float4 *>(v.p + offr[p] + ok);
                        if (v.func != NMFX_PRO_NONE) x = pro4<HEAVY>(v.func, x, *reinterpret_cast<const float4 *>(v.p2 + offr[p] + ok), v.e1, v.e2);
                    }
                    reg[p] = x;
                }
            } else {
#pragma unroll
                for (int p = 0; p < NCH; ++p) {
                    long ok; int gk;
                    dec_k(v, k0 + q + p * LSTEP, ok, gk);
                    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (gr[0] + gk >= 0) {
                        x = *reinterpret_cast<const float4 *>(v.p + offr[0] + ok);
                        if (v.func != NMFX_PRO_NONE) x = pro4<HEAVY>(v.func, x, *reinterpret_cast<const float4 *>(v.p2 + offr[0] + ok), v.e1, v.e2);
                    }
                    reg[p] = x;
                }
            }
        } else {
#pragma unroll
            for (int p = 0; p < NCH; ++p) {
                int line = q + p * LSTEP;
                float4 x;
                if (KC) {
                    int r = r_tile0 + line, kc = k0 + 4 * c;
                    x = make_float4(elem(v, r, kc, R, Kend), elem(v, r, kc + 1, R, Kend), elem(v, r, kc + 2, R, Kend),
                                    elem(v, r, kc + 3, R, Kend));
                } else {
                    int r = r_tile0 + 4 * c, kc = k0 + line;
                    x = make_float4(elem(v, r, kc, R, Kend), elem(v, r + 1, kc, R, Kend), elem(v, r + 2, kc, R, Kend),
                                    elem(v, r + 3, kc, R, Kend));
                }
                reg[p] = x;
            }
        }
    }
    __device__ __forceinline__ void store(float *S) const {
#pragma unroll
        for (int p = 0; p < NCH; ++p) {
            int line = q + p * LSTEP;
            if (KC) {
                S[(4 * c + 0) * LDS_STRIDE + line] = reg[p].x;
                S[(4 * c + 1) * LDS_STRIDE + line] = reg[p].y;
                S[(4 * c + 2) * LDS_STRIDE + line] = reg[p].z;
                S[(4 * c + 3) * LDS_STRIDE + line] = reg[p].w;
            } else {
                *reinterpret_cast<float4 *>(&S[line * LDS_STRIDE + 4 * c]) = reg[p];
            }
        }
    }
};


// HEAVY: instantiations that carry the powf-based element maps / alpha-beta cost (kept out of the common kernels: the
// inlined powf bodies cost registers and scratch in every variant otherwise)
template <int BM, int BN, bool A_KC, bool B_KC, bool FAST, bool HEAVY = false>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using LA = Loader<BM, A_KC, FAST, HEAVY>;
    using LB = Loader<BN, B_KC, FAST, HEAVY>;
    constexpr int LDA_S = LA::LDS_STRIDE, LDB_S = LB::LDS_STRIDE;
    constexpr int A_SZ = BK * LDA_S, B_SZ = BK * LDB_S;
    constexpr int A_SZ_AL = (A_SZ + 3) & ~3, B_SZ_AL = (B_SZ + 3) & ~3;
    constexpr int BUF_SZ = A_SZ_AL + B_SZ_AL;  // buffer b: [A tile | B tile] at smem + b*BUF_SZ
    constexpr int MR = BM / 64, NR = BN / 64;  // 32x32 MFMA tiles per wave along i / j

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wi0 = (wave & 1) * (BM / 2), wj0 = (wave >> 1) * (BN / 2);
    const int i_tile0 = blockIdx.x * BM, j_tile0 = blockIdx.y * BN;

    long kbeg = 0, kend = p.Kc;
    float *C = p.C;
    if (p.splitk > 1) {
        kbeg = (long)blockIdx.z * p.kc_per_split;
        kend = kbeg + p.kc_per_split < p.Kc ? kbeg + p.kc_per_split : p.Kc;
        C += (long)blockIdx.z * p.slab_stride;
    }

    LA la; LB lb;
    la.init(p.A, tid, i_tile0, p.M);
    lb.init(p.B, tid, j_tile0, p.N);

    f32x16 acc[NR][MR];
#pragma unroll
    for (int a = 0; a < NR; ++a)
#pragma unroll
        for (int b = 0; b < MR; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.0f;

    const int ntiles = (int)((kend - kbeg + BK - 1) / BK);
    if (ntiles > 0) {
        la.load(p.A, i_tile0, (int)kbeg, p.M, kend);
        lb.load(p.B, j_tile0, (int)kbeg, p.N, kend);
        la.store(smem);
        lb.store(smem + A_SZ_AL);
    }
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        if (t + 1 < ntiles) {
            la.load(p.A, i_tile0, (int)kbeg + (t + 1) * BK, p.M, kend);
            lb.load(p.B, j_tile0, (int)kbeg + (t + 1) * BK, p.N, kend);
        }
        const float *Ac = smem + cur * BUF_SZ, *Bc = Ac + A_SZ_AL;
        // operands of step kk+1 are read from LDS before the MFMAs of step kk issue (in-order wave: a read placed after
        // them would only start when the last MFMA has issued); sched_barrier keeps hipcc from re-clustering the reads
        float fa[NR], fb[MR], ga[NR], gb[MR];
#pragma unroll
        for (int a = 0; a < NR; ++a) fa[a] = Bc[h * LDB_S + wj0 + 32 * a + l31];
#pragma unroll
        for (int b = 0; b < MR; ++b) fb[b] = Ac[h * LDA_S + wi0 + 32 * b + l31];
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            if (kk + 1 < BK / 2) {
#pragma unroll
                for (int a = 0; a < NR; ++a) ga[a] = Bc[(2 * kk + 2 + h) * LDB_S + wj0 + 32 * a + l31];
#pragma unroll
                for (int b = 0; b < MR; ++b) gb[b] = Ac[(2 * kk + 2 + h) * LDA_S + wi0 + 32 * b + l31];
            }
#pragma unroll
            for (int a = 0; a < NR; ++a)
#pragma unroll
                for (int b = 0; b < MR; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a], fb[b], acc[a][b], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < NR; ++a) fa[a] = ga[a];
#pragma unroll
            for (int b = 0; b < MR; ++b) fb[b] = gb[b];
            __builtin_amdgcn_sched_barrier(0);
        }
        if (t + 1 < ntiles) {
            la.store(smem + (cur ^ 1) * BUF_SZ);
            lb.store(smem + (cur ^ 1) * BUF_SZ + A_SZ_AL);
        }
        __syncthreads();
    }

    // epilogue: acc[a][b][e] = C[i][j], i = i_tile0+wi0+32b+l31, j = j_tile0+wj0+32a+(e&3)+8(e>>2)+4h
    double part = 0.0;
#pragma unroll
    for (int a = 0; a < NR; ++a)
#pragma unroll
        for (int b = 0; b < MR; ++b) {
            const long i = i_tile0 + wi0 + 32 * b + l31;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const long j = j_tile0 + wj0 + 32 * a + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (FAST || (i < p.M && j < p.N)) {
                    float s = acc[a][b][e];
                    if (p.epi == EPI_COST) {
                        if (p.cost_ncols == 0 || j < p.cost_ncols) part += div_term<HEAVY>(p.cost_div, p.Vref[i + p.ldv * j], s, p.cost_alpha, p.cost_beta);
                        if (p.store_c) C[i + p.ldc * j] = s;
                    } else {
                        if (p.accumulate) s += C[i + p.ldc * j];
                        if (p.clamp0) s = fmaxf(s, 0.0f);   // cnmfsc.m:262  V_hat = max(V_hat + dW*Hs, 0)
                        C[i + p.ldc * j] = s;
                    }
                }
            }
        }
    if (p.epi == EPI_COST) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        double *red = reinterpret_cast<double *>(smem);
        __syncthreads();
        if (lane == 0) red[wave] = part;
        __syncthreads();
        if (tid == 0) p.cost_partials[(long)blockIdx.y * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    }
}

template <int BM, int BN, bool A_KC, bool B_KC, bool FAST, bool HEAVY = false>
static nmfx_status launch_cfg(hipStream_t st, const GemmParams &p) {
    using LA = Loader<BM, A_KC, FAST, HEAVY>;
    using LB = Loader<BN, B_KC, FAST, HEAVY>;
    constexpr int A_SZ_AL = (BK * LA::LDS_STRIDE + 3) & ~3, B_SZ_AL = (BK * LB::LDS_STRIDE + 3) & ~3;
    const size_t lds = sizeof(float) * 2 * (A_SZ_AL + B_SZ_AL);
    auto kern = gemm_kernel<BM, BN, A_KC, B_KC, FAST, HEAVY>;
    static LdsAttrOnce lds_attr;
    TRY(lds_attr.set(reinterpret_cast<const void *>(kern), (int)lds));
    dim3 grid((unsigned)((p.M + BM - 1) / BM), (unsigned)((p.N + BN - 1) / BN), (unsigned)(p.splitk > 1 ? p.splitk : 1));
    hipLaunchKernelGGL(kern, grid, dim3(NTHREADS), lds, st, p);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}


static bool view_fast_ok(const OpView &v) {
    if ((reinterpret_cast<uintptr_t>(v.p) & 15) || (v.p2 && (reinterpret_cast<uintptr_t>(v.p2) & 15))) return false;
    if (v.ld % 4) return false;
    if (v.mode >= VIEW_HSTACK_KC && (v.blk % 4)) return false;
    if (v.mode == VIEW_WSTACK_KC && (v.tstride % 4)) return false;
    return true;
}

template <int BM, int BN, bool FAST, bool HEAVY = false>
static nmfx_status dispatch_views(hipStream_t st, const GemmParams &p) {
    const bool akc = is_kc(p.A.mode), bkc = is_kc(p.B.mode);
    if (akc && bkc) return launch_cfg<BM, BN, true, true, FAST, HEAVY>(st, p);
    if (akc) return launch_cfg<BM, BN, true, false, FAST, HEAVY>(st, p);
    if (bkc) return launch_cfg<BM, BN, false, true, FAST, HEAVY>(st, p);
    return launch_cfg<BM, BN, false, false, FAST, HEAVY>(st, p);
}

void gemm_tile_shape(long M, long N, int &bm, int &bn) {
    bm = 128; bn = 128;
    if (M % 128 != 0 && M % 64 == 0) bm = 64;   // (320, 448, ...: whole 64-row tiles instead of the guarded-edge kernel)
    if (bm == 128 && N % 128 != 0 && N % 64 == 0 && N <= 192) bn = 64;
    // few output tiles (W*(H*H'): 64 of them at C2): half-height tiles double the workgroups of a launch that cannot fill the chip anyway
    if (bm == 128 && bn == 128 && M % 128 == 0 && N % 128 == 0 && (M / 128) * (N / 128) < 256 && M >= 256) bm = 64;
}

// pipelined kernel: no powf maps.  vec = float4 loads: 16-byte aligned views
// whose contiguous dimension is a multiple of 4 (edge tiles in M, N and the last k-tile are guarded chunk-wise); otherwise the
// dword-load variant (odd leading dimensions), which only needs stacked r-views to keep 4-element chunks inside one t block.
static bool pipe_ok(const GemmParams &p, int &bm, int &bn, bool &fast, bool &heavy, bool &vec) {
    gemm_tile_shape(p.M, p.N, bm, bn);
    const long kspan = p.splitk > 1 ? p.kc_per_split : p.Kc;
    const bool views = view_fast_ok(p.A) && view_fast_ok(p.B);
    fast = (p.M % bm == 0) && (p.N % bn == 0) && (p.Kc % BK == 0) && (kspan % BK == 0) && views;
    heavy = p.A.func == NMFX_PRO_POWPROD || p.B.func == NMFX_PRO_POWPROD || (p.epi == EPI_COST && p.cost_div == NMFX_DIV_AB);
    auto kview_ok = [](const OpView &v) { return !(v.mode >= VIEW_HSTACK_KC && is_kc(v.mode)) || v.blk >= 4; };   // a 4-chunk crosses at most one block edge
    auto rview_ok = [](const OpView &) { return true; };   // stacked r-views with K % 4 != 0: the dword variant decodes every row of a chunk
    const bool a_dim = is_kc(p.A.mode) ? p.Kc % 4 == 0 : p.M % 4 == 0;
    const bool b_dim = is_kc(p.B.mode) ? p.Kc % 4 == 0 : p.N % 4 == 0;
    vec = views && a_dim && b_dim;
    return !heavy && kview_ok(p.A) && kview_ok(p.B) && rview_ok(p.A) && rview_ok(p.B) && (p.splitk <= 1 || kspan % BK == 0);
}
bool gemm_pipe_eligible(const GemmParams &p) {
    int bm, bn; bool fast, heavy, vec;
    return p.M > 0 && p.N > 0 && pipe_ok(p, bm, bn, fast, heavy, vec);
}

nmfx_status launch_gemm(hipStream_t st, const GemmParams &p, long *blocks_out) {
    if (blocks_out) *blocks_out = 0;
    if (p.M <= 0 || p.N <= 0) return NMFX_OK;
    int bm, bn;
    bool fast, heavy, vec;
    const bool pipe = pipe_ok(p, bm, bn, fast, heavy, vec);
    if (p.zbatch > 0 && !pipe) { set_error("launch_gemm: z-batched launch needs the pipelined kernel"); return NMFX_ERR_INVALID; }
    if (!fast || heavy) bm = bn = 128;
    if (heavy) fast = fast && (p.M % 128 == 0) && (p.N % 128 == 0);
    if (blocks_out) *blocks_out = ((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn);
    if (pipe) {
        const long kspan = p.splitk > 1 ? p.kc_per_split : p.Kc;
        int tm, tn;
        gemm_tile_shape(p.M, p.N, tm, tn);
        auto ktile_ok = [](const OpView &v) { return !(v.mode >= VIEW_HSTACK_KC && is_kc(v.mode)) || v.blk % BK == 0; };   // tile-uniform t
        const bool whole = vec && p.M % tm == 0 && p.N % tn == 0 && p.Kc % BK == 0 && kspan % BK == 0 && ktile_ok(p.A) && ktile_ok(p.B);
        return whole ? dispatch_pipe_whole(st, p, tm, tn) : dispatch_pipe_edge(st, p, bm, bn, vec);
    }
    if (heavy) return fast ? dispatch_views<128, 128, true, true>(st, p) : dispatch_views<128, 128, false, true>(st, p);
    if (bm == 64) return fast ? dispatch_views<64, 128, true>(st, p) : dispatch_views<128, 128, false>(st, p);
    if (bn == 64) return fast ? dispatch_views<128, 64, true>(st, p) : dispatch_views<128, 128, false>(st, p);
    return fast ? dispatch_views<128, 128, true>(st, p) : dispatch_views<128, 128, false>(st, p);
}

long gemm_grid_blocks(long M, long N) {   // upper bound over the tile shapes launch_gemm may pick
    return ((M + 63) / 64) * ((N + 63) / 64);
}

int gemm_pick_split(long M, long N, long Kc) {
    int bm, bn;
    gemm_tile_shape(M, N, bm, bn);
    const long tiles = ((M + bm - 1) / bm) * ((N + bn - 1) / bn);
    const long ktiles = (Kc + BK - 1) / BK;
    int split = 1;
    if (tiles < 512 && ktiles >= 8) {   // two workgroups fit a CU: aim for >= 512 of them
        split = (int)((512 + tiles - 1) / tiles);
        if (split > 256) split = 256;   // K x K Grams over n: one output tile, all parallelism must come from the contraction (one workgroup per CU)
        if (split > ktiles / 4) split = (int)(ktiles / 4);
    }
    return split < 1 ? 1 : split;
}

// slabs of the VALU kernel for tiny products (tiny_gemm_kernel below): a small output over a long contraction needs the contraction split
// finely to put a few hundred waves on the chip
static bool tiny_size(long M, long N, long Kc) {
    const double lim = (double)(1 << 25);
    return M > 0 && N > 0 && Kc > 0 && 2.0 * (double)M * (double)N * (double)Kc <= lim;
}
static long tiny_split(long M, long N, long Kc, long *per) {
    long S = 1;
    *per = Kc;
    if (Kc >= 256 && M * N <= 65536) {
        *per = std::max<long>(32, (Kc + 63) / 64);
        S = (Kc + *per - 1) / *per;
    }
    return S;
}
size_t gemm_scratch_bytes(long M, long N, long Kc) {
    long split = gemm_pick_split(M, N, Kc), per = 0;
    if (tiny_size(M, N, Kc)) split = std::max(split, tiny_split(M, N, Kc, &per));
    return split > 1 ? sizeof(float) * (size_t)M * (size_t)N * split : 0;
}

// Products too small for a 128 x 128 MFMA tile and its software pipeline to be worth starting (a 32 x 32 Gram over 513 rows took 30 us, the four
// K x K products of a 513 x 2000, K = 32 euclidean iteration 80 of its 137 us): one thread per output element on the VALU, fp64 accumulation,
// the contraction split over blockIdx.y into slabs when the output alone cannot fill the chip
__global__ __launch_bounds__(256) void tiny_gemm_kernel(const float *A, long lda, int a_kc, const float *B, long ldb, int b_kc, long M, long N, long Kc, long per,
                                                        float *C, long ldc, long slab_stride) {
    const long o = (long)blockIdx.x * 256 + threadIdx.x;
    if (o >= M * N) return;
    const long r = o % M, c = o / M;
    const long k0 = (long)blockIdx.y * per, k1 = k0 + per < Kc ? k0 + per : Kc;
    const float *a = a_kc ? A + lda * r : A + r;
    const float *b = b_kc ? B + ldb * c : B + c;
    const long sa = a_kc ? 1 : lda, sb = b_kc ? 1 : ldb;
    double acc = 0.0;
    for (long k = k0; k < k1; ++k) acc = fma((double)a[k * sa], (double)b[k * sb], acc);
    C[(long)blockIdx.y * slab_stride + r + ldc * c] = (float)acc;
}
static bool tiny_plain(const OpView &v) { return (v.mode == VIEW_RC || v.mode == VIEW_KC) && v.func == NMFX_PRO_NONE && !v.p2; }

nmfx_status gemm_auto(hipStream_t st, GemmParams p, void *scratch, size_t scratch_bytes) {
    if (p.epi == EPI_STORE && !p.accumulate && !p.clamp0 && p.zbatch == 0 && p.M > 0 && p.N > 0 && p.Kc > 0 && tiny_plain(p.A) && tiny_plain(p.B) &&
        tiny_size(p.M, p.N, p.Kc)) {
        long per = p.Kc, S = (p.ldc == p.M && scratch) ? tiny_split(p.M, p.N, p.Kc, &per) : 1;
        if (S > 1) {
            const long cap = (long)(scratch_bytes / (sizeof(float) * (size_t)p.M * p.N));   // slabs the scratch holds
            if (cap < S) { per = cap >= 2 ? (p.Kc + cap - 1) / cap : p.Kc; S = cap >= 2 ? (p.Kc + per - 1) / per : 1; }
        }
        if (S <= 1) { S = 1; per = p.Kc; }
        float *dst = S > 1 ? static_cast<float *>(scratch) : p.C;
        hipLaunchKernelGGL(tiny_gemm_kernel, dim3((unsigned)((p.M * p.N + 255) / 256), (unsigned)S), dim3(256), 0, st, p.A.p, p.A.ld, is_kc(p.A.mode) ? 1 : 0,
                           p.B.p, p.B.ld, is_kc(p.B.mode) ? 1 : 0, p.M, p.N, p.Kc, per, dst, S > 1 ? p.M : p.ldc, p.M * p.N);
        NMFX_HIP(hipGetLastError());
        if (S > 1) return reduce_slabs(st, dst, (int)S, p.M * p.N, p.M * p.N, p.C, 0);
        return NMFX_OK;
    }
    const long ktiles = (p.Kc + BK - 1) / BK;
    int split = 1;
    if (p.epi == EPI_STORE && scratch) {
        split = gemm_pick_split(p.M, p.N, p.Kc);
        while (split > 1 && sizeof(float) * (size_t)p.M * p.N * split > scratch_bytes) --split;
    }
    if (split <= 1) {
        p.splitk = 1;
        return launch_gemm(st, p);
    }
    long per = ((ktiles + split - 1) / split) * BK;
    split = (int)((p.Kc + per - 1) / per);
    float *Cout = p.C;
    const long ldc_out = p.ldc;
    const int acc_out = p.accumulate;
    p.splitk = split;
    p.kc_per_split = per;
    p.C = static_cast<float *>(scratch);
    p.ldc = p.M;
    p.slab_stride = p.M * p.N;
    p.accumulate = 0;
    nmfx_status s = launch_gemm(st, p);
    if (s != NMFX_OK) return s;
    if (ldc_out == p.M) return reduce_slabs(st, p.C, split, p.slab_stride, p.M * p.N, Cout, acc_out);
    for (long j = 0; j < p.N; ++j) {  // strided destination (rare): column by column
        s = reduce_slabs(st, p.C + j * p.M, split, p.slab_stride, p.M, Cout + j * ldc_out, acc_out);
        if (s != NMFX_OK) return s;
    }
    return NMFX_OK;
}

}  // namespace nmfx

// The small products of the euclidean Gram form with float64 accumulation, on the fp64 matrix core (v_mfma_f64_16x16x4_f64, gfx950):
//     P = W * (H*H')          (nmf.m:149-150 in the form of SURVEY A.2: V_hat*H' = W*(H*H'))
// The K-long fp32 accumulation chain of this product -- not the storage of W -- is what put H-fixed problems past the 1e-5 contract (round 4: 1.06e-5 ... 1.7e-5
// on W; the chain's rounding differs from iteration to iteration and the W update amplifies it; scripts/emu_precision.py).  m*K*K multiply-adds: a few
// per cent of the n-long contractions next to it, so it is simply done in double, with the float64 master copy of W as the left operand.
#include <type_traits>
#include "nmfx_internal.h"

namespace nmfx {

namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int G64_BM = 64, G64_BN = 32, G64_BK = 16;
constexpr int G64_LD = 80;   // LDS row stride in doubles: 16 (mod 32), so the two k-rows a 32-lane group of ds_read_b64 touches fall on disjoint bank halves

// C(i, j) = sum_k A(i, k) * B(k, j);  A(i, k) = A[i + lda*k] (float64 or fp32), B(k, j) = B[k + ldb*j] (fp32 or float64), C[i + ldc*j] (float64 and / or fp32)
// 4 waves, each a 32 x 16 block of the 64 x 32 tile (two MFMA blocks along i).  The MFMA is fed transposed (first operand = B', second = A) so that the 16 lanes of a
// result register run along i, the contiguous dimension of C.
// These products are small (cnmf at C4: 4096 x 512 x 512) and the kernel is latency-bound, not MFMA-bound: the PMC pass of round 5's first version (64 x 64 tiles, one
// tile prefetched) shows the matrix core busy 13 % and the waves waiting 63 % of their cycles (profiles/r5_20_c4_pmc.md).  Measured at the engine's shapes
// (scripts/bench_gemm64.py, profiles/r5_24_gemm64_tiles.txt): 64 x 64 tiles 77 / 104 us (4096 x 512 x 512 / 16384 x 256 x 256), 64 x 32 tiles -- twice the workgroups,
// four and more per CU -- 65 / 69 us, 32 x 32 tiles 81 / 78 us; a second tile in flight (two register sets: the tile that goes to LDS at the end of a trip was
// requested a trip earlier) is worth 1-2 us on top.  33 TFLOP/s of the fp64 matrix core's 78; C4's product 106 -> 71 us inside the iteration.
template <bool A64, bool B64>
__global__ __launch_bounds__(256) void gemm64_kernel(const void *__restrict__ Ap, long lda, const void *__restrict__ Bp, long ldb, long M, long N, long Kc,
                                                      double *__restrict__ C64, float *__restrict__ C32, long ldc) {
    constexpr int NB = 1, MB = 2, PF = 2;                  // the 64 x 32 tile with two tiles in flight: the fastest of the (MB, NB, PF) variants tried, see above
    constexpr int BN = 32 * NB, BPT = BN * G64_BK / 256;   // B elements per thread and tile
    constexpr int LDB = NB == 2 ? G64_LD : 48;             // 48 = 16 (mod 32) as well
    constexpr int BM = 32 * MB, APT = BM * G64_BK / 256, LDA = MB == 2 ? G64_LD : 48;
    __shared__ double As[2][G64_BK * LDA], Bs[2][G64_BK * LDB];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w & 1, wn = w >> 1;
    const long i0 = (long)blockIdx.x * BM, j0 = (long)blockIdx.y * BN;
    // loaders: A -- thread (row = tid & 63, k = 4*(tid >> 6) + u); B -- thread (col = tid / (16 / BPT), k = BPT*(tid % (16 / BPT)) + u)
    const int a_row = tid % BM, a_k = APT * (tid / BM);
    const int b_col = tid / (G64_BK / BPT), b_k = BPT * (tid % (G64_BK / BPT));
    const bool a_ok = i0 + a_row < M, b_ok = j0 + b_col < N;
    double ra[2][APT], rb[2][BPT];
    auto gload = [&](long k0, int set) {
#pragma unroll
        for (int u = 0; u < APT; ++u) {
            const long ka = k0 + a_k + u;
            double va = 0.0;
            if (a_ok && ka < Kc) va = A64 ? static_cast<const double *>(Ap)[i0 + a_row + lda * ka] : (double)static_cast<const float *>(Ap)[i0 + a_row + lda * ka];
            ra[set][u] = va;
        }
#pragma unroll
        for (int u = 0; u < BPT; ++u) {
            const long kb = k0 + b_k + u;
            double vb = 0.0;
            if (b_ok && kb < Kc) vb = B64 ? static_cast<const double *>(Bp)[kb + ldb * (j0 + b_col)] : (double)static_cast<const float *>(Bp)[kb + ldb * (j0 + b_col)];
            rb[set][u] = vb;
        }
    };
    auto lstore = [&](int buf, int set) {
#pragma unroll
        for (int u = 0; u < APT; ++u) As[buf][(a_k + u) * LDA + a_row] = ra[set][u];
#pragma unroll
        for (int u = 0; u < BPT; ++u) Bs[buf][(b_k + u) * LDB + b_col] = rb[set][u];
    };
    f64x4 acc[MB][NB];
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[a][b][e] = 0.0;
    const int nk = (int)((Kc + G64_BK - 1) / G64_BK);
    gload(0, 0);
    lstore(0, 0);
    if (PF == 2 && nk > 1) gload((long)G64_BK, 1);
    __syncthreads();
    const int l15 = lane & 15, lk = lane >> 4;
    auto trip = [&](int k1, auto set_far_c) {   // tile k1 is in LDS buffer k1 & 1; tile k1 + 1 in register set (k1 + 1) & 1; tile k1 + 2 is requested into set k1 & 1
        constexpr int set_far = decltype(set_far_c)::value, set_next = set_far ^ 1;
        const int buf = k1 & 1;
        if (PF == 2) { if (k1 + 2 < nk) gload((long)(k1 + 2) * G64_BK, set_far); }
        else if (k1 + 1 < nk) gload((long)(k1 + 1) * G64_BK, set_next);   // PF == 1: one tile ahead, one register set in use at a time
#pragma unroll
        for (int kk = 0; kk < G64_BK / 4; ++kk) {
            double af[MB], bf[NB];
#pragma unroll
            for (int a = 0; a < MB; ++a) af[a] = As[buf][(4 * kk + lk) * LDA + 16 * MB * wm + 16 * a + l15];
#pragma unroll
            for (int b = 0; b < NB; ++b) bf[b] = Bs[buf][(4 * kk + lk) * LDB + 16 * NB * wn + 16 * b + l15];
#pragma unroll
            for (int a = 0; a < MB; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[b], af[a], acc[a][b], 0, 0, 0);   // D(row = j, col = i)
        }
        // careful with the order: set_far was requested above and must not be overwritten before it is stored -- it is stored in the NEXT trip (as set_next there)
        if (k1 + 1 < nk) lstore(buf ^ 1, set_next);   // (the other LDS buffer: last read one trip ago, behind the barrier below)
        __syncthreads();
    };
    for (int kt = 0; kt < nk; kt += 2) {   // two trips per turn: the register sets swap roles by NAME (no dynamic register indexing)
        trip(kt, std::integral_constant<int, 0>{});
        if (kt + 1 < nk) trip(kt + 1, std::integral_constant<int, 1>{});
    }
    // acc[a][b][e]: column (lane & 15) -> i, row (lane >> 4) + 4 e -> j
#pragma unroll
    for (int a = 0; a < MB; ++a) {
        const long i = i0 + 16 * MB * wm + 16 * a + l15;
        if (i >= M) continue;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const long j = j0 + 16 * NB * wn + 16 * b + lk + 4 * e;
                if (j >= N) continue;
                const double v = acc[a][b][e];
                if (C64) C64[i + ldc * j] = v;
                if (C32) C32[i + ldc * j] = (float)v;
            }
    }
}

}  // namespace

nmfx_status gemm64(hipStream_t st, long M, long N, long Kc, const double *A64, const float *A32, long lda, const double *B64, const float *B32, long ldb,
                   double *C64, float *C32, long ldc) {
    if (M <= 0 || N <= 0) return NMFX_OK;
    if ((!A64 && !A32) || (!B64 && !B32) || (!C64 && !C32) || Kc <= 0) { set_error("gemm64: bad arguments"); return NMFX_ERR_INVALID; }
    const void *A = A64 ? static_cast<const void *>(A64) : static_cast<const void *>(A32);
    const void *B = B64 ? static_cast<const void *>(B64) : static_cast<const void *>(B32);
    dim3 grid((unsigned)((M + G64_BM - 1) / G64_BM), (unsigned)((N + G64_BN - 1) / G64_BN));
    if (A64 && B64) hipLaunchKernelGGL((gemm64_kernel<true, true>), grid, dim3(256), 0, st, A, lda, B, ldb, M, N, Kc, C64, C32, ldc);
    else if (A64) hipLaunchKernelGGL((gemm64_kernel<true, false>), grid, dim3(256), 0, st, A, lda, B, ldb, M, N, Kc, C64, C32, ldc);
    else if (B64) hipLaunchKernelGGL((gemm64_kernel<false, true>), grid, dim3(256), 0, st, A, lda, B, ldb, M, N, Kc, C64, C32, ldc);
    else hipLaunchKernelGGL((gemm64_kernel<false, false>), grid, dim3(256), 0, st, A, lda, B, ldb, M, N, Kc, C64, C32, ldc);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

}  // namespace nmfx

extern "C" nmfx_status nmfx_gemm64(void *stream, int64_t M, int64_t N, int64_t Kc, const double *A64, const float *A32, int64_t lda, const double *B64,
                                   const float *B32, int64_t ldb, double *C64, float *C32, int64_t ldc) {
    return nmfx::gemm64(static_cast<hipStream_t>(stream), M, N, Kc, A64, A32, lda, B64, B32, ldb, C64, C32, ldc);
}

// The small products of the euclidean Gram form with float64 accumulation, on the fp64 matrix core (v_mfma_f64_16x16x4_f64, gfx950):
//     P = W * (H*H')          (nmf.m:149-150 in the form of SURVEY A.2: V_hat*H' = W*(H*H'))
// The K-long fp32 accumulation chain of this product -- not the storage of W -- is what put H-fixed problems past the 1e-5 contract (round 4: 1.06e-5 ... 1.7e-5
// on W; the chain's rounding differs from iteration to iteration and the W update amplifies it; scripts/emu_precision.py).  m*K*K multiply-adds: a few
// per cent of the n-long contractions next to it, so it is simply done in double, with the float64 master copy of W as the left operand.
#include "nmfx_internal.h"

namespace nmfx {

namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int G64_BM = 64, G64_BN = 64, G64_BK = 16;
constexpr int G64_LD = 80;   // LDS row stride in doubles: 16 (mod 32), so the two k-rows a 32-lane group of ds_read_b64 touches fall on disjoint bank halves

// C(i, j) = sum_k A(i, k) * B(k, j);  A(i, k) = A[i + lda*k] (float64 or fp32), B(k, j) = B[k + ldb*j] (fp32 or float64), C[i + ldc*j] (float64 and / or fp32)
// 4 waves, each a 32 x 32 block of the 64 x 64 tile as 2 x 2 MFMA blocks.  The MFMA is fed transposed (first operand = B', second = A) so that the 16 lanes of
// a result register run along i, the contiguous dimension of C.
template <bool A64, bool B64>
__global__ __launch_bounds__(256) void gemm64_kernel(const void *__restrict__ Ap, long lda, const void *__restrict__ Bp, long ldb, long M, long N, long Kc,
                                                      double *__restrict__ C64, float *__restrict__ C32, long ldc) {
    __shared__ double As[2][G64_BK * G64_LD], Bs[2][G64_BK * G64_LD];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w & 1, wn = w >> 1;
    const long i0 = (long)blockIdx.x * G64_BM, j0 = (long)blockIdx.y * G64_BN;
    // loaders: A -- thread (row = tid & 63, k = 4*(tid >> 6) + u); B -- thread (col = tid >> 2, k = 4*(tid & 3) + u)
    const int a_row = tid & 63, a_k = 4 * (tid >> 6);
    const int b_col = tid >> 2, b_k = 4 * (tid & 3);
    const bool a_ok = i0 + a_row < M, b_ok = j0 + b_col < N;
    double ra[4], rb[4];
    auto gload = [&](long k0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long ka = k0 + a_k + u, kb = k0 + b_k + u;
            double va = 0.0, vb = 0.0;
            if (a_ok && ka < Kc) va = A64 ? static_cast<const double *>(Ap)[i0 + a_row + lda * ka] : (double)static_cast<const float *>(Ap)[i0 + a_row + lda * ka];
            if (b_ok && kb < Kc) vb = B64 ? static_cast<const double *>(Bp)[kb + ldb * (j0 + b_col)] : (double)static_cast<const float *>(Bp)[kb + ldb * (j0 + b_col)];
            ra[u] = va; rb[u] = vb;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            As[buf][(a_k + u) * G64_LD + a_row] = ra[u];
            Bs[buf][(b_k + u) * G64_LD + b_col] = rb[u];
        }
    };
    f64x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[a][b][e] = 0.0;
    const int nk = (int)((Kc + G64_BK - 1) / G64_BK);
    gload(0);
    lstore(0);
    __syncthreads();
    const int l15 = lane & 15, lk = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((long)(kt + 1) * G64_BK);
#pragma unroll
        for (int kk = 0; kk < G64_BK / 4; ++kk) {
            double af[2], bf[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) af[a] = As[buf][(4 * kk + lk) * G64_LD + 32 * wm + 16 * a + l15];
#pragma unroll
            for (int b = 0; b < 2; ++b) bf[b] = Bs[buf][(4 * kk + lk) * G64_LD + 32 * wn + 16 * b + l15];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[b], af[a], acc[a][b], 0, 0, 0);   // D(row = j, col = i)
        }
        if (kt + 1 < nk) lstore(buf ^ 1);   // (the other buffer: last read one iteration ago, behind the barrier below)
        __syncthreads();
    }
    // acc[a][b][e]: column (lane & 15) -> i, row (lane >> 4) + 4 e -> j
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const long i = i0 + 32 * wm + 16 * a + l15;
        if (i >= M) continue;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const long j = j0 + 32 * wn + 16 * b + lk + 4 * e;
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
    dim3 grid((unsigned)((M + G64_BM - 1) / G64_BM), (unsigned)((N + G64_BN - 1) / G64_BN));
    const void *A = A64 ? static_cast<const void *>(A64) : static_cast<const void *>(A32);
    const void *B = B64 ? static_cast<const void *>(B64) : static_cast<const void *>(B32);
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

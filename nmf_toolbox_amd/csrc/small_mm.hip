// Small products next to the big passes of a euclidean iteration (gfx950): the K x K Gram matrices of the Gram form
//     V_hat*H' = W*(H*H')   and   W'*V_hat = (W'*W)*H                         (nmf.m:150,181)
// and the H update with its denominator product folded in.  On the general GEMM these cost 30 us each for ~7 us of MFMA work (a
// 128 x 128 output is ONE tile of that kernel: all its parallelism is split-K through a two-k-tile-ahead LDS pipeline that never fills),
// plus slab-reduction launches; together with the H update they were 190 us of BASELINE config 2's 1.71 ms iteration.
//
//   h_update_gram    H(:, j) <- H(:, j) .* sum_z Gn_z(:, j) ./ max(G*H(:, j) + lambda, eps)   (nmf.m:181,199): G*H for 32 columns per wave on
//                    the MFMA (G from L2, H's columns as the other operand), the numerator slabs summed on the fly, H rewritten in place.
#include "gemm_common.h"

namespace nmfx {

namespace {

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int e = 0; e < 16; ++e) z[e] = 0.0f;
    return z;
}

// one workgroup = 32 columns of H; its NW = min(KB, 4) waves share them and split the KB = K / 32 blocks of 32 output rows k among themselves
// (wave w takes blocks w, w + NW, ...): D(j, k) = sum_l H(l, j) * G(l, k).  Many small waves per SIMD instead of one fat one: the loop is a
// chain of L2-latency loads (G) and the occupancy is what hides it.
template <int KB>
__global__ __launch_bounds__(64 * (KB < 4 ? KB : 4)) void h_update_gram_kernel(float *__restrict__ H, const float *__restrict__ G, const float *__restrict__ Gn, int n_slabs,
                                                                                  long slab_stride, long n, const float *__restrict__ lam, const uint8_t *__restrict__ fix,
                                                                                  double *__restrict__ H64) {
    constexpr int K = 32 * KB, NW = KB < 4 ? KB : 4, NB = (KB + NW - 1) / NW;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const long j0 = (long)blockIdx.x * 32;
    const long jl = j0 + l31 < n ? j0 + l31 : n - 1;          // clamped: columns past the end are computed and dropped
    const float *hp = H + (long)K * jl + 4 * h;
    f32x16 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = zero16();
    // first operand (registers <-> column j): four consecutive l of this lane's column as ONE 16-byte load; any pairing of contraction indices
    // is legal as long as both operands agree, so lane half h takes l = 8q + 4h + r in MFMA step r
    // second operand (lanes <-> k): G(k, l) = G[k + K*l], symmetric
#pragma unroll 4
    for (int q = 0; q < K / 8; ++q) {
        const float4 x = *reinterpret_cast<const float4 *>(hp + 8 * q);
        float y[4][NB];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int kb = wave + NW * b;
                y[r][b] = G[32 * (kb < KB ? kb : 0) + l31 + (long)K * (8 * q + 4 * h + r)];
            }
        const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[r], y[r][b], acc[b], 0, 0, 0);
    }
    __syncthreads();   // H is updated in place: every wave of the workgroup has read the workgroup's 32 columns before anyone overwrites a row block of them
    // acc[b][e] = Gp(k, j), k = 32 kb + l31 (lanes: contiguous in memory), j = j0 + (e & 3) + 8 (e >> 2) + 4 h
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int kb = wave + NW * b;
        if (kb >= KB) continue;
        const int k = 32 * kb + l31;
        if (fix && fix[k]) continue;
        const float lm = lam ? lam[k] : 0.0f;
        float neg[16], hv[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const long j = j0 + (e & 3) + 8 * (e >> 2) + 4 * h;
            const long idx = k + (long)K * (j < n ? j : n - 1);
            hv[e] = H[idx];
            neg[e] = Gn[idx];
            for (int sl = 1; sl < n_slabs; ++sl) neg[e] += Gn[idx + sl * slab_stride];   // split partial sums, fixed order (deterministic)
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const long j = j0 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (j >= n) continue;
            if (H64) {   // float64 master: the update in double, both arrays written
                const double hn = H64[k + (long)K * j] * ((double)neg[e] / fmax((double)acc[b][e] + (double)lm, 2.220446049250313e-16));
                H64[k + (long)K * j] = hn;
                H[k + (long)K * j] = (float)hn;
            } else H[k + (long)K * j] = hv[e] * (neg[e] / fmaxf(acc[b][e] + lm, NMFX_EPS_F));   // nmf.m:199
        }
    }
}

// The line-search objective of nmfsc's H search through the quadratic expansion (see aux.hip::quad_rows_kernel for the argument), on the K x n layout and
// with G*D on the MFMA exactly as in h_update_gram: D = Hc - H for the workgroup's 32 columns, acc = G*D, then
//     partials[block] = sum_{k, j} D(k, j) * (2*grad(k, j) + acc(k, j))          ( = 2*<grad, D> + <G*D, D> over the block's columns )
template <int KB>
__global__ __launch_bounds__(64 * (KB < 4 ? KB : 4)) void quad_cols_kernel(const float *__restrict__ H, const float *__restrict__ Hc, const float *__restrict__ grad,
                                                                              const float *__restrict__ G, long n, double *__restrict__ partials) {
    constexpr int K = 32 * KB, NW = KB < 4 ? KB : 4, NB = (KB + NW - 1) / NW;
    __shared__ double red[NW];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const long j0 = (long)blockIdx.x * 32;
    const long jl = j0 + l31 < n ? j0 + l31 : n - 1;          // clamped: columns past the end are computed and dropped
    const long ho = (long)K * jl + 4 * h;
    f32x16 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = zero16();
#pragma unroll 4
    for (int q = 0; q < K / 8; ++q) {
        const float4 xc = *reinterpret_cast<const float4 *>(Hc + ho + 8 * q), xb = *reinterpret_cast<const float4 *>(H + ho + 8 * q);
        float y[4][NB];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int kb = wave + NW * b;
                y[r][b] = G[32 * (kb < KB ? kb : 0) + l31 + (long)K * (8 * q + 4 * h + r)];
            }
        const float xs[4] = {xc.x - xb.x, xc.y - xb.y, xc.z - xb.z, xc.w - xb.w};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[r], y[r][b], acc[b], 0, 0, 0);
    }
    // acc[b][e] = (G*D)(k, j), k = 32 kb + l31, j = j0 + (e & 3) + 8 (e >> 2) + 4 h
    double t = 0.0;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int kb = wave + NW * b;
        if (kb >= KB) continue;
        const int k = 32 * kb + l31;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const long j = j0 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (j < n) {
                const long idx = k + (long)K * j;
                const float dd = Hc[idx] - H[idx];
                t += (double)dd * (2.0 * (double)grad[idx] + (double)acc[b][e]);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    if (lane == 0) red[wave] = t;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int w = 0; w < NW; ++w) s += red[w];
        partials[blockIdx.x] = s;
    }
}

}  // namespace

int quad_cols_blocks(long n) { return (int)((n + 31) / 32); }
nmfx_status quad_cols(hipStream_t st, const float *H, const float *Hc, const float *grad, const float *G, int K, long n, double *partials) {
    if (!h_update_gram_supported(K)) { set_error("quad_cols: K = %d", K); return NMFX_ERR_INVALID; }
    dim3 grid((unsigned)((n + 31) / 32));
#define NMFX_QC(KB) hipLaunchKernelGGL((quad_cols_kernel<KB>), grid, dim3(64 * (KB < 4 ? KB : 4)), 0, st, H, Hc, grad, G, n, partials)
    switch (K / 32) {
    case 1: NMFX_QC(1); break;
    case 2: NMFX_QC(2); break;
    case 3: NMFX_QC(3); break;
    case 4: NMFX_QC(4); break;
    case 5: NMFX_QC(5); break;
    case 6: NMFX_QC(6); break;
    case 7: NMFX_QC(7); break;
    default: NMFX_QC(8); break;
    }
#undef NMFX_QC
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

bool h_update_gram_supported(int K) { return K % 32 == 0 && K >= 32 && K <= 256; }

// H (K x n, in place) <- H .* (sum of n_slabs numerator slabs) ./ max(G*H + lambda, eps), G (K x K) symmetric    (nmf.m:181,199 in Gram form)
nmfx_status h_update_gram(hipStream_t st, float *H, const float *G, const float *Gn, int n_slabs, long slab_stride, int K, long n, const float *lam,
                          const uint8_t *fix, double *H64) {
    if (!h_update_gram_supported(K)) { set_error("h_update_gram: K = %d", K); return NMFX_ERR_INVALID; }
    if (n <= 0) return NMFX_OK;
    dim3 grid((unsigned)((n + 31) / 32));
#define NMFX_HUG(KB) hipLaunchKernelGGL((h_update_gram_kernel<KB>), grid, dim3(64 * (KB < 4 ? KB : 4)), 0, st, H, G, Gn, n_slabs, slab_stride, n, lam, fix, H64)
    switch (K / 32) {
    case 1: NMFX_HUG(1); break;
    case 2: NMFX_HUG(2); break;
    case 3: NMFX_HUG(3); break;
    case 4: NMFX_HUG(4); break;
    case 5: NMFX_HUG(5); break;
    case 6: NMFX_HUG(6); break;
    case 7: NMFX_HUG(7); break;
    default: NMFX_HUG(8); break;
    }
#undef NMFX_HUG
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

}  // namespace nmfx

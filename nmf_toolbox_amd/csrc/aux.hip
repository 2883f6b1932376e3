// Bandwidth-class kernels around the MFMA contractions: deterministic split-K reduction, fp64
// column/row reductions, the multiplicative-update epilogues (nmf.m:168-169,199; cnmf.m:193-199,231)
// and cost assembly (nmf.m:206-218).  All reductions accumulate in fp64 and are order-deterministic.
#include <algorithm>
#include "nmfx_internal.h"

namespace nmfx {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// block-wide sum of up to 16 waves; result valid in every thread
template <int NW>
__device__ __forceinline__ double block_sum(double v, double *red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[w];
    return s;
}

// ---- split-K slab reduction -------------------------------------------------------------------
__global__ void reduce_slabs_kernel(const float *slabs, int nslab, long stride, long count, float *out, int accumulate) {
    long idx = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (idx + 3 < count && ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(slabs)) & 15) == 0 && (stride & 3) == 0) {
        const float4 s0 = *reinterpret_cast<const float4 *>(slabs + idx);
        double sx = s0.x, sy = s0.y, sz = s0.z, sw = s0.w;   // the slabs are summed in double, in slab order (deterministic)
        for (int z = 1; z < nslab; ++z) {
            float4 t = *reinterpret_cast<const float4 *>(slabs + z * stride + idx);
            sx += t.x; sy += t.y; sz += t.z; sw += t.w;
        }
        if (accumulate) {
            float4 o = *reinterpret_cast<float4 *>(out + idx);
            sx += o.x; sy += o.y; sz += o.z; sw += o.w;
        }
        *reinterpret_cast<float4 *>(out + idx) = make_float4((float)sx, (float)sy, (float)sz, (float)sw);
    } else {
        for (long e = idx; e < idx + 4 && e < count; ++e) {
            double s = slabs[e];
            for (int z = 1; z < nslab; ++z) s += slabs[z * stride + e];
            if (accumulate) s += out[e];
            out[e] = (float)s;
        }
    }
}
// Many slabs of a small output (the K x K Gram products: 128 x 128 floats in up to 256 split-K slabs): one thread per output quad gave 16 workgroups to pull 16 MB
// through (31 us at C2), a two-level pair of launches 11 us (rounds 3-5).  ONE launch: 16 threads share an output quad and split the slabs among themselves (thread
// l takes slabs l, l + 16, ...: 16 consecutive threads read 256 contiguous bytes of a slab), LDS combines their 16 float64 partial sums in lane order.  count / 64
// workgroups.  The order of the sum is fixed by (nslab, count) alone: run-to-run deterministic.
__global__ __launch_bounds__(256) void reduce_slabs_lanes_kernel(const float *__restrict__ slabs, int nslab, long stride, long count, float *__restrict__ out, int accumulate) {
    __shared__ double part[16][16][4 + 1];
    const int qd = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const long idx = ((long)blockIdx.x * 16 + qd) * 4;
    const bool vec = idx + 3 < count && ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(slabs)) & 15) == 0 && (stride & 3) == 0;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    if (vec) {
#pragma unroll 4
        for (int z = sl; z < nslab; z += 16) {
            const float4 t = *reinterpret_cast<const float4 *>(slabs + z * stride + idx);
            s[0] += t.x; s[1] += t.y; s[2] += t.z; s[3] += t.w;
        }
    } else {
        for (int e = 0; e < 4; ++e)
            if (idx + e < count)
                for (int z = sl; z < nslab; z += 16) s[e] += slabs[z * stride + idx + e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) part[sl][qd][e] = s[e];
    __syncthreads();
    if (sl != 0) return;
#pragma unroll
    for (int l = 1; l < 16; ++l)
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += part[l][qd][e];
    if (vec) {
        if (accumulate) {
            const float4 o = *reinterpret_cast<const float4 *>(out + idx);
            s[0] += o.x; s[1] += o.y; s[2] += o.z; s[3] += o.w;
        }
        *reinterpret_cast<float4 *>(out + idx) = make_float4((float)s[0], (float)s[1], (float)s[2], (float)s[3]);
    } else {
        for (int e = 0; e < 4; ++e)
            if (idx + e < count) out[idx + e] = (float)(accumulate ? s[e] + out[idx + e] : s[e]);
    }
}
nmfx_status reduce_slabs(hipStream_t st, const float *slabs, int nslab, long slab_stride, long count, float *out, int accumulate) {
    if (count <= 0) return NMFX_OK;
    if (nslab >= 16 && count <= (1L << 18)) {
        hipLaunchKernelGGL(reduce_slabs_lanes_kernel, dim3((unsigned)((count + 63) / 64)), dim3(256), 0, st, slabs, nslab, slab_stride, count, out, accumulate);
        NMFX_HIP(hipGetLastError());
        return NMFX_OK;
    }
    long nthr = (count + 3) / 4;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, st, slabs, nslab, slab_stride, count, out, accumulate);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// ---- cnmf H-step numerator from Q = W_flat' * X (KT x ncols):  Gn(k, j) = sum_t Q((t,k), j + t), j + t < nvalid   (cnmf.m:217-226) ----
__global__ void shift_sum_kernel(const float *Q, int K, int T, long n, long nvalid, float *Gn) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)K * n) return;
    const long j = idx / K;
    const int k = (int)(idx - j * K);
    const long KT = (long)K * T;
    float s = 0.0f;
    for (int t = 0; t < T; ++t)
        if (j + t < nvalid) s += Q[(long)t * K + k + KT * (j + t)];
    Gn[idx] = s;
}
nmfx_status shift_sum(hipStream_t st, const float *Q, int K, int T, long n, long nvalid, float *Gn) {
    const long count = (long)K * n;
    if (count <= 0) return NMFX_OK;
    hipLaunchKernelGGL(shift_sum_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, Q, K, T, n, nvalid, Gn);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}
// dst = [zeros(K, pad) | src (K x n) | zeros(K, pad_right)]: the zero left halo the cnmf fused passes read for columns j - t < 0; the
// right one lets the lag-form Gram products below read H(:, j + d) past the last column
__global__ void pad_left_kernel(const float *src, long count, long padcount, long rightcount, float *dst) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < padcount) dst[idx] = 0.0f;
    if (idx < count) dst[padcount + idx] = src[idx];
    if (idx < rightcount) dst[padcount + count + idx] = 0.0f;
}
nmfx_status pad_left(hipStream_t st, const float *src, int K, long n, int pad, float *dst, int pad_right) {
    const long count = (long)K * n, padcount = (long)K * pad, rightcount = (long)K * pad_right;
    const long tot = std::max(count, std::max(padcount, rightcount));
    hipLaunchKernelGGL(pad_left_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, src, count, padcount, rightcount, dst);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// ---- cnmf Gram products by lag (euclidean, unsharded).  The KT x KT Gram of the stacked shifted H (cnmf.m:191-192 with V_hat*Hs' =
// W_flat*(Hs*Hs')) has only T distinct K x K blocks up to boundary terms:
//   G[(t1,k1),(t2,k2)] = sum_{j >= max(t1,t2)} H(k1, j-t1) H(k2, j-t2) = L_d(k1,k2) - sum_{u = n-t1}^{n-1-d} H(k1,u) H(k2,u+d),   d = t1-t2 >= 0
// with the lag Grams L_d = sum_{u=0}^{n-1-d} H(:,u) H(:,u+d)' (one K x T*K GEMM over the zero-padded H instead of a KT x KT one: T times
// fewer flops); d < 0 by symmetry.  L is stored as L[k1 + K*((T-1-d)*K + k2)].
__global__ void gram_from_lags_kernel(const float *L, const float *H, int K, int T, long n, float *G) {
    const long KT = (long)K * T;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= KT * KT) return;
    int a = (int)(idx % KT), b = (int)(idx / KT);
    if (a / K < b / K) { const int tmp = a; a = b; b = tmp; }   // symmetric: evaluate the (t1 >= t2) twin
    const int t1 = a / K, k1 = a - t1 * K, t2 = b / K, k2 = b - t2 * K, d = t1 - t2;
    float corr = 0.0f;
    for (long u = n - t1; u <= n - 1 - d; ++u) corr += H[k1 + (long)K * u] * H[k2 + (long)K * (u + d)];
    G[idx] = L[k1 + (long)K * ((long)(T - 1 - d) * K + k2)] - corr;
}
nmfx_status gram_from_lags(hipStream_t st, const float *L, const float *H, int K, int T, long n, float *G) {
    const long cnt = (long)K * T * K * T;
    hipLaunchKernelGGL(gram_from_lags_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, L, H, K, T, n, G);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}
// H-step denominator sum_t W_t' * lshift_t(V_hat) = sum_{t,t'} CC[(t,.),(t',.)] * H(:, j+t-t') (cnmf.m:217-226 without V_hat, CC = W_flat'*W_flat):
// away from the last T-1 columns the (t, t') terms depend on d = t-t' only, so E_d = sum_{t-t'=d} CC_(t,t') (2T-1 blocks of K x K) and one
// K x n GEMM with contraction (2T-1)*K over the zero-padded H replace the T^2-block contraction.  E[k + K*(p*K + k')], p = T-1-d.
__global__ void lag_sum_kernel(const float *CC, int K, int T, float *E) {
    const long KT = (long)K * T;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)K * (2 * T - 1) * K) return;
    const int k = (int)(idx % K);
    const long c = idx / K;
    const int p = (int)(c / K), kp = (int)(c - (long)p * K), d = T - 1 - p;
    float s = 0.0f;
    for (int tp = 0; tp < T; ++tp) {
        const int t = tp + d;
        if (t >= 0 && t < T) s += CC[(long)t * K + k + KT * ((long)tp * K + kp)];
    }
    E[idx] = s;
}
nmfx_status lag_sum(hipStream_t st, const float *CC, int K, int T, float *E) {
    const long cnt = (long)K * (2 * T - 1) * K;
    hipLaunchKernelGGL(lag_sum_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, CC, K, T, E);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}
// ... and the last T-1 columns, where lshift_t (cnmf.m:219) drops the terms with j + t > n - 1, evaluated term by term:
//   Gp(k, j) = sum_{t <= n-1-j} sum_{t'} sum_{k'} CC[(t,k),(t',k')] * H(k', j+t-t')     (j+t-t' >= 0)
__global__ __launch_bounds__(256) void gp_tail_kernel(const float *CC, const float *H, int K, int T, long n, float *Gp) {
    // one workgroup per output (k, j); CC is symmetric, so column (t,k) of it is read contiguously along (t',k')
    __shared__ float red[4];
    const int KT = K * T;
    const long j = n - (T - 1) + blockIdx.x / K;
    const int k = blockIdx.x % K;
    const int tmax = (int)(n - 1 - j);
    float s = 0.0f;
    for (int t = 0; t <= tmax; ++t) {
        const float *cc = CC + (long)KT * ((long)t * K + k);
        for (int c = threadIdx.x; c < KT; c += 256) {
            const int tp = c / K, kp = c - tp * K;
            const long col = j + t - tp;
            if (col >= 0) s += cc[c] * H[kp + (long)K * col];
        }
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) Gp[k + (long)K * j] = red[0] + red[1] + red[2] + red[3];
}
nmfx_status gp_tail(hipStream_t st, const float *CC, const float *H, int K, int T, long n, float *Gp) {
    if (T < 2) return NMFX_OK;
    hipLaunchKernelGGL(gp_tail_kernel, dim3((unsigned)((T - 1) * K)), dim3(256), 0, st, CC, H, K, T, n, Gp);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// ---- single-process multi-GPU all-reduce, reduce-scatter half: this device sums slice [off, off + count) of every device's buffer
// (peer-mapped over xGMI) in the fixed order 0 .. ndev-1 and writes it back into its own buffer; the all-gather half is plain peer copies.
// One device owns each slice, so every device ends with bit-identical sums.
__global__ void peer_reduce_kernel(PeerPtrs bufs, int ndev, int self, long off, long count) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= count) return;
    if (i + 3 < count && ((off + i) & 3) == 0) {
        float4 s = *reinterpret_cast<const float4 *>(bufs.p[0] + off + i);
        for (int h = 1; h < ndev; ++h) {
            const float4 t = *reinterpret_cast<const float4 *>(bufs.p[h] + off + i);
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        *reinterpret_cast<float4 *>(bufs.p[self] + off + i) = s;
    } else {
        for (long e = i; e < i + 4 && e < count; ++e) {
            float s = bufs.p[0][off + e];
            for (int h = 1; h < ndev; ++h) s += bufs.p[h][off + e];
            bufs.p[self][off + e] = s;
        }
    }
}
nmfx_status peer_reduce(hipStream_t st, const PeerPtrs &bufs, int ndev, int self, long off, long count) {
    if (count <= 0) return NMFX_OK;
    const long nthr = (count + 3) / 4;
    hipLaunchKernelGGL(peer_reduce_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, st, bufs, ndev, self, off, count);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// ---- column reductions (one workgroup per column) ----------------------------------------------
__device__ __forceinline__ double red_f(int mode, float x) {
    return mode == 1 ? (double)x * (double)x : (mode == 2 ? (double)fabsf(x) : (double)x);
}
__global__ __launch_bounds__(256) void col_reduce_kernel(const float *X, long rows, long ld, int mode, double *out) {
    __shared__ double red[4];
    const float *x = X + ld * blockIdx.x;
    double s = 0.0;
    if ((rows & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        const float4 *x4 = reinterpret_cast<const float4 *>(x);
        for (long i = threadIdx.x; i < rows / 4; i += 256) {
            const float4 v = x4[i];
            s += (red_f(mode, v.x) + red_f(mode, v.y)) + (red_f(mode, v.z) + red_f(mode, v.w));
        }
    } else {
        for (long i = threadIdx.x; i < rows; i += 256) s += red_f(mode, x[i]);
    }
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}
nmfx_status col_reduce(hipStream_t st, const float *X, long rows, long ld, int ncols, int mode, double *out) {
    if (ncols <= 0) return NMFX_OK;
    hipLaunchKernelGGL(col_reduce_kernel, dim3(ncols), dim3(256), 0, st, X, rows, ld, mode, out);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// the same over float64 columns (the master copy of W): mode 0 sum, 1 sum of squares, 2 sum of |x|
__global__ __launch_bounds__(256) void col_reduce64_kernel(const double *X, long rows, long ld, int mode, double *out) {
    __shared__ double red[4];
    const double *x = X + ld * blockIdx.x;
    double s = 0.0;
    for (long i = threadIdx.x; i < rows; i += 256) { const double v = x[i]; s += mode == 0 ? v : (mode == 1 ? v * v : fabs(v)); }
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}
nmfx_status col_reduce64(hipStream_t st, const double *X, long rows, long ld, int ncols, int mode, double *out) {
    if (ncols <= 0) return NMFX_OK;
    hipLaunchKernelGGL(col_reduce64_kernel, dim3(ncols), dim3(256), 0, st, X, rows, ld, mode, out);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// out[c] = sum_i X[i + ld*c].^e in fp64 (alpha-beta cost constant sum(V.^(alpha+beta)), nmf.m:214); MATLAB power semantics for e in {0, 1}
__global__ __launch_bounds__(256) void col_reduce_pow_kernel(const float *X, long rows, long ld, float e, double *out) {
    __shared__ double red[4];
    const float *x = X + ld * blockIdx.x;
    double s = 0.0;
    for (long i = threadIdx.x; i < rows; i += 256) s += (double)(e == 0.0f ? 1.0f : (e == 1.0f ? x[i] : powf(x[i], e)));
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}
nmfx_status col_reduce_pow(hipStream_t st, const float *X, long rows, long ld, int ncols, float e, double *out) {
    if (ncols <= 0) return NMFX_OK;
    hipLaunchKernelGGL(col_reduce_pow_kernel, dim3(ncols), dim3(256), 0, st, X, rows, ld, e, out);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}
// out = in.^e
__global__ void pow_map_kernel(const float *in, float *out, long count, float e) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < count) out[idx] = powf(in[idx], e);
}
nmfx_status pow_map(hipStream_t st, const float *in, float *out, long count, float e) {
    if (count <= 0) return NMFX_OK;
    hipLaunchKernelGGL(pow_map_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, in, out, count, e);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// ---- row reductions: out[k] = sum_j f(X[k + ld*j]); two deterministic stages ---------------------
constexpr int RR_BLOCKS = 256;
__global__ __launch_bounds__(256) void row_reduce_stage1(const float *X, int rows, long ld, long ncols, int mode, double *part) {
    __shared__ double red[4][64];
    const int kk = threadIdx.x & 63, jj = threadIdx.x >> 6;
    const long per = (ncols + gridDim.x - 1) / gridDim.x;
    const long j0 = per * blockIdx.x, j1 = (j0 + per < ncols) ? j0 + per : ncols;
    for (int kb = 0; kb < rows; kb += 64) {
        const int k = kb + kk;
        double s = 0.0;
        if (k < rows)
            for (long j = j0 + jj; j < j1; j += 4) s += red_f(mode, X[k + ld * j]);
        __syncthreads();
        red[jj][kk] = s;
        __syncthreads();
        if (jj == 0 && k < rows) part[(long)blockIdx.x * rows + k] = red[0][kk] + red[1][kk] + red[2][kk] + red[3][kk];
    }
}
// same reduction for rows % 4 == 0 and 16-byte aligned columns: float4 along the rows, 16 column lanes, four loads in flight
__global__ __launch_bounds__(256) void row_reduce_stage1_v4(const float *X, int rows, long ld, long ncols, int mode, double *part) {
    __shared__ double red[16][65];
    const int kq = threadIdx.x & 15, jj = threadIdx.x >> 4;       // 16 float4 = 64 rows, 16 column lanes
    const long per = (ncols + gridDim.x - 1) / gridDim.x;
    const long j0 = per * blockIdx.x, j1 = (j0 + per < ncols) ? j0 + per : ncols;
    for (int kb = 0; kb < rows; kb += 64) {
        const int k = kb + 4 * kq;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        if (k < rows) {
            long j = j0 + jj;
            for (; j + 48 < j1; j += 64) {
                const float4 a = *reinterpret_cast<const float4 *>(X + k + ld * j), b = *reinterpret_cast<const float4 *>(X + k + ld * (j + 16));
                const float4 c = *reinterpret_cast<const float4 *>(X + k + ld * (j + 32)), d = *reinterpret_cast<const float4 *>(X + k + ld * (j + 48));
                s0 += (red_f(mode, a.x) + red_f(mode, b.x)) + (red_f(mode, c.x) + red_f(mode, d.x));
                s1 += (red_f(mode, a.y) + red_f(mode, b.y)) + (red_f(mode, c.y) + red_f(mode, d.y));
                s2 += (red_f(mode, a.z) + red_f(mode, b.z)) + (red_f(mode, c.z) + red_f(mode, d.z));
                s3 += (red_f(mode, a.w) + red_f(mode, b.w)) + (red_f(mode, c.w) + red_f(mode, d.w));
            }
            for (; j < j1; j += 16) {
                const float4 a = *reinterpret_cast<const float4 *>(X + k + ld * j);
                s0 += red_f(mode, a.x); s1 += red_f(mode, a.y); s2 += red_f(mode, a.z); s3 += red_f(mode, a.w);
            }
        }
        __syncthreads();
        red[jj][4 * kq + 0] = s0; red[jj][4 * kq + 1] = s1; red[jj][4 * kq + 2] = s2; red[jj][4 * kq + 3] = s3;
        __syncthreads();
        if (threadIdx.x < 64 && kb + (int)threadIdx.x < rows) {
            double t = 0.0;
#pragma unroll
            for (int u = 0; u < 16; ++u) t += red[u][threadIdx.x];
            part[(long)blockIdx.x * rows + kb + threadIdx.x] = t;
        }
    }
}
__global__ __launch_bounds__(64) void row_reduce_stage2(const double *part, int rows, int nblk, double *out) {
    const int k = blockIdx.x;   // one wave per row; fixed summation order (deterministic)
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 64) s += part[(long)b * rows + k];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[k] = s;
}
nmfx_status row_reduce(hipStream_t st, const float *X, int rows, long ld, long ncols, int mode, double *out, void *scratch) {
    if (rows <= 0) return NMFX_OK;
    double *part = static_cast<double *>(scratch);  // RR_BLOCKS * rows doubles
    if ((rows & 3) == 0 && (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0)
        hipLaunchKernelGGL(row_reduce_stage1_v4, dim3(RR_BLOCKS), dim3(256), 0, st, X, rows, ld, ncols, mode, part);
    else
        hipLaunchKernelGGL(row_reduce_stage1, dim3(RR_BLOCKS), dim3(256), 0, st, X, rows, ld, ncols, mode, part);
    hipLaunchKernelGGL(row_reduce_stage2, dim3(rows), dim3(64), 0, st, part, rows, RR_BLOCKS, out);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}
size_t row_reduce_scratch_bytes(int rows) { return sizeof(double) * (size_t)RR_BLOCKS * rows; }

// ---- W update: one workgroup per column c = k + K*t -------------------------------------------
//   dn = sum_i W.*P  (= diag(H*B'*W)),  dp = sum_i W.*N
//   W <- W .* ((N + W*dn).^e ./ max((P + W*dp).^e + lambda, eps))          nmf.m:168 / cnmf.m:193
// the same update on the float64 master copy of W: every sweep in double, both arrays written (the fp32 one is what the MFMA passes contract)
// NW waves per workgroup (BT threads): one workgroup per column; 16 waves where the columns are long and few (C2: 128 columns of 8192 rows -- a sweep is one trip to
// HBM instead of two, and a launch is three dependent sweeps)
template <int NW>
__device__ void w_update64_body(const WUpdateParams &p, double *red) {
    constexpr int BT = 64 * NW;
    const int c = blockIdx.x, k = c % p.K;
    double *w = p.W64 + p.m * c;
    float *w32 = p.W + p.m * c;
    const int nch = p.n_chunks > 1 ? p.n_chunks : 1;
    const long cr = p.m / nch, KT = (long)p.K * p.T;
    const float *pp = p.P ? p.P + p.m * c : nullptr;
    const double *pp64 = p.P64 ? p.P64 + p.m * c : nullptr;
    const double pv = p.Pvec ? p.Pvec[c] : (p.Pvecf ? (double)p.Pvecf[c] : 0.0);
    const float *ncol = p.N + p.m * c;   // (one chunk: column c of N is contiguous -- and the 64-bit division below is most of a sweep's instructions)
    auto nat = [&](long i) { if (nch == 1) return (double)ncol[i]; const long ch = i / cr; return (double)p.N[ch * cr * KT + cr * c + (i - ch * cr)]; };
    auto pat = [&](long i) { return pp64 ? pp64[i] : (pp ? (double)pp[i] : pv); };
    const bool fixed = p.fixW && p.fixW[k];
    const bool plain = p.rule == 1;   // lnmf.m:69
    double dn = 0.0, dp = 0.0;
    if (p.stats_in) { dn = p.dndp[c]; dp = p.dndp[KT + c]; }
    else if (p.dndp || (!fixed && !plain)) {
        // (every sweep: four elements per thread and trip, all loads issued before the first use -- the same elements in the same order as one per trip, but four
        // loads in flight per array instead of one: a column is m / BT dependent trips to HBM otherwise, and the update was 0.11 ms at C3)
        for (long i0 = threadIdx.x; i0 < p.m; i0 += (4 * BT)) {
            double wv[4], nv[4], qv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const long i = i0 + BT * u; const bool ok = i < p.m; wv[u] = ok ? w[i] : 0.0; nv[u] = ok ? nat(i) : 0.0; qv[u] = ok ? pat(i) : 0.0; }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (i0 + BT * u < p.m) { dn = fma(wv[u], qv[u], dn); dp = fma(wv[u], nv[u], dp); }
        }
        dn = block_sum<NW>(dn, red);
        dp = block_sum<NW>(dp, red);
        if (p.dndp && threadIdx.x == 0) { p.dndp[c] = dn; p.dndp[KT + c] = dp; }
    }
    if (p.stats_only) return;
    if (fixed) {
        if (p.fuse_norm != 0 && p.colsum_out) {   // fixed column: untouched, but its sum is still part of the H-step denominator
            double cs = 0.0;
            for (long i = threadIdx.x; i < p.m; i += BT) cs += w[i];
            cs = block_sum<NW>(cs, red);
            if (threadIdx.x == 0) p.colsum_out[c] = cs;
        }
        return;
    }
    const double lam = p.lamW ? (double)p.lamW[k] : 0.0, eps = 2.220446049250313e-16, ie = (double)p.inv_exp;
    double ss = 0.0;
    for (long i0 = threadIdx.x; i0 < p.m; i0 += (4 * BT)) {
        double wv[4], nv[4], qv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const long i = i0 + BT * u; const bool ok = i < p.m; wv[u] = ok ? w[i] : 0.0; nv[u] = ok ? nat(i) : 0.0; qv[u] = ok ? pat(i) : 0.0; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long i = i0 + BT * u;
            if (i >= p.m) continue;
            const double wi = wv[u], ni = nv[u], pi = qv[u];
            double neg = plain ? ni : fma(wi, dn, ni);
            double pos = plain ? pi : fma(wi, dp, pi);
            if (p.inv_exp != 1.0f) { neg = pow(neg, ie); pos = pow(pos, ie); }
            const double wn = wi * (neg / fmax(pos + lam, eps));   // nmf.m:168 / cnmf.m:193
            w[i] = wn;
            if (p.fuse_norm == 0) w32[i] = (float)wn;
            ss += plain ? wn : wn * wn;
        }
    }
    ss = block_sum<NW>(ss, red);
    if (threadIdx.x == 0) p.sumsq[c] = ss;
    if (p.fuse_norm == 0) return;
    const double f = p.fuse_norm == 2 ? 1.0 / ss : 1.0 / sqrt(ss);   // nmf.m:169 / lnmf.m:70
    double cs = 0.0;
    for (long i0 = threadIdx.x; i0 < p.m; i0 += (4 * BT)) {
        double wv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const long i = i0 + BT * u; wv[u] = i < p.m ? w[i] : 0.0; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const long i = i0 + BT * u; if (i < p.m) { const double v = wv[u] * f; w[i] = v; w32[i] = (float)v; cs += v; } }
    }
    if (p.colsum_out) {
        cs = block_sum<NW>(cs, red);
        if (threadIdx.x == 0) p.colsum_out[c] = cs;
    }
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void w_update_kernel(const WUpdateParams p) {
    constexpr int BT = 64 * NW;
    __shared__ double red[NW];
    const int c = blockIdx.x;
    const int k = c % p.K;
    float *w = p.W + p.m * c;
    if (p.fin_on && blockIdx.x == gridDim.x - 1) {   // the cost of the state this W step started from (gram_cost_finish_kernel, same arithmetic and order)
        const bool exact = *p.fin_exact_flag != 0;
        double s = 0.0, t = 0.0;
        if (exact) { for (int i = threadIdx.x; i < p.fin_nparts; i += BT) s += p.fin_partials[i]; }
        else if (p.fin_rank0) { for (int cc = threadIdx.x; cc < p.fin_nc; cc += BT) s += 0.5 * p.dndp[cc] - p.dndp[p.fin_nc + cc]; }
        s = block_sum<NW>(s, red);
        if (p.fin_l1W) for (int cc = threadIdx.x; cc < p.fin_nW; cc += BT) t += (double)p.fin_lamW[cc % p.fin_K] * p.fin_l1W[cc];
        if (p.fin_l1H) for (int kk = threadIdx.x; kk < p.fin_K; kk += BT) t += (double)p.fin_lamH[kk] * p.fin_l1H[kk];
        t = block_sum<NW>(t, red);
        if (threadIdx.x == 0) {
            const double cst = (exact ? 0.5 * s : 0.5 * p.fin_sumVV[0] + s) + t;
            *p.fin_out = cst;
            if (p.fin_out2) *p.fin_out2 = cst;
        }
    }
    if (p.W64) { w_update64_body<NW>(p, red); return; }
    if (p.fixW && p.fixW[k]) {
        if (p.dndp && !p.stats_in) {   // fixed column: untouched, but <W, N> and <W, P> of the Gram-form cost run over every column
            const float *pp = p.P ? p.P + p.m * c : nullptr;
            const float pv = p.Pvec ? (float)p.Pvec[c] : (p.Pvecf ? p.Pvecf[c] : 0.0f);
            const int nch = p.n_chunks > 1 ? p.n_chunks : 1;
            const long cr = p.m / nch, KT = (long)p.K * p.T;
            double dn = 0.0, dp = 0.0;
            for (long i = threadIdx.x; i < p.m; i += BT) {
                const long ch = i / cr;
                const double wi = (double)w[i];
                dn += wi * (double)(pp ? pp[i] : pv);
                dp += wi * (double)p.N[ch * cr * KT + cr * c + (i - ch * cr)];
            }
            dn = block_sum<NW>(dn, red);
            dp = block_sum<NW>(dp, red);
            if (threadIdx.x == 0) { p.dndp[c] = dn; p.dndp[KT + c] = dp; }
        }
        if (p.stats_only) return;
        if (p.fuse_norm != 0 && p.colsum_out) {   // fixed column: untouched, but its sum is still part of the H-step denominator
            double cs = 0.0;
            for (long i = threadIdx.x; i < p.m; i += BT) cs += (double)w[i];
            cs = block_sum<NW>(cs, red);
            if (threadIdx.x == 0) p.colsum_out[c] = cs;
        }
        return;
    }
    // numerator column c: plain m x KT layout, or n_chunks contiguous (cr x KT) row blocks (chunk ch holds rows [ch*cr, (ch+1)*cr))
    const int nch = p.n_chunks > 1 ? p.n_chunks : 1;
    const long cr = p.m / nch;
    const long KT = (long)p.K * p.T;
    const float *pp = p.P ? p.P + p.m * c : nullptr;
    const float pv = p.Pvec ? (float)p.Pvec[c] : (p.Pvecf ? p.Pvecf[c] : 0.0f);
    double dn = 0.0, dp = 0.0;
    const float lam = p.lamW ? p.lamW[k] : 0.0f;
    const bool vec = (cr & 3) == 0 && ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(p.N) | reinterpret_cast<uintptr_t>(pp)) & 15) == 0;
    const bool plain = p.rule == 1;   // lnmf.m:69: W .* (N ./ max(P, eps)), no diagonal terms; the column statistic is the L1 sum
    auto upd = [&](float wi, float ni, float pi, float fdn, float fdp) {
        float neg = plain ? ni : fmaf(wi, fdn, ni);
        float pos = plain ? pi : fmaf(wi, fdp, pi);
        if (p.inv_exp != 1.0f) { neg = powf(neg, p.inv_exp); pos = powf(pos, p.inv_exp); }
        return wi * (neg / fmaxf(pos + lam, NMFX_EPS_F));
    };
    double ss = 0.0;
    if (vec) {
        const long c4n = cr / 4;
        const float4 pvv = make_float4(pv, pv, pv, pv);
        if (p.stats_in) { dn = p.dndp[c]; dp = p.dndp[KT + c]; }
        else {
        for (int ch = 0; ch < nch; ++ch) {
            const float4 *w4 = reinterpret_cast<const float4 *>(w + ch * cr), *n4 = reinterpret_cast<const float4 *>(p.N + ch * cr * KT + cr * c);
            const float4 *p4 = reinterpret_cast<const float4 *>(pp ? pp + ch * cr : nullptr);
            for (long i = threadIdx.x; i < c4n; i += BT) {
                const float4 a = w4[i], b = n4[i], c4 = pp ? p4[i] : pvv;
                dn += ((double)a.x * c4.x + (double)a.y * c4.y) + ((double)a.z * c4.z + (double)a.w * c4.w);
                dp += ((double)a.x * b.x + (double)a.y * b.y) + ((double)a.z * b.z + (double)a.w * b.w);
            }
        }
        dn = block_sum<NW>(dn, red);
        dp = block_sum<NW>(dp, red);
        if (p.dndp && threadIdx.x == 0) { p.dndp[c] = dn; p.dndp[KT + c] = dp; }
        if (p.stats_only) return;
        }
        const float fdn = (float)dn, fdp = (float)dp;
        for (int ch = 0; ch < nch; ++ch) {
            float4 *w4 = reinterpret_cast<float4 *>(w + ch * cr);
            const float4 *n4 = reinterpret_cast<const float4 *>(p.N + ch * cr * KT + cr * c);
            const float4 *p4 = reinterpret_cast<const float4 *>(pp ? pp + ch * cr : nullptr);
            for (long i = threadIdx.x; i < c4n; i += BT) {
                const float4 a = w4[i], b = n4[i], c4 = pp ? p4[i] : pvv;
                float4 o;
                o.x = upd(a.x, b.x, c4.x, fdn, fdp); o.y = upd(a.y, b.y, c4.y, fdn, fdp);
                o.z = upd(a.z, b.z, c4.z, fdn, fdp); o.w = upd(a.w, b.w, c4.w, fdn, fdp);
                w4[i] = o;
                ss += plain ? ((double)o.x + o.y) + ((double)o.z + o.w) : ((double)o.x * o.x + (double)o.y * o.y) + ((double)o.z * o.z + (double)o.w * o.w);
            }
        }
    } else {
        auto nat = [&](long i) { const long ch = i / cr; return p.N[ch * cr * KT + cr * c + (i - ch * cr)]; };
        if (p.stats_in) { dn = p.dndp[c]; dp = p.dndp[KT + c]; }
        else {
        for (long i = threadIdx.x; i < p.m; i += BT) {
            const float wi = w[i];
            dn += (double)wi * (double)(pp ? pp[i] : pv);
            dp += (double)wi * (double)nat(i);
        }
        dn = block_sum<NW>(dn, red);
        dp = block_sum<NW>(dp, red);
        if (p.dndp && threadIdx.x == 0) { p.dndp[c] = dn; p.dndp[KT + c] = dp; }
        if (p.stats_only) return;
        }
        const float fdn = (float)dn, fdp = (float)dp;
        for (long i = threadIdx.x; i < p.m; i += BT) {
            const float wn = upd(w[i], nat(i), pp ? pp[i] : pv, fdn, fdp);
            w[i] = wn;
            ss += plain ? (double)wn : (double)wn * (double)wn;
        }
    }
    ss = block_sum<NW>(ss, red);
    if (threadIdx.x == 0) p.sumsq[c] = ss;
    if (p.fuse_norm == 0) return;
    // nmf.m:169 / lnmf.m:70 on the column just written: every thread re-reads exactly the elements it stored (same factor expression as
    // w_normalize_kernel, so W is bit-identical to the two-launch sequence)
    const float f = (float)(p.fuse_norm == 2 ? 1.0 / ss : 1.0 / sqrt(ss));
    double cs = 0.0;
    if (vec) {
        for (int ch = 0; ch < nch; ++ch) {
            float4 *w4 = reinterpret_cast<float4 *>(w + ch * cr);
            for (long i = threadIdx.x; i < cr / 4; i += BT) {
                float4 v = w4[i];
                v.x *= f; v.y *= f; v.z *= f; v.w *= f;
                w4[i] = v;
                cs += ((double)v.x + v.y) + ((double)v.z + v.w);
            }
        }
    } else {
        for (long i = threadIdx.x; i < p.m; i += BT) { const float v = w[i] * f; w[i] = v; cs += (double)v; }
    }
    if (p.colsum_out) {
        cs = block_sum<NW>(cs, red);
        if (threadIdx.x == 0) p.colsum_out[c] = cs;
    }
}
nmfx_status w_update(hipStream_t st, const WUpdateParams &p) {
    if (p.m >= 4096 && p.K * p.T <= 256) hipLaunchKernelGGL(w_update_kernel<16>, dim3(p.K * p.T), dim3(1024), 0, st, p);
    else hipLaunchKernelGGL(w_update_kernel<4>, dim3(p.K * p.T), dim3(256), 0, st, p);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// ---- euclidean cost in Gram form -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gram_decide_kernel(const double *dndp, int nc, const double *sumVV, double ratio_min, int *exact_flag, int *host_slot, int stamp) {
    __shared__ double red[4];
    double sn = 0.0, sp = 0.0;
    for (int c = threadIdx.x; c < nc; c += 256) { sn += dndp[c]; sp += dndp[nc + c]; }
    sn = block_sum<4>(sn, red);
    sp = block_sum<4>(sp, red);
    if (threadIdx.x == 0) {
        int flag = *exact_flag;
        if (flag == 0) {
            const double half_vv = 0.5 * sumVV[1], data = half_vv - sp + 0.5 * sn;
            if (!(data >= ratio_min * half_vv)) { flag = 1; *exact_flag = 1; }   // (NaN lands here too)
        }
        // the state of the flag AFTER decision number `stamp`, for the host: it latches on the decision of a fixed, earlier W update (engine.hip::gram_active),
        // never on "whatever the flag happens to be when the host looks"
        if (host_slot) __hip_atomic_store(host_slot, (stamp << 1) | flag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
nmfx_status gram_decide(hipStream_t st, const double *dndp, int nc, const double *sumVV, double ratio_min, int *exact_flag, int *host_slot, int stamp) {
    hipLaunchKernelGGL(gram_decide_kernel, dim3(1), dim3(256), 0, st, dndp, nc, sumVV, ratio_min, exact_flag, host_slot, stamp);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}
__global__ __launch_bounds__(256) void gram_cost_finish_kernel(const double *dndp, int nc, const double *sumVV, int rank0, const int *exact_flag, const double *partials,
                                                               int nparts, const double *l1W, int nW, const float *lamW, const double *l1H, int K, const float *lamH,
                                                               double *out, double *out2) {
    __shared__ double red[4];
    const bool exact = *exact_flag != 0;
    double s = 0.0, t = 0.0;
    if (exact) { for (int i = threadIdx.x; i < nparts; i += 256) s += partials[i]; }
    else if (rank0) { for (int c = threadIdx.x; c < nc; c += 256) s += 0.5 * dndp[c] - dndp[nc + c]; }
    s = block_sum<4>(s, red);
    if (l1W) for (int c = threadIdx.x; c < nW; c += 256) t += (double)lamW[c % K] * l1W[c];
    if (l1H) for (int k = threadIdx.x; k < K; k += 256) t += (double)lamH[k] * l1H[k];
    t = block_sum<4>(t, red);
    if (threadIdx.x == 0) {
        const double cst = (exact ? 0.5 * s : 0.5 * sumVV[0] + s) + t;
        *out = cst;
        if (out2) *out2 = cst;
    }
}
nmfx_status gram_cost_finish(hipStream_t st, const double *dndp, int nc, const double *sumVV, int rank0, const int *exact_flag, const double *partials, int nparts,
                             const double *l1W, int nW, const float *lamW, const double *l1H, int K, const float *lamH, double *out, double *out2) {
    hipLaunchKernelGGL(gram_cost_finish_kernel, dim3(1), dim3(256), 0, st, dndp, nc, sumVV, rank0, exact_flag, partials, nparts, l1W, nW, lamW, l1H, K, lamH, out, out2);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// nmf.m:169   W(:,k) <- W(:,k) * (1/sqrt(sum W(:,k).^2))
// cnmf.m:196-199  W(:,k,:) <- W(:,k,:) / (norm(squeeze(W(:,k,:)),'fro') / T)
__global__ __launch_bounds__(256) void w_normalize_kernel(float *W, long m, int K, int T, const double *sumsq, const uint8_t *fix,
                                                          int cnmf_rule, double *f_out, int kvalid, double *W64) {
    const int c = blockIdx.x, k = c % K, t = c / K;
    if (fix && fix[k]) return;
    if (kvalid > 0 && k >= kvalid) {         // zero padding components (nmfx_engine_desc.K_valid): 0 * (1/0) must not turn into NaN
        if (cnmf_rule == 1 && f_out && t == 0 && threadIdx.x == 0) f_out[k] = 1.0;   // (cnmf.m:165 rescales row k of H by it: the zero row stays zero)
        return;
    }
    float *w = W + m * c;
    if (W64) {   // float64 master: scaled in double, both arrays written
        double *w64 = W64 + m * c;
        double nrm;
        if (cnmf_rule == 1) {
            double s = 0.0;
            for (int tt = 0; tt < T; ++tt) s += sumsq[k + K * tt];
            nrm = sqrt(s) / (double)T;
            if (f_out && t == 0 && threadIdx.x == 0) f_out[k] = nrm;
        } else nrm = cnmf_rule == 2 ? sumsq[c] : sqrt(sumsq[c]);
        for (long i = threadIdx.x; i < m; i += 256) { const double v = w64[i] / nrm; w64[i] = v; w[i] = (float)v; }
        return;
    }
    if (cnmf_rule == 1) {
        double s = 0.0;
        for (int tt = 0; tt < T; ++tt) s += sumsq[k + K * tt];
        const double nrm = sqrt(s) / (double)T;
        const float f = (float)nrm;
        for (long i = threadIdx.x; i < m; i += 256) w[i] = w[i] / f;
        if (f_out && t == 0 && threadIdx.x == 0) f_out[k] = nrm;
    } else {
        const float f = (float)(cnmf_rule == 2 ? 1.0 / sumsq[c] : 1.0 / sqrt(sumsq[c]));   // rule 2: lnmf.m:70  W * diag(1 ./ sum(W,1))
        if ((m & 3) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0) {
            float4 *w4 = reinterpret_cast<float4 *>(w);
            for (long i = threadIdx.x; i < m / 4; i += 256) { float4 v = w4[i]; v.x *= f; v.y *= f; v.z *= f; v.w *= f; w4[i] = v; }
        } else {
            for (long i = threadIdx.x; i < m; i += 256) w[i] = w[i] * f;
        }
    }
}
nmfx_status w_normalize(hipStream_t st, float *W, long m, int K, int T, const double *sumsq, const uint8_t *fix, int cnmf_rule,
                        double *f_out, int kvalid, double *W64) {
    hipLaunchKernelGGL(w_normalize_kernel, dim3(K * T), dim3(256), 0, st, W, m, K, T, sumsq, fix, cnmf_rule, f_out, kvalid, W64);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// dst (rd x cols) <- src (rs x cols), both column-major: rows beyond rs are zero (padding K), rows beyond rd are dropped (un-padding)
__global__ void repack_rows_kernel(const float *src, int rs, float *dst, int rd, long count) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count) return;
    const int k = (int)(idx % rd);
    const long j = idx / rd;
    dst[idx] = k < rs ? src[k + (long)rs * j] : 0.0f;
}
nmfx_status repack_rows(hipStream_t st, const float *src, int rs, float *dst, int rd, long cols) {
    const long count = (long)rd * cols;
    if (count <= 0) return NMFX_OK;
    hipLaunchKernelGGL(repack_rows_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, src, rs, dst, rd, count);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// the float64 master of H: H64(k,:) *= s[k] and H = (float)H64   (cnmf.m:165)
__global__ void scale_rows64_kernel(double *H64, float *H, int K, long count, const double *s) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < count) { const double v = s[idx % K] * H64[idx]; H64[idx] = v; H[idx] = (float)v; }
}
nmfx_status scale_rows64(hipStream_t st, double *H64, float *H, int K, long n, const double *s) {
    const long count = (long)K * n;
    if (count <= 0) return NMFX_OK;
    hipLaunchKernelGGL(scale_rows64_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, H64, H, K, count, s);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}
// dst (rd x cols, float64) <- src (rs x cols, float64): rows beyond rs are zero (padding K)
__global__ void repack_rows64_kernel(const double *src, int rs, double *dst, int rd, long count) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count) return;
    const int k = (int)(idx % rd);
    const long j = idx / rd;
    dst[idx] = k < rs ? src[k + (long)rs * j] : 0.0;
}
nmfx_status repack_rows64(hipStream_t st, const double *src, int rs, double *dst, int rd, long cols) {
    const long count = (long)rd * cols;
    if (count <= 0) return NMFX_OK;
    hipLaunchKernelGGL(repack_rows64_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, src, rs, dst, rd, count);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}
// out = (float)in
__global__ void cvt_d2f_long_kernel(const double *in, float *out, long count) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < count) out[idx] = (float)in[idx];
}
nmfx_status cvt_f64_to_f32(hipStream_t st, const double *in, float *out, long count) {
    if (count <= 0) return NMFX_OK;
    hipLaunchKernelGGL(cvt_d2f_long_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, in, out, count);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// H(k,:) *= s[k]   (cnmf.m:165)
__global__ void scale_rows_kernel(float *H, int K, long count, const double *s) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < count) H[idx] = (float)(s[idx % K] * (double)H[idx]);   // (in double, as scale_rows64 forms the owner's copy of the same column)
}
nmfx_status scale_rows(hipStream_t st, float *H, int K, long n, const double *s) {
    long count = (long)K * n;
    hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, H, K, count, s);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// H <- H .* (Gn.^e ./ max(Gp.^e + lambda, eps))     nmf.m:199 / cnmf.m:231
__global__ void h_update_kernel(float *H, const float *Gn, const float *Gp, const double *Gpvec, int K, long count,
                                const float *lamH, const uint8_t *fixH, float inv_exp, int n_slabs, long slab_stride, double *H64) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count) return;
    const int k = (int)(idx % K);
    if (fixH && fixH[k]) return;
    float neg = Gn[idx];
    for (int sl = 1; sl < n_slabs; ++sl) neg += Gn[idx + sl * slab_stride];   // split partial sums, fixed order (deterministic)
    const float gsum = neg;
    if (H64) {   // float64 master: nmf.m:199 in double, both arrays written
        double dneg = (double)neg, dpos = Gp ? (double)Gp[idx] : Gpvec[k];
        const double h = H64[idx];
        double hn;
        if (inv_exp == -2.0f) hn = sqrt(h * dneg);   // lnmf.m:76
        else {
            if (inv_exp != 1.0f) { dneg = pow(dneg, (double)inv_exp); dpos = pow(dpos, (double)inv_exp); }
            hn = h * (dneg / fmax(dpos + (lamH ? (double)lamH[k] : 0.0), 2.220446049250313e-16));
        }
        H64[idx] = hn;
        H[idx] = (float)hn;
        return;
    }
    float pos = Gp ? Gp[idx] : (float)Gpvec[k];
    if (inv_exp != 1.0f) { neg = powf(neg, inv_exp); pos = powf(pos, inv_exp); }
    const float lam = lamH ? lamH[k] : 0.0f;
    if (inv_exp == -2.0f) { H[idx] = sqrtf(H[idx] * gsum); return; }   // lnmf.m:76  H = sqrt(H .* (W'*(V./V_hat)))
    H[idx] = H[idx] * (neg / fmaxf(pos + lam, NMFX_EPS_F));
}
__global__ void h_update_shift_kernel(float *H, const float *Q, const float *Gp, int K, int T, long n, long nvalid, const float *lamH, const uint8_t *fixH,
                                      float *Hpad, long padcount, long rightcount, double *H64) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x, count = (long)K * n;
    if (Hpad) {
        if (idx < padcount) Hpad[idx] = 0.0f;
        if (idx < rightcount) Hpad[padcount + count + idx] = 0.0f;
    }
    if (idx >= count) return;
    const long j = idx / K;
    const int k = (int)(idx - j * K);
    float h = H[idx];
    if (!(fixH && fixH[k])) {
        const long KT = (long)K * T;
        float neg = 0.0f;
        for (int t = 0; t < T; ++t)
            if (j + t < nvalid) neg += Q[(long)t * K + k + KT * (j + t)];   // (shift_sum_kernel's order)
        const float lam = lamH ? lamH[k] : 0.0f;
        if (H64) {   // float64 master: cnmf.m:231 in double
            const double hn = H64[idx] * ((double)neg / fmax((double)Gp[idx] + (double)lam, 2.220446049250313e-16));
            H64[idx] = hn;
            h = (float)hn;
        } else
        h = h * (neg / fmaxf(Gp[idx] + lam, NMFX_EPS_F));                   // cnmf.m:231 (h_update_kernel with inv_exp == 1)
        H[idx] = h;
    }
    if (Hpad) Hpad[padcount + idx] = h;
}
nmfx_status h_update_shift(hipStream_t st, float *H, const float *Q, const float *Gp, int K, int T, long n, long nvalid, const float *lamH, const uint8_t *fixH,
                           float *Hpad, int padL, int padR, double *H64) {
    const long count = (long)K * n, padcount = (long)K * padL, rightcount = (long)K * padR;
    const long tot = std::max(count, std::max(padcount, rightcount));
    if (tot <= 0) return NMFX_OK;
    hipLaunchKernelGGL(h_update_shift_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, H, Q, Gp, K, T, n, nvalid, lamH, fixH, Hpad, padcount, rightcount, H64);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}
nmfx_status h_update(hipStream_t st, float *H, const float *Gn, const float *Gp, const double *Gpvec, int K, long n,
                     const float *lamH, const uint8_t *fixH, float inv_exp, int n_slabs, long slab_stride, double *H64) {
    long count = (long)K * n;
    if (count <= 0) return NMFX_OK;
    hipLaunchKernelGGL(h_update_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, H, Gn, Gp, Gpvec, K, count, lamH, fixH, inv_exp,
                       n_slabs, slab_stride, H64);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// constrainednmf.m:213-237: Z <- Z .* ((Gn*A').^e ./ max((Gp*A').^e + lambda, eps)), then H = Z*A.
// A is the 0/1 label matrix of samples sorted by label (constrainednmf.m:163-170): right-multiplying by A' is a segmented
// column sum (segment c = the columns that share Z column c), and Z*A copies Z column c to every column of its segment.
// One block per segment; threadIdx.x walks K (coalesced), threadIdx.y strides the segment's columns.
__global__ __launch_bounds__(256) void z_update_kernel(float *Z, float *H, const float *Gn, const float *Gp, const double *Gpvec, int K,
                                                       const long *seg, const float *lamZ, const uint8_t *fixZ, float inv_exp, int gather_only) {
    __shared__ float red[2][4][64];
    const long c = blockIdx.x, j0 = seg[c], j1 = seg[c + 1];
    const int tx = threadIdx.x, ty = threadIdx.y;
    for (int k0 = 0; k0 < K; k0 += 64) {
        const int k = k0 + tx;
        float z = 0.f;
        if (gather_only) {
            if (k < K) z = Z[k + (long)K * c];
        } else {
            float sn = 0.f, sp = 0.f;
            const float zold = k < K ? Z[k + (long)K * c] : 0.f;                  // read by every wave BEFORE the barrier, written after it
            if (k < K)
                for (long j = j0 + ty; j < j1; j += 4) {
                    sn += Gn[k + (long)K * j];
                    if (Gp) sp += Gp[k + (long)K * j];
                }
            __syncthreads();
            red[0][ty][tx] = sn; red[1][ty][tx] = sp;
            __syncthreads();
            if (k < K) {
                float neg = red[0][0][tx] + red[0][1][tx] + red[0][2][tx] + red[0][3][tx];
                float pos = Gp ? red[1][0][tx] + red[1][1][tx] + red[1][2][tx] + red[1][3][tx]
                               : (float)(Gpvec[k] * (double)(j1 - j0));          // W'*ones(m,n)*A' (constrainednmf.m:220)
                z = zold;
                if (!(fixZ && fixZ[k])) {
                    if (inv_exp != 1.0f) { neg = powf(neg, inv_exp); pos = powf(pos, inv_exp); }
                    z = z * (neg / fmaxf(pos + (lamZ ? lamZ[k] : 0.f), NMFX_EPS_F));
                    if (ty == 0) Z[k + (long)K * c] = z;
                }
            }
        }
        if (k < K)
            for (long j = j0 + ty; j < j1; j += 4) H[k + (long)K * j] = z;        // H = Z*A (constrainednmf.m:237)
    }
}
nmfx_status z_update(hipStream_t st, float *Z, float *H, const float *Gn, const float *Gp, const double *Gpvec, int K, long nz, const long *seg,
                     const float *lamZ, const uint8_t *fixZ, float inv_exp, int gather_only) {
    if (nz <= 0) return NMFX_OK;
    hipLaunchKernelGGL(z_update_kernel, dim3((unsigned)nz), dim3(64, 4), 0, st, Z, H, Gn, Gp, Gpvec, K, seg, lamZ, fixZ, inv_exp, gather_only);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// SortDictionary.m:33-42: per basis column, the last row index (1-based) whose cumulative sum is <= half the column total
// (1 when there is none).  One 256-thread block per column; every thread owns a contiguous chunk of rows, the chunk sums are
// scanned through LDS and each thread then walks its chunk, all in fp64 and in the input's own dtype.
template <class T>
__global__ __launch_bounds__(256) void center_of_gravity_kernel(const T *W, long m, long ld, int *cog) {
    __shared__ double part[256];
    __shared__ long best[256];
    __shared__ double total_s;
    const T *w = W + ld * blockIdx.x;
    const int tid = threadIdx.x;
    const long chunk = (m + 255) / 256, i0 = tid * chunk, i1 = i0 + chunk < m ? i0 + chunk : m;
    double loc = 0.0;
    for (long i = i0; i < i1; ++i) loc += (double)w[i];
    part[tid] = loc;
    __syncthreads();
    if (tid == 0) {   // 256-entry sequential exclusive scan: keeps the left-to-right order of cumsum (SortDictionary.m:35)
        double run = 0.0;
        for (int t = 0; t < 256; ++t) { const double v = part[t]; part[t] = run; run += v; }
        total_s = run;
    }
    __syncthreads();
    const double half = total_s / 2;
    double cs = part[tid];
    long last = 0;
    for (long i = i0; i < i1; ++i) {
        cs += (double)w[i];
        if (cs <= half) last = i + 1;
    }
    __syncthreads();
    best[tid] = last;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o && best[tid + o] > best[tid]) best[tid] = best[tid + o];
        __syncthreads();
    }
    if (tid == 0) cog[blockIdx.x] = best[0] > 0 ? (int)best[0] : 1;     // SortDictionary.m:38-42
}
nmfx_status center_of_gravity(hipStream_t st, const void *W, int is_f64, long m, int K, int *cog) {
    if (K <= 0) return NMFX_OK;
    if (is_f64) hipLaunchKernelGGL(center_of_gravity_kernel<double>, dim3(K), dim3(256), 0, st, (const double *)W, m, m, cog);
    else hipLaunchKernelGGL(center_of_gravity_kernel<float>, dim3(K), dim3(256), 0, st, (const float *)W, m, m, cog);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// out(:, j) = in(:, order[j])  (by_rows == 0, SortDictionary.m:44)   |   out(k, :) = in(order[k], :)  (by_rows == 1, SortDictionary.m:46)
template <class T>
__global__ void permute_kernel(const T *in, T *out, long rows, long cols, const int *order, int by_rows) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * cols) return;
    const long r = idx % rows, c = idx / rows;
    out[idx] = by_rows ? in[order[r] + rows * c] : in[r + rows * order[c]];
}
nmfx_status permute(hipStream_t st, const void *in, void *out, int is_f64, long rows, long cols, const int *order, int by_rows) {
    const long count = rows * cols;
    if (count <= 0) return NMFX_OK;
    const dim3 g((unsigned)((count + 255) / 256)), b(256);
    if (is_f64) hipLaunchKernelGGL(permute_kernel<double>, g, b, 0, st, (const double *)in, (double *)out, rows, cols, order, by_rows);
    else hipLaunchKernelGGL(permute_kernel<float>, g, b, 0, st, (const float *)in, (float *)out, rows, cols, order, by_rows);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// cost = scale * sum(partials) + sum_c lamW[c%K]*l1W[c] + sum_k lamH[k]*l1H[k]      nmf.m:206-218
__global__ __launch_bounds__(256) void finish_cost_kernel(const double *partials, int count, double scale, const double *l1W, int nW,
                                                          const float *lamW, const double *l1H, int K, const float *lamH, double *out,
                                                          const double *dotA, const double *dotB, int ndot, const double *minus, const double *pre_c,
                                                          double pre_a, double pre_b, double *out2, const double *cvt_src, float *cvt_dst, int ncvt) {
    __shared__ double red[4];
    for (int i = threadIdx.x; i < ncvt; i += 256) cvt_dst[i] = (float)cvt_src[i];
    double s = 0.0, t = 0.0;
    for (int i = threadIdx.x; i < count; i += 256) s += partials[i];
    s = block_sum<4>(s, red);
    if (pre_c || pre_b != 0.0) s += pre_a * (pre_c ? *pre_c : 0.0) + pre_b;   // constants of the divergence inside the scaled sum (fused IS / alpha-beta)
    s *= scale;   // scale may be -Inf (alpha-beta divergence with alpha*beta == 0, nmf.m:214): apply it to the SUM
    if (l1W) for (int c = threadIdx.x; c < nW; c += 256) t += (double)lamW[c % K] * l1W[c];
    if (l1H) for (int k = threadIdx.x; k < K; k += 256) t += (double)lamH[k] * l1H[k];
    if (dotA) for (int k = threadIdx.x; k < ndot; k += 256) t += dotA[k] * dotB[k];   // closed-form sum(V_hat) of the KL cost
    t = block_sum<4>(t, red);
    if (threadIdx.x == 0) {
        const double cst = (count > 0 ? s : 0.0) + t - ((dotA && minus) ? *minus : 0.0);
        *out = cst;
        if (out2) *out2 = cst;
    }
}
nmfx_status finish_cost(hipStream_t st, const double *partials, int count, double scale, const double *l1W, int nW, const float *lamW,
                        const double *l1H, int K, const float *lamH, double *out, const double *dotA, const double *dotB, int ndot,
                        const double *minus, const double *pre_c, double pre_a, double pre_b, double *out2, const double *cvt_src, float *cvt_dst, int ncvt) {
    hipLaunchKernelGGL(finish_cost_kernel, dim3(1), dim3(256), 0, st, partials, count, scale, l1W, nW, lamW, l1H, K, lamH, out, dotA, dotB,
                       ndot, minus, pre_c, pre_a, pre_b, out2, cvt_src, cvt_dst, cvt_src && cvt_dst ? ncvt : 0);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// out[0] = scale * sum(partials) (or *src when partials == nullptr), also published to a host-mapped slot: slot[0] = the value, then -- after a
// system-scope fence -- slot[1] = seq (bit pattern), which the host thread polls instead of synchronising the stream
__global__ __launch_bounds__(256) void publish_obj_kernel(const double *partials, int count, double scale, const double *src, double *out, double *slot, unsigned long long seq) {
    __shared__ double red[4];
    double s = 0.0;
    if (partials) {
        for (int i = threadIdx.x; i < count; i += 256) s += partials[i];
        s = block_sum<4>(s, red) * scale;
    } else s = *src;
    if (threadIdx.x == 0) {
        if (out) *out = s;
        if (slot) {
            __hip_atomic_store(slot, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(slot) + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
nmfx_status publish_obj(hipStream_t st, const double *partials, int count, double scale, const double *src, double *out, double *slot, unsigned long long seq) {
    hipLaunchKernelGGL(publish_obj_kernel, dim3(1), dim3(256), 0, st, partials, count, scale, src, out, slot, seq);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// ---- line-search objectives of nmfsc without a pass over V ---------------------------------------------------------------------
// 0.5*||V - W*H||^2 is QUADRATIC in either factor, so along a line search that moves one of them (nmfsc.m:152-175: H -> Hnew, W fixed; :203-226 the
// other way round) the objective of a candidate follows from the gradient the search already has:
//     obj(X + D) - obj(X) = <grad, D> + 0.5 * sum_rows D(i,:) * G * D(i,:)'        G = W'*W (rows of H' move)  or  H*H' (rows of W move)
// with grad = dH' = ((W*H - V)'*W) or dW = (W*H - V)*H' at the point the search starts from.  Every term is of the size of the difference itself -- no
// ||V||^2-sized numbers to cancel -- so the accept test of nmfsc.m:164 / :215 is decided at least as sharply as by two fp32 evaluations of the
// objective, for K*K*R multiply-adds on the VALU instead of a 2*m*n*K pass.  X, Xc, grad: R x K column-major (rows of W, or of the transposed copy
// of H, are the K-vectors); one thread per row, D(i,:) in registers, G through LDS in row chunks.  partials[block] = 2*<grad, D> + sum D*G*D'
// (so that the 0.5 of the objective readers gives the difference).
template <int K>
__global__ __launch_bounds__(256) void quad_rows_kernel(const float *__restrict__ X, const float *__restrict__ Xc, const float *__restrict__ grad, const float *__restrict__ G,
                                                        long R, double *partials) {
    constexpr int KC = 32;                                               // rows of G per workgroup (blockIdx.y): K/32 workgroups share a block of 256 rows
    __shared__ double red[4];
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const bool ok = i < R;
    const int k0 = blockIdx.y * KC;
    float d[K];
    double t1 = 0.0;
#pragma unroll
    for (int l = 0; l < K; ++l) {
        d[l] = ok ? Xc[i + R * l] - X[i + R * l] : 0.0f;
        if (blockIdx.y == 0) t1 += ok ? (double)grad[i + R * l] * (double)d[l] : 0.0;   // (uniform)
    }
    float dk[KC];                                                        // D(i, k0 + kk): loaded again (k0 is a run-time value: d[k0 + kk] would send d[] to scratch),
#pragma unroll                                                          // all of them up front -- inside the loop each pair of loads was an exposed round trip
    for (int kk = 0; kk < KC; ++kk) dk[kk] = ok ? Xc[i + R * (k0 + kk)] - X[i + R * (k0 + kk)] : 0.0f;
    double t2 = 0.0;
    const float *__restrict__ Gk = G + (long)k0 * K;                     // wave-uniform addresses: the rows of G arrive through the scalar cache (s_load), not LDS
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) {
        const float4 *g4 = reinterpret_cast<const float4 *>(Gk + kk * K);   // G is symmetric: row k as stored
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
#pragma unroll
        for (int l = 0; l < K / 4; ++l) {
            const float4 g = g4[l];
            s0 = fmaf(g.x, d[4 * l], s0); s1 = fmaf(g.y, d[4 * l + 1], s1); s2 = fmaf(g.z, d[4 * l + 2], s2); s3 = fmaf(g.w, d[4 * l + 3], s3);
        }
        t2 += (double)dk[kk] * ((double)(s0 + s1) + (double)(s2 + s3));
    }
    const double tot = block_sum<4>(2.0 * t1 + t2, red);
    if (threadIdx.x == 0) partials[(long)blockIdx.y * gridDim.x + blockIdx.x] = tot;
}
int quad_rows_blocks(long R, int K) { return (int)((R + 255) / 256) * (K / 32); }
bool quad_rows_supported(int K) { return K >= 32 && K <= 256 && K % 32 == 0; }
nmfx_status quad_rows(hipStream_t st, const float *X, const float *Xc, const float *grad, const float *G, long R, int K, double *partials) {
    const dim3 grid((unsigned)((R + 255) / 256), (unsigned)(K / 32)), block(256);
#define NMFX_QR(KK) case KK: hipLaunchKernelGGL(quad_rows_kernel<KK>, grid, block, 0, st, X, Xc, grad, G, R, partials); break;
    switch (K) {
        NMFX_QR(32) NMFX_QR(64) NMFX_QR(96) NMFX_QR(128) NMFX_QR(160) NMFX_QR(192) NMFX_QR(224) NMFX_QR(256)
        default: set_error("quad_rows: K = %d not supported", K); return NMFX_ERR_UNSUPPORTED;
    }
#undef NMFX_QR
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// partials[block] = sum d .* (2*a + b) over the block's elements (fp64): the two inner products of the quadratic expansion when G*D came from a GEMM (cnmfsc)
__global__ __launch_bounds__(256) void dot_2a_b_kernel(const float *__restrict__ d, const float *__restrict__ a, const float *__restrict__ b, long count, double *partials) {
    __shared__ double red[4];
    double s = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long)gridDim.x * 256) s += (double)d[i] * (2.0 * (double)a[i] + (double)b[i]);
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}
int dot_2a_b_blocks(long count) { const long b = (count + 2047) / 2048; return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b)); }
nmfx_status dot_2a_b(hipStream_t st, const float *d, const float *a, const float *b, long count, double *partials) {
    hipLaunchKernelGGL(dot_2a_b_kernel, dim3((unsigned)dot_2a_b_blocks(count)), dim3(256), 0, st, d, a, b, count, partials);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

__global__ void fill_kernel(float *p, long count, float v) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < count) p[idx] = v;
}
nmfx_status fill_f32(hipStream_t st, float *p, long count, float v) {
    if (count <= 0) return NMFX_OK;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, p, count, v);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// out = y + a*x   (nmfsc.m:154 Hnew = H - stepsize*dH)
__global__ void axpy_kernel(long count, float a, const float *x, const float *y, float *out) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < count) out[idx] = y[idx] + a * x[idx];
}
nmfx_status axpy_f32(hipStream_t st, long count, float a, const float *x, const float *y, float *out) {
    if (count <= 0) return NMFX_OK;
    hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, count, a, x, y, out);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// X <- X .* (neg ./ max(pos, eps))    nmfsc.m:182,232
__global__ void mu_plain_kernel(float *X, const float *neg, const float *pos, long count) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < count) X[idx] = X[idx] * (neg[idx] / fmaxf(pos[idx], NMFX_EPS_F));
}
nmfx_status mu_plain(hipStream_t st, float *X, const float *neg, const float *pos, long count) {
    if (count <= 0) return NMFX_OK;
    hipLaunchKernelGGL(mu_plain_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, X, neg, pos, count);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// cnmfsc.m:261 on one time slice, in one launch: Xnew = X0 .* (neg ./ max(pos, eps)) and dX = Xnew - X0 (the V_hat correction of cnmfsc.m:262 wants the difference);
// the same arithmetic, in the same order, as copy + mu_plain + axpy_f32
__global__ void mu_plain_diff_kernel(const float *X0, const float *neg, const float *pos, long count, float *Xnew, float *dX) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count) return;
    const float x0 = X0[idx], xn = x0 * (neg[idx] / fmaxf(pos[idx], NMFX_EPS_F));
    Xnew[idx] = xn;
    dX[idx] = xn + (-1.0f) * x0;
}
nmfx_status mu_plain_diff(hipStream_t st, const float *X0, const float *neg, const float *pos, long count, float *Xnew, float *dX) {
    if (count <= 0) return NMFX_OK;
    hipLaunchKernelGGL(mu_plain_diff_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, X0, neg, pos, count, Xnew, dX);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// cnmfsc.m:257-263, the whole slice loop of the multiplicative W branch in ONE launch and without V_hat.  The reference keeps V_hat = sum_s W_s*rshift_s(H) up to date
// slice by slice (cnmfsc.m:262) only to contract it with rshift_t(H)' in the next slice (cnmfsc.m:259); with G = Hs*Hs' (KT x KT, the Gram of the stacked shifts)
//   pos_t = V_hat*rshift_t(H)' = sum_s Wcur_s * G[(s,.),(t,.)],     Wcur_s = the UPDATED slice for s < t, W0_s for s >= t
// row by row of W: no pass over an m x n array at all (per slice the reference's form costs one read and one write of V_hat).  The max(., 0) of cnmfsc.m:262 never
// binds on the data cnmfsc accepts: V_hat + dW*Hs = sum_s Wcur_s*Hs_s is a sum of products of non-negative factors (V >= 0 is checked at cnmfsc.m:67, H leaves the
// projection / the multiplicative update non-negative, W_t = W0_t .* neg ./ max(pos, eps) >= 0); it only removes negative ROUNDING residue, at the 1e-16 level in the
// reference's float64.  One workgroup per 16 rows of W: the rows' KT entries live in LDS and are overwritten slice by slice, G streams through LDS in chunks of 32 rows
// (prefetched into registers while the previous chunk is contracted), 32-term fp32 partial sums accumulated in double.  N = V*H_stack' (m x KT) from the one fused
// pass over V.  The last slice's correction is dead in the reference too (cnmfsc.m:269 re-forms V_hat).
template <int K, int ROWS, int CH>
__global__ __launch_bounds__(256) void cnmfsc_w_slices_kernel(const float *W0, const float *Nn, const float *G, long m, int T, float *W) {
    constexpr int RG = 256 / K, RPT = ROWS / RG, GPT = CH * K / 256;
    extern __shared__ float lds_ws[];
    const int KT = K * T, ldw = KT + 4;
    float *w = lds_ws;               // [ROWS][ldw]: the current W of these rows
    float *g = lds_ws + ROWS * ldw;  // [CH][K]: a chunk of G(:, (t,.)); afterwards pos_t of these rows, [ROWS][K]
    const int tid = threadIdx.x, k = tid % K, rg = tid / K;
    const long row0 = (long)blockIdx.x * ROWS;
    for (int i = tid; i < ROWS * KT; i += 256) {
        const int r = i % ROWS, c = i / ROWS;
        w[r * ldw + c] = row0 + r < m ? W0[row0 + r + m * (long)c] : 0.0f;
    }
    for (int t = 0; t < T; ++t) {
        const float *Gt = G + (long)t * K;   // G is symmetric: column (t,k) read as row (t,k), contiguous along k
        double acc[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) acc[r] = 0.0;
        float gn[GPT];
#pragma unroll
        for (int q = 0; q < GPT; ++q) { const int i = tid + 256 * q; gn[q] = Gt[i % K + (long)KT * (i / K)]; }
        for (int c0 = 0; c0 < KT; c0 += CH) {
            __syncthreads();   // the previous chunk is consumed (first chunk: the rows' slice t-1 / the initial load is visible, pos_(t-1) is consumed)
#pragma unroll
            for (int q = 0; q < GPT; ++q) g[tid + 256 * q] = gn[q];
            __syncthreads();
            if (c0 + CH < KT) {
#pragma unroll
                for (int q = 0; q < GPT; ++q) { const int i = tid + 256 * q; gn[q] = Gt[i % K + (long)KT * (c0 + CH + i / K)]; }
            }
            float a32[RPT];
#pragma unroll
            for (int r = 0; r < RPT; ++r) a32[r] = 0.0f;
#pragma unroll
            for (int cc = 0; cc < CH; cc += 4) {
                const float g0 = g[cc * K + k], g1 = g[(cc + 1) * K + k], g2 = g[(cc + 2) * K + k], g3 = g[(cc + 3) * K + k];
#pragma unroll
                for (int r = 0; r < RPT; ++r) {
                    const float4 wv = *reinterpret_cast<const float4 *>(&w[(rg * RPT + r) * ldw + c0 + cc]);
                    a32[r] += wv.x * g0 + wv.y * g1 + wv.z * g2 + wv.w * g3;
                }
            }
#pragma unroll
            for (int r = 0; r < RPT; ++r) acc[r] += (double)a32[r];
        }
        __syncthreads();       // every read of the last chunk and of the rows' W is done
#pragma unroll
        for (int r = 0; r < RPT; ++r) g[(rg * RPT + r) * K + k] = (float)acc[r];
        __syncthreads();
        for (int i = tid; i < ROWS * K; i += 256) {   // W_t = W0_t .* (neg ./ max(pos, eps))   cnmfsc.m:261, 16 consecutive rows per column: 64-byte segments
            const int r = i % ROWS, kk = i / ROWS;
            if (row0 + r >= m) continue;
            const long idx = row0 + r + m * ((long)t * K + kk);
            const float xn = w[r * ldw + t * K + kk] * (Nn[idx] / fmaxf(g[r * K + kk], NMFX_EPS_F));
            W[idx] = xn;
            w[r * ldw + t * K + kk] = xn;
        }
    }
}
nmfx_status cnmfsc_w_slices(hipStream_t st, const float *W0, const float *Nn, const float *G, long m, int K, int T, float *W) {
    if (m <= 0) return NMFX_OK;
    if (K != 32 && K != 64 && K != 128) { set_error("cnmfsc_w_slices: K = %d, T = %d is not served", K, T); return NMFX_ERR_UNSUPPORTED; }
    // rows per workgroup: 16, or 8 while that still leaves the CUs with fewer than two workgroups each (the kernel is a chain of LDS round trips: the CUs want
    // several workgroups to switch between); G chunks of 64 rows at K = 64 (16 KB; K*T is then a multiple of 64): half the barriers
    const bool small = (m + 15) / 16 < 512 && K <= 64;
    const int rows = small ? 8 : 16, ch = K == 64 ? 64 : 32;
    const size_t lds = sizeof(float) * ((size_t)rows * ((size_t)K * T + 4) + (size_t)ch * K);
    if (lds > 64 * 1024) { set_error("cnmfsc_w_slices: K = %d, T = %d is not served", K, T); return NMFX_ERR_UNSUPPORTED; }
    const dim3 grid((unsigned)((m + rows - 1) / rows)), block(256);
    if (K == 32 && small) hipLaunchKernelGGL((cnmfsc_w_slices_kernel<32, 8, 32>), grid, block, lds, st, W0, Nn, G, m, T, W);
    else if (K == 32) hipLaunchKernelGGL((cnmfsc_w_slices_kernel<32, 16, 32>), grid, block, lds, st, W0, Nn, G, m, T, W);
    else if (K == 64 && small) hipLaunchKernelGGL((cnmfsc_w_slices_kernel<64, 8, 64>), grid, block, lds, st, W0, Nn, G, m, T, W);
    else if (K == 64) hipLaunchKernelGGL((cnmfsc_w_slices_kernel<64, 16, 64>), grid, block, lds, st, W0, Nn, G, m, T, W);
    else hipLaunchKernelGGL((cnmfsc_w_slices_kernel<128, 16, 32>), grid, block, lds, st, W0, Nn, G, m, T, W);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// X <- X .* (neg ./ (pos + eps))     cnmfsc.m:202 (plus, not max)
__global__ void mu_plus_eps_kernel(float *X, const float *neg, const float *pos, long count) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < count) X[idx] = X[idx] * (neg[idx] / (pos[idx] + NMFX_EPS_F));
}
nmfx_status mu_plus_eps(hipStream_t st, float *X, const float *neg, const float *pos, long count) {
    if (count <= 0) return NMFX_OK;
    hipLaunchKernelGGL(mu_plus_eps_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, X, neg, pos, count);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// column scaling by s[c] or 1/s[c] (nmfsc.m:185-187)
__global__ void scale_cols_kernel(float *X, long rows, long count, const double *s, int use_sqrt, int divide) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count) return;
    double f = s[idx / rows];
    if (use_sqrt) f = sqrt(f);
    X[idx] = divide ? (float)(1.0 / f) * X[idx] : X[idx] * (float)f;
}
nmfx_status scale_cols(hipStream_t st, float *X, long rows, int ncols, const double *s, int use_sqrt, int divide) {
    long count = rows * ncols;
    if (count <= 0) return NMFX_OK;
    hipLaunchKernelGGL(scale_cols_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, X, rows, count, s, use_sqrt, divide);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// out (cols x rows, column-major) = in' ; 32x32 LDS tiles
__global__ __launch_bounds__(256) void transpose_kernel(const float *in, long rows, long cols, float *out) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long rtiles = (rows + 31) / 32;   // 1-D grid over the tiles: neither extent is bound by the 65535 limit of grid.y
    const long r0 = ((long)blockIdx.x % rtiles) * 32, c0 = ((long)blockIdx.x / rtiles) * 32;
    for (int y = ty; y < 32; y += 8)
        if (r0 + tx < rows && c0 + y < cols) tile[y][tx] = in[(r0 + tx) + rows * (c0 + y)];
    __syncthreads();
    for (int y = ty; y < 32; y += 8)
        if (c0 + tx < cols && r0 + y < rows) out[(c0 + tx) + cols * (r0 + y)] = tile[tx][y];
}
nmfx_status transpose_f32(hipStream_t st, const float *in, long rows, long cols, float *out) {
    if (rows <= 0 || cols <= 0) return NMFX_OK;
    const long tiles = ((rows + 31) / 32) * ((cols + 31) / 32);
    if (tiles > 0x7fffffffL) { set_error("transpose_f32: %ld x %ld is too large", rows, cols); return NMFX_ERR_INVALID; }
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)tiles), dim3(256), 0, st, in, rows, cols, out);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// broadcast vector helper for KL: Pvec[c = k + K*t] = sum_{j < n - t} H[k,j]  given full row sums and the tail columns
//   (cnmf.m:191-192 with V_pos = ones: ones(m,n) * H_shifted' = rowsum of the first n-t columns)
__global__ void kl_pvec_kernel(const double *rowsum, const float *H, int K, long n, int T, double *Pvec, int hL) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= K * T) return;
    const int k = c % K, t = c / K;
    double s = rowsum[k];
    for (int d = 0; d < t; ++d) s -= (double)H[k + K * (n - 1 - d)];
    for (int d = 1; d <= t && d <= hL; ++d) s += (double)H[(long)k - (long)K * d];   // columns -d of a shard's left halo
    Pvec[c] = s;
}
nmfx_status kl_pvec(hipStream_t st, const double *rowsum, const float *H, int K, long n, int T, double *Pvec, int halo_left) {
    hipLaunchKernelGGL(kl_pvec_kernel, dim3((K * T + 63) / 64), dim3(64), 0, st, rowsum, H, K, n, T, Pvec, halo_left);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// Gpvec[k] = sum_t colsum(W_t)[k]   (cnmf.m:220-221: V_pos = ones is NOT shifted for KL)
__global__ void sum_over_t_kernel(const double *colsum, int K, int T, double *out) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    double s = 0.0;
    for (int t = 0; t < T; ++t) s += colsum[k + K * t];
    out[k] = s;
}
nmfx_status sum_over_t(hipStream_t st, const double *colsum, int K, int T, double *out) {
    hipLaunchKernelGGL(sum_over_t_kernel, dim3((K + 63) / 64), dim3(64), 0, st, colsum, K, T, out);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// out[0] = sum of v[0..count)   (fp64)
__global__ __launch_bounds__(256) void sum_vec_kernel(const double *v, long count, double *out) {
    __shared__ double red[4];
    double s = 0.0;
    for (long i = threadIdx.x; i < count; i += 256) s += v[i];
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0) *out = s;
}
nmfx_status sum_vec(hipStream_t st, const double *v, long count, double *out) {
    hipLaunchKernelGGL(sum_vec_kernel, dim3(1), dim3(256), 0, st, v, count, out);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// ---- nmfsc with a handful of components (K <= 8): gradients and objective in fp64 on the VALU ------------------------------------
// The Hoyer projection that follows the gradient step amplifies what it is given (it removes most of a row's mass; 70x on K = 3
// problems), so an fp32-accumulated gradient -- fine at K >= 8 -- costs parity there: H off by 1.2e-5 (scripts/fuzz_campaign_sc.py).
// With K this small the contractions are bandwidth work, so they are done exactly: r = W*H - V and the sums over it in doubles,
// results left as doubles for projfunc (dir64).  V is read once; W / H come out of L2.
//   dHT (n x KV, optional)  dH' = (W'*(W*H - V))'            nmfsc.m:144-148
//   partials[gridDim.x]     sum (W*H - V).^2 per workgroup   nmfsc.m:139,161  (the caller halves the total)
template <int KV>
__global__ __launch_bounds__(256) void smallk_dh_kernel(const float *V, long m, long n, const float *W, const float *H, int ldh, double *dHT, double *partials) {
    __shared__ double red[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double cost = 0.0;
    for (long j = (long)blockIdx.x * 4 + w; j < n; j += (long)gridDim.x * 4) {   // one wave per column of V
        double h[KV], acc[KV];
#pragma unroll
        for (int k = 0; k < KV; ++k) { h[k] = (double)H[k + (long)ldh * j]; acc[k] = 0.0; }
        const float *v = V + m * j;
        for (long i = lane; i < m; i += 64) {
            double wv[KV], sv = 0.0;
#pragma unroll
            for (int k = 0; k < KV; ++k) { wv[k] = (double)W[i + m * k]; sv = fma(wv[k], h[k], sv); }
            const double r = sv - (double)v[i];
            cost = fma(r, r, cost);
#pragma unroll
            for (int k = 0; k < KV; ++k) acc[k] = fma(wv[k], r, acc[k]);
        }
        if (dHT) {
#pragma unroll
            for (int k = 0; k < KV; ++k) { const double t = wave_sum(acc[k]); if (lane == 0) dHT[j + n * k] = t; }
        }
    }
    cost = block_sum<4>(cost, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = cost;
}
//   slabs[blockIdx.y][m x KV]   (W*H - V) * H' over the column chunk [y*cpc, (y+1)*cpc)   nmfsc.m:194-200 ; partials as above
template <int KV>
__global__ __launch_bounds__(256) void smallk_dw_kernel(const float *V, long m, long n, const float *W, const float *H, int ldh, long cpc, double *slabs, double *partials) {
    __shared__ double red[4];
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const bool ok = i < m;
    const long c0 = (long)blockIdx.y * cpc, c1 = c0 + cpc < n ? c0 + cpc : n;
    double wv[KV], acc[KV], cost = 0.0;
#pragma unroll
    for (int k = 0; k < KV; ++k) { wv[k] = ok ? (double)W[i + m * k] : 0.0; acc[k] = 0.0; }
    for (long j = c0; j < c1; ++j) {
        double h[KV], sv = 0.0;
#pragma unroll
        for (int k = 0; k < KV; ++k) { h[k] = (double)H[k + (long)ldh * j]; sv = fma(wv[k], h[k], sv); }   // wave-uniform loads
        const double r = sv - (ok ? (double)V[i + m * j] : 0.0);
        cost = fma(r, r, cost);
#pragma unroll
        for (int k = 0; k < KV; ++k) acc[k] = fma(r, h[k], acc[k]);
    }
    if (ok) {
#pragma unroll
        for (int k = 0; k < KV; ++k) slabs[(long)blockIdx.y * m * KV + i + m * k] = acc[k];
    }
    cost = block_sum<4>(cost, red);
    if (threadIdx.x == 0) partials[(long)blockIdx.y * gridDim.x + blockIdx.x] = cost;
}
__global__ void sum_slabs_f64_kernel(const double *slabs, int nslab, long count, double *out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count) return;
    double s = slabs[idx];
    for (int z = 1; z < nslab; ++z) s += slabs[(long)z * count + idx];
    out[idx] = s;
}
int smallk_max() { return 8; }
int smallk_dw_chunks(long m, long n) {
    const long gx = (m + 255) / 256;
    long c = (2048 + gx - 1) / gx;
    const long cmax = (n + 63) / 64;
    if (c > cmax) c = cmax;
    return (int)(c < 1 ? 1 : c);
}
int smallk_partials(long m, long n) { return (int)std::max<long>(std::min<long>((n + 3) / 4, 2048), (m + 255) / 256 * smallk_dw_chunks(m, n)); }
template <int KV>
static nmfx_status smallk_launch(hipStream_t st, int Kv, const float *V, long m, long n, const float *W, const float *H, int ldh, double *dHT, double *dW, double *slabs,
                                 double *partials, int *nparts) {
    if constexpr (KV > 1) { if (Kv < KV) return smallk_launch<KV - 1>(st, Kv, V, m, n, W, H, ldh, dHT, dW, slabs, partials, nparts); }
    if (dW) {
        const int nch = smallk_dw_chunks(m, n);
        const long cpc = (n + nch - 1) / nch;
        const dim3 grid((unsigned)((m + 255) / 256), (unsigned)nch);
        hipLaunchKernelGGL(smallk_dw_kernel<KV>, grid, dim3(256), 0, st, V, m, n, W, H, ldh, cpc, nch == 1 ? dW : slabs, partials);
        NMFX_HIP(hipGetLastError());
        if (nch > 1) {
            hipLaunchKernelGGL(sum_slabs_f64_kernel, dim3((unsigned)((m * KV + 255) / 256)), dim3(256), 0, st, slabs, nch, m * KV, dW);
            NMFX_HIP(hipGetLastError());
        }
        *nparts = (int)(grid.x * grid.y);
        return NMFX_OK;
    }
    const unsigned blocks = (unsigned)std::min<long>((n + 3) / 4, 2048);
    hipLaunchKernelGGL(smallk_dh_kernel<KV>, dim3(blocks), dim3(256), 0, st, V, m, n, W, H, ldh, dHT, partials);
    NMFX_HIP(hipGetLastError());
    *nparts = (int)blocks;
    return NMFX_OK;
}
// dW != nullptr: dW (m x Kv doubles) = (W*H - V)*H' (slabs: smallk_dw_chunks(m, n) * m * Kv doubles); else dHT (n x Kv doubles, may be nullptr:
// objective only) = (W'*(W*H - V))'.  partials[*nparts] always receive sum (W*H - V).^2.  W: m x Kv (column stride m), H: column stride ldh.
nmfx_status smallk_grad(hipStream_t st, int Kv, const float *V, long m, long n, const float *W, const float *H, int ldh, double *dHT, double *dW, double *slabs,
                        double *partials, int *nparts) {
    if (Kv < 1 || Kv > 8) { set_error("smallk_grad: K = %d", Kv); return NMFX_ERR_INVALID; }
    return smallk_launch<8>(st, Kv, V, m, n, W, H, ldh, dHT, dW, slabs, partials, nparts);
}

// cnmfsc, sparse-W branch on small problems: dW_t = (V_hat - V) * rshift_t(H)' (cnmfsc.m:221-224) accumulated in fp64, one thread per
// output element and column chunk.  The columns of W are short there and the Hoyer projection amplifies the fp32 accumulation noise of
// an MFMA contraction over n (W off by 1.3e-5 on 71 x 218 and 388 x 156 problems in scripts/fuzz_campaign_sc.py); the work is m*n*K
// fp64 FMAs, so this is for small problems only (the caller decides).
// ... and, since round 6, the residual those contractions run on: R64 = sum_{t < Tn} W_t * rshift_t(H) - V in float64 (one thread per element, K*Tn fp64 FMAs), instead of
// the difference of the fp32 V_hat and V.  On ill-conditioned sparse-W problems the search amplifies what its gradient carries by 200x and more PER ITERATION
// (scripts/cnmfsc_sparse_w_outlier.py: 222 x 100, K = 32, T = 2 moves 5.6e-6 in three iterations under a 3e-8 rounding of its inputs in the float64 algorithm
// itself); the rounding of V_hat to fp32 was seven times that perturbation, and W ended at 4.4e-5 (round 5's one campaign problem outside the contract).
__global__ __launch_bounds__(256) void recon_resid64_kernel(const float *V, const float *W, long m, long n, int K, int Tn, const float *H, double *R64) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (i >= m) return;
    double acc = 0.0;
    for (int t = 0; t < Tn && t <= j; ++t) {
        const float *w = W + i + m * (long)K * t, *h = H + (long)K * (j - t);
        for (int k = 0; k < K; ++k) acc = fma((double)w[m * k], (double)h[k], acc);
    }
    R64[i + m * j] = acc - (double)V[i + m * j];
}
nmfx_status recon_resid64(hipStream_t st, const float *V, const float *W, long m, long n, int K, int Tn, const float *H, double *R64) {
    if (m <= 0 || n <= 0) return NMFX_OK;
    if (n > 65535) { set_error("recon_resid64: n = %ld (small problems only)", n); return NMFX_ERR_INVALID; }
    hipLaunchKernelGGL(recon_resid64_kernel, dim3((unsigned)((m + 255) / 256), (unsigned)n), dim3(256), 0, st, V, W, m, n, K, Tn, H, R64);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}
__global__ __launch_bounds__(256) void resid_xht64_kernel(const float *V, const float *Vh, const double *R64, long m, long n, const float *H, int K, int t, long cpc, double *slabs) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y;
    if (i >= m) return;
    long c0 = (long)blockIdx.z * cpc, c1 = c0 + cpc < n ? c0 + cpc : n;
    if (c0 < t) c0 = t;                                   // rshift_t(H)(:, j) = H(:, j - t), zero for j < t
    double acc = 0.0;
    if (R64) { for (long j = c0; j < c1; ++j) acc = fma(R64[i + m * j], (double)H[k + (long)K * (j - t)], acc); }
    else for (long j = c0; j < c1; ++j) acc = fma((double)Vh[i + m * j] - (double)V[i + m * j], (double)H[k + (long)K * (j - t)], acc);
    slabs[(long)blockIdx.z * m * K + i + m * k] = acc;
}
nmfx_status resid_xht64(hipStream_t st, const float *V, const float *Vh, const double *R64, long m, long n, const float *H, int K, int t, double *slabs, int nch, double *out) {
    const long cpc = (n + nch - 1) / nch;
    hipLaunchKernelGGL(resid_xht64_kernel, dim3((unsigned)((m + 255) / 256), (unsigned)K, (unsigned)nch), dim3(256), 0, st, V, Vh, R64, m, n, H, K, t, cpc, nch == 1 ? out : slabs);
    NMFX_HIP(hipGetLastError());
    if (nch > 1) {
        hipLaunchKernelGGL(sum_slabs_f64_kernel, dim3((unsigned)((m * K + 255) / 256)), dim3(256), 0, st, slabs, nch, m * K, out);
        NMFX_HIP(hipGetLastError());
    }
    return NMFX_OK;
}

// ... and the sparse-H branch: dH'(j, k) = sum_t sum_i W_t(i, k) * (V_hat - V)(i, j + t), j + t < n (cnmfsc.m:160-168), in fp64: one wave per
// output element, lanes along i (the columns of V_hat / V and of W_t are contiguous there); out is n x K (the layout projfunc reads)
__global__ __launch_bounds__(256) void resid_hgrad64_kernel(const float *V, const float *Vh, const double *R64, long m, long n, const float *W, int K, int T, double *outT) {
    const int lane = threadIdx.x & 63;
    const long o = (long)blockIdx.x * 4 + (threadIdx.x >> 6);   // o = j + n * k
    if (o >= n * K) return;
    const long j = o % n;
    const int k = (int)(o / n);
    double acc = 0.0;
    for (int t = 0; t < T && j + t < n; ++t) {
        const float *w = W + m * ((long)k + (long)K * t);
        if (R64) {
            const double *rr = R64 + m * (j + t);
            for (long i = lane; i < m; i += 64) acc = fma((double)w[i], rr[i], acc);
            continue;
        }
        const float *v = V + m * (j + t), *vh = Vh + m * (j + t);
        for (long i = lane; i < m; i += 64) acc = fma((double)w[i], (double)vh[i] - (double)v[i], acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) outT[o] = acc;
}
nmfx_status resid_hgrad64(hipStream_t st, const float *V, const float *Vh, const double *R64, long m, long n, const float *W, int K, int T, double *outT) {
    hipLaunchKernelGGL(resid_hgrad64_kernel, dim3((unsigned)((n * K + 3) / 4)), dim3(256), 0, st, V, Vh, R64, m, n, W, K, T, outT);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// nmfsc on small problems with K > 8 (no V_hat in HBM there): R64 = W*H - V as doubles (one thread per element, K fp64 FMAs each) with the
// objective's sum of squares per workgroup; the two contractions above then run on it (Vh = nullptr: X IS the residual)
__global__ __launch_bounds__(256) void resid64_kernel(const float *V, long m, long n, const float *W, const float *H, int K, int ldh, double *R64, double *partials) {
    __shared__ double red[4];
    const long total = m * n;
    double cost = 0.0;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long i = e % m, j = e / m;
        double sv = 0.0;
        for (int k = 0; k < K; ++k) sv = fma((double)W[i + m * k], (double)H[k + (long)ldh * j], sv);
        const double r = sv - (double)V[e];
        if (R64) R64[e] = r;
        cost = fma(r, r, cost);
    }
    cost = block_sum<4>(cost, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = cost;
}
int resid64_blocks(long m, long n) { return (int)std::min<long>((m * n + 255) / 256, 2048); }
nmfx_status resid64(hipStream_t st, const float *V, long m, long n, const float *W, const float *H, int K, int ldh, double *R64, double *partials, int *nparts) {
    const int blocks = resid64_blocks(m, n);
    hipLaunchKernelGGL(resid64_kernel, dim3((unsigned)blocks), dim3(256), 0, st, V, m, n, W, H, K, ldh, R64, partials);
    NMFX_HIP(hipGetLastError());
    *nparts = blocks;
    return NMFX_OK;
}
// dH'(j, k) = sum_i W(i, k) * R64(i, j)   (n x K doubles): one wave per output element
__global__ __launch_bounds__(256) void r64_wt_kernel(const double *R64, long m, long n, const float *W, int K, double *outT) {
    const int lane = threadIdx.x & 63;
    const long o = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= n * K) return;
    const long j = o % n;
    const float *w = W + m * (o / n);
    const double *r = R64 + m * j;
    double acc = 0.0;
    for (long i = lane; i < m; i += 64) acc = fma((double)w[i], r[i], acc);
    acc = wave_sum(acc);
    if (lane == 0) outT[o] = acc;
}
// dW(i, k) = sum_j R64(i, j) * H(k, j)   (m x K doubles): one thread per output element and column chunk
__global__ __launch_bounds__(256) void r64_ht_kernel(const double *R64, long m, long n, const float *H, int K, int ldh, long cpc, double *slabs) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y;
    if (i >= m) return;
    const long c0 = (long)blockIdx.z * cpc, c1 = c0 + cpc < n ? c0 + cpc : n;
    double acc = 0.0;
    for (long j = c0; j < c1; ++j) acc = fma(R64[i + m * j], (double)H[k + (long)ldh * j], acc);
    slabs[(long)blockIdx.z * m * K + i + m * k] = acc;
}
nmfx_status r64_wt(hipStream_t st, const double *R64, long m, long n, const float *W, int K, double *outT) {
    hipLaunchKernelGGL(r64_wt_kernel, dim3((unsigned)((n * K + 3) / 4)), dim3(256), 0, st, R64, m, n, W, K, outT);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}
nmfx_status r64_ht(hipStream_t st, const double *R64, long m, long n, const float *H, int K, int ldh, double *slabs, int nch, double *out) {
    const long cpc = (n + nch - 1) / nch;
    hipLaunchKernelGGL(r64_ht_kernel, dim3((unsigned)((m + 255) / 256), (unsigned)K, (unsigned)nch), dim3(256), 0, st, R64, m, n, H, K, ldh, cpc, nch == 1 ? out : slabs);
    NMFX_HIP(hipGetLastError());
    if (nch > 1) {
        hipLaunchKernelGGL(sum_slabs_f64_kernel, dim3((unsigned)((m * K + 255) / 256)), dim3(256), 0, st, slabs, nch, m * K, out);
        NMFX_HIP(hipGetLastError());
    }
    return NMFX_OK;
}

// packed buffer helpers for the multi-GPU exchange: doubles <-> floats
__global__ void d2f_kernel(const double *in, float *out, int count) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = (float)in[i];
}
__global__ void f2d_kernel(const float *in, double *out, int count) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = (double)in[i];
}
nmfx_status d2f(hipStream_t st, const double *in, float *out, int count) {
    if (count <= 0) return NMFX_OK;
    hipLaunchKernelGGL(d2f_kernel, dim3((count + 255) / 256), dim3(256), 0, st, in, out, count);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}
nmfx_status f2d(hipStream_t st, const float *in, double *out, int count) {
    if (count <= 0) return NMFX_OK;
    hipLaunchKernelGGL(f2d_kernel, dim3((count + 255) / 256), dim3(256), 0, st, in, out, count);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}


// host dtype <-> device fp32 conversion (device side, so uploads of float64 MATLAB arrays stay PCIe-bound)
__global__ void cvt_d2f_kernel(const double *in, float *out, long count, double inv_scale) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = (float)(inv_scale == 1.0 ? in[i] : in[i] / inv_scale);
}
__global__ void cvt_f2f_kernel(const float *in, float *out, long count, double inv_scale) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = inv_scale == 1.0 ? in[i] : (float)((double)in[i] / inv_scale);
}
__global__ void cvt_f2d_kernel(const float *in, double *out, long count) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = (double)in[i];
}
nmfx_status cvt_to_f32(hipStream_t st, const void *in, int dtype, float *out, long count, double divide_by) {
    if (count <= 0) return NMFX_OK;
    dim3 g((unsigned)((count + 255) / 256)), b(256);
    if (dtype == NMFX_F64) hipLaunchKernelGGL(cvt_d2f_kernel, g, b, 0, st, static_cast<const double *>(in), out, count, divide_by);
    else hipLaunchKernelGGL(cvt_f2f_kernel, g, b, 0, st, static_cast<const float *>(in), out, count, divide_by);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}
nmfx_status cvt_to_f64(hipStream_t st, const float *in, double *out, long count) {
    if (count <= 0) return NMFX_OK;
    hipLaunchKernelGGL(cvt_f2d_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, in, out, count);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// ---- nmfsc.m:57-62 on a device-resident shard: max / -min in two deterministic stages, and the rescale -----------------------------------
constexpr int MM_BLOCKS = 1024;
__global__ __launch_bounds__(256) void minmax_stage1_kernel(const float *X, long count, float *part) {
    __shared__ float smx[4], smn[4];
    float mx = -INFINITY, mn = INFINITY;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long)gridDim.x * 256) { const float v = X[i]; mx = fmaxf(mx, v); mn = fminf(mn, v); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, o)); mn = fminf(mn, __shfl_xor(mn, o)); }
    if ((threadIdx.x & 63) == 0) { smx[threadIdx.x >> 6] = mx; smn[threadIdx.x >> 6] = mn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
        part[2 * blockIdx.x + 1] = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
    }
}
__global__ __launch_bounds__(256) void minmax_stage2_kernel(const float *part, int nblk, double *out) {
    __shared__ float smx[4], smn[4];
    float mx = -INFINITY, mn = INFINITY;
    for (int b = threadIdx.x; b < nblk; b += 256) { mx = fmaxf(mx, part[2 * b]); mn = fminf(mn, part[2 * b + 1]); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, o)); mn = fminf(mn, __shfl_xor(mn, o)); }
    if ((threadIdx.x & 63) == 0) { smx[threadIdx.x >> 6] = mx; smn[threadIdx.x >> 6] = mn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = (double)fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
        out[1] = -(double)fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
    }
}
__global__ void scale_div_kernel(const float *X, long count, double divide_by, float *out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = (float)((double)X[i] / divide_by);   // the value the host-side ingest of the blocking call produces (host_io.hip::narrow)
}

// out_dev[0] = max(X), out_dev[1] = -min(X): both combine under a MAX all-reduce (nmfsc.m:57-62 on a column shard).  The caller has selected X_dev's device.
nmfx_status minmax_dev(hipStream_t st, const float *X_dev, long count, double *out_dev) {
    // stage-1 partials live in a per-call device buffer freed in stream order (a shard is preprocessed once per factorisation: not a hot path)
    float *part = nullptr;
    NMFX_HIP(hipMallocAsync(reinterpret_cast<void **>(&part), sizeof(float) * 2 * MM_BLOCKS, st));
    const int nblk = (int)std::min<long>(MM_BLOCKS, (count + 255) / 256);
    hipLaunchKernelGGL(minmax_stage1_kernel, dim3(nblk), dim3(256), 0, st, X_dev, count, part);
    hipLaunchKernelGGL(minmax_stage2_kernel, dim3(1), dim3(256), 0, st, part, nblk, out_dev);
    hipError_t le = hipGetLastError();
    hipError_t fe = hipFreeAsync(part, st);
    if (le != hipSuccess || fe != hipSuccess) { set_error("nmfx_minmax_dev: %s", hipGetErrorString(le != hipSuccess ? le : fe)); return NMFX_ERR_HIP; }
    return NMFX_OK;
}
// out = X / divide_by in double, rounded once (nmfsc.m:62).  divide_by == 0 divides all the same: an all-zero V becomes NaN as in the reference.
nmfx_status scale_div(hipStream_t st, const float *X_dev, long count, double divide_by, float *out_dev) {
    hipLaunchKernelGGL(scale_div_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, X_dev, count, divide_by, out_dev);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

}  // namespace nmfx

// Hoyer's L1/L2 projection (projfunc.m:13-65) for `count` vectors stored as the columns of X
// (len x count, column-major), in place.  One 1024-thread workgroup per vector; the vector lives in
// registers (+ LDS beyond 16 elements per thread) as fp64 for the whole iteration (len <= 32768), every reduction (sum, w'w, w'v, v'v,
// |Z|, all(v>=0)) is a wave-shuffle + LDS tree in fp64 so the discrete branches (v<=0 sets,
// nmfsc.m:164 objective test downstream) follow the float64 reference.  Bandwidth-class: the only
// HBM traffic is one read and one write of the vector.
#include <vector>

#include "nmfx_internal.h"

namespace nmfx {

constexpr int PF_THREADS = 1024;
constexpr int PF_WAVES = PF_THREADS / 64;
constexpr int PF_MAX_ITERS = 100000;  // safety cap: the reference loops forever on NaN input

struct Red4 { double a, b, c, d; };

__device__ __forceinline__ Red4 block_red4(Red4 v, double *red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        v.a += __shfl_xor(v.a, o);
        v.b += __shfl_xor(v.b, o);
        v.c += __shfl_xor(v.c, o);
        v.d += __shfl_xor(v.d, o);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) { red[wave * 4 + 0] = v.a; red[wave * 4 + 1] = v.b; red[wave * 4 + 2] = v.c; red[wave * 4 + 3] = v.d; }
    __syncthreads();
    Red4 r = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int w = 0; w < PF_WAVES; ++w) { r.a += red[w * 4 + 0]; r.b += red[w * 4 + 1]; r.c += red[w * 4 + 2]; r.d += red[w * 4 + 3]; }
    return r;
}

// Working vector: element e of thread tid is x[tid + e*1024].  The first ER elements per thread live in registers (fp64, 2 VGPRs
// each), the next EL in LDS (fp64, [e][tid]: conflict-free b64 accesses) -- 16 + 16 covers len <= 32768 (a row of H at BASELINE
// config 5) with 128 KiB of LDS and no scratch spills (the all-register variant needed 64 + temporaries > 128 VGPRs at 1024 threads
// and spilled 212 of them).  TIO = float (engine buffers) or double (nmfx_projfunc on float64 input: no fp32 rounding anywhere).
// dir != nullptr fuses the line-search step into the load: s = x + mu*dir (fp32, as the separate axpy kernel computed it; nmfsc.m:154).
template <int ER, int EL, typename TIO>
__global__ __launch_bounds__(PF_THREADS) void projfunc_kernel(TIO *X, long len, double k1, double k2, int nn, int *usediters, const float *dir, float mu) {
    __shared__ double red[PF_WAVES * 4];
    extern __shared__ __attribute__((aligned(16))) double vl[];   // [EL][PF_THREADS]
    constexpr int EPT = ER + EL;
    static_assert(EPT <= 64, "one mask bit per element");
    TIO *x = X + len * blockIdx.x;
    const float *dx = dir ? dir + len * blockIdx.x : nullptr;
    const int tid = threadIdx.x;
    const double N = (double)len;
    double vr[ER > 0 ? ER : 1];
    unsigned long long zmask = 0ull, negmask = 0ull;
    auto get = [&](int e) -> double { return e < ER ? vr[e < ER ? e : 0] : vl[(e - ER) * PF_THREADS + tid]; };
    auto put = [&](int e, double val) { if (e < ER) vr[e < ER ? e : 0] = val; else vl[(e - ER) * PF_THREADS + tid] = val; };

    Red4 r = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const long i = tid + (long)e * PF_THREADS;
        double s = 0.0;
        if (i < len) {
            if (sizeof(TIO) == 4 && dx) s = (double)((float)x[i] + mu * dx[i]);
            else s = (double)x[i];
            if (!nn) { if (s < 0) negmask |= 1ull << e; s = fabs(s); }   // projfunc.m:16-19
        }
        put(e, s);
        r.a += s;
    }
    r = block_red4(r, red);
    const double shift0 = (k1 - r.a) / N;                                  // projfunc.m:22
#pragma unroll
    for (int e = 0; e < EPT; ++e) put(e, get(e) + shift0);

    double nz = 0.0;
    int j = 0;
    for (;;) {
        const double mid = k1 / (N - nz);                                  // projfunc.m:31-32
        r.a = r.b = r.c = r.d = 0.0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const long i = tid + (long)e * PF_THREADS;
            if (i < len) {
                const double ve = get(e);
                const double w = ve - (((zmask >> e) & 1ull) ? 0.0 : mid); // projfunc.m:33
                r.a += w * w;                                              // projfunc.m:34
                r.b += w * ve;                                             // projfunc.m:35
                r.c += ve * ve;                                            // projfunc.m:36
            }
        }
        r = block_red4(r, red);
        const double a = r.a, b = 2.0 * r.b, c = r.c - k2;
        const double disc = b * b - 4.0 * a * c;
        const double alphap = (-b + (disc > 0.0 ? sqrt(disc) : 0.0)) / (2.0 * a);   // projfunc.m:37 real(sqrt(.))
        r.a = r.b = r.c = r.d = 0.0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const long i = tid + (long)e * PF_THREADS;
            if (i < len) {
                const double ve = get(e);
                const double w = ve - (((zmask >> e) & 1ull) ? 0.0 : mid);
                const double vn = alphap * w + ve;                         // projfunc.m:38
                put(e, vn);
                if (!(vn >= 0.0)) r.a += 1.0;                              // projfunc.m:40 all(v>=0)
            }
        }
        r = block_red4(r, red);
        if (r.a == 0.0 || j >= PF_MAX_ITERS) break;                        // projfunc.m:40-44
        ++j;                                                               // projfunc.m:46
        zmask = 0ull;
        r.a = r.b = r.c = r.d = 0.0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const long i = tid + (long)e * PF_THREADS;
            if (i < len) {
                double ve = get(e);
                if (ve <= 0.0) { zmask |= 1ull << e; ve = 0.0; put(e, 0.0); r.b += 1.0; }   // projfunc.m:49-50
                r.a += ve;                                                              // projfunc.m:51
            }
        }
        r = block_red4(r, red);
        nz = r.b;
        const double shift = (k1 - r.a) / (N - nz);                        // projfunc.m:52
#pragma unroll
        for (int e = 0; e < EPT; ++e)
            if (!((zmask >> e) & 1ull)) put(e, get(e) + shift);            // projfunc.m:52-53
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const long i = tid + (long)e * PF_THREADS;
        if (i < len) x[i] = (TIO)(((negmask >> e) & 1ull) ? -get(e) : get(e));         // projfunc.m:58-60
    }
    if (usediters && tid == 0) usediters[blockIdx.x] = j + 1;
}

template <int ER, int EL, typename TIO>
static nmfx_status launch_pf(hipStream_t st, TIO *X, long len, int count, double k1, double k2, int nn, int *usediters, const float *dir, float mu) {
    auto kern = projfunc_kernel<ER, EL, TIO>;
    const size_t ldsb = sizeof(double) * EL * PF_THREADS;
    static bool attr_done = false;
    if (ldsb > 48 * 1024 && !attr_done) {
        NMFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(count), dim3(PF_THREADS), ldsb, st, X, len, k1, k2, nn, usediters, dir, mu);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// Any length: the working vector lives in a global fp64 scratch row (L2-resident for realistic sizes) instead of registers.
template <typename TIO>
__global__ __launch_bounds__(PF_THREADS) void projfunc_long_kernel(TIO *X, long len, double k1, double k2, int nn, int *usediters, double *scratch,
                                                                  unsigned char *flags, const float *dir, float mu) {
    __shared__ double red[PF_WAVES * 4];
    TIO *x = X + len * blockIdx.x;
    const float *dx = dir ? dir + len * blockIdx.x : nullptr;
    double *v = scratch + len * blockIdx.x;
    unsigned char *fl = flags + len * blockIdx.x;   // bit0: in Z, bit1: was negative
    const int tid = threadIdx.x;
    const double N = (double)len;
    Red4 r = {0.0, 0.0, 0.0, 0.0};
    for (long i = tid; i < len; i += PF_THREADS) {
        double s = (sizeof(TIO) == 4 && dx) ? (double)((float)x[i] + mu * dx[i]) : (double)x[i];
        unsigned char f = 0;
        if (!nn) { if (s < 0) f = 2; s = fabs(s); }
        v[i] = s; fl[i] = f; r.a += s;
    }
    r = block_red4(r, red);
    const double shift0 = (k1 - r.a) / N;
    for (long i = tid; i < len; i += PF_THREADS) v[i] += shift0;
    double nz = 0.0;
    int j = 0;
    for (;;) {
        const double mid = k1 / (N - nz);
        r.a = r.b = r.c = r.d = 0.0;
        for (long i = tid; i < len; i += PF_THREADS) {
            const double vi = v[i], w = vi - ((fl[i] & 1) ? 0.0 : mid);
            r.a += w * w; r.b += w * vi; r.c += vi * vi;
        }
        r = block_red4(r, red);
        const double a = r.a, b = 2.0 * r.b, c = r.c - k2;
        const double disc = b * b - 4.0 * a * c;
        const double alphap = (-b + (disc > 0.0 ? sqrt(disc) : 0.0)) / (2.0 * a);
        r.a = r.b = r.c = r.d = 0.0;
        for (long i = tid; i < len; i += PF_THREADS) {
            const double vi = v[i], w = vi - ((fl[i] & 1) ? 0.0 : mid);
            const double vn = alphap * w + vi;
            v[i] = vn;
            if (!(vn >= 0.0)) r.a += 1.0;
        }
        r = block_red4(r, red);
        if (r.a == 0.0 || j >= PF_MAX_ITERS) break;
        ++j;
        r.a = r.b = r.c = r.d = 0.0;
        for (long i = tid; i < len; i += PF_THREADS) {
            double vi = v[i];
            unsigned char f = fl[i] & 2;
            if (vi <= 0.0) { f |= 1; vi = 0.0; v[i] = 0.0; r.b += 1.0; }
            fl[i] = f;
            r.a += vi;
        }
        r = block_red4(r, red);
        nz = r.b;
        const double shift = (k1 - r.a) / (N - nz);
        for (long i = tid; i < len; i += PF_THREADS)
            if (!(fl[i] & 1)) v[i] += shift;
    }
    for (long i = tid; i < len; i += PF_THREADS) x[i] = (TIO)((fl[i] & 2) ? -v[i] : v[i]);
    if (usediters && tid == 0) usediters[blockIdx.x] = j + 1;
}

// ---- the same projection with every vector split over ranks (column-sharded H of nmfsc) --------------
// projfunc.m:22-53 as four phases; between phases the caller all-reduces red[4*count] (sum over ranks), so every rank sees the
// same sums and takes the same branches.  state[2k] = |Z| of vector k (global), state[2k+1] = 1 once all(v >= 0) held.
__global__ __launch_bounds__(PF_THREADS) void pfd_init_kernel(const float *X, long len, int nn, double *V, unsigned char *F, double *red, double *state) {
    __shared__ double sred[PF_WAVES * 4];
    const long k = blockIdx.x;
    const float *x = X + len * k;
    double *v = V + len * k;
    unsigned char *fl = F + len * k;
    Red4 r = {0.0, 0.0, 0.0, 0.0};
    for (long i = threadIdx.x; i < len; i += PF_THREADS) {
        double s = (double)x[i];
        unsigned char f = 0;
        if (!nn) { if (s < 0) f = 2; s = fabs(s); }                               // projfunc.m:16-19
        v[i] = s; fl[i] = f; r.a += s;
    }
    r = block_red4(r, sred);
    if (threadIdx.x == 0) { red[4 * k] = r.a; red[4 * k + 1] = 0.0; red[4 * k + 2] = 0.0; red[4 * k + 3] = 0.0; state[2 * k] = 0.0; state[2 * k + 1] = 0.0; }
}
// in: red = {sum(v), |Z|} (global).  v += (k1 - sum)/(N - |Z|) off Z (projfunc.m:22 / 52-53); out: red = {w'w, w'v, v'v} (31-36)
__global__ __launch_bounds__(PF_THREADS) void pfd_shift_sums_kernel(double *V, const unsigned char *F, long len, double N, double k1, double *red, double *state) {
    __shared__ double sred[PF_WAVES * 4];
    const long k = blockIdx.x;
    const bool done = state[2 * k + 1] != 0.0;
    const double sum = red[4 * k], nz = red[4 * k + 1];
    Red4 r = {0.0, 0.0, 0.0, 0.0};
    if (!done) {
        double *v = V + len * k;
        const unsigned char *fl = F + len * k;
        const double shift = (k1 - sum) / (N - nz), mid = k1 / (N - nz);
        for (long i = threadIdx.x; i < len; i += PF_THREADS) {
            const bool z = fl[i] & 1;
            const double vi = z ? v[i] : v[i] + shift;
            if (!z) v[i] = vi;
            const double w = vi - (z ? 0.0 : mid);
            r.a += w * w; r.b += w * vi; r.c += vi * vi;
        }
    }
    r = block_red4(r, sred);                                                      // also orders the reads of red above before the writes below
    if (threadIdx.x == 0) {
        if (!done) state[2 * k] = nz;
        red[4 * k] = r.a; red[4 * k + 1] = r.b; red[4 * k + 2] = r.c; red[4 * k + 3] = 0.0;
    }
}
// in: red = {w'w, w'v, v'v} (global).  v += alphap*w (projfunc.m:37-38); out: red = {#(v < 0 or NaN)} for the all(v>=0) test (40)
__global__ __launch_bounds__(PF_THREADS) void pfd_step_kernel(double *V, const unsigned char *F, long len, double N, double k1, double k2, double *red, const double *state) {
    __shared__ double sred[PF_WAVES * 4];
    const long k = blockIdx.x;
    const bool done = state[2 * k + 1] != 0.0;
    const double a = red[4 * k], b = 2.0 * red[4 * k + 1], c = red[4 * k + 2] - k2;
    Red4 r = {0.0, 0.0, 0.0, 0.0};
    if (!done) {
        double *v = V + len * k;
        const unsigned char *fl = F + len * k;
        const double disc = b * b - 4.0 * a * c;
        const double alphap = (-b + (disc > 0.0 ? sqrt(disc) : 0.0)) / (2.0 * a);  // real(sqrt(.)), projfunc.m:37
        const double mid = k1 / (N - state[2 * k]);
        for (long i = threadIdx.x; i < len; i += PF_THREADS) {
            const double vi = v[i], w = vi - ((fl[i] & 1) ? 0.0 : mid);
            const double vn = alphap * w + vi;
            v[i] = vn;
            if (!(vn >= 0.0)) r.a += 1.0;
        }
    }
    r = block_red4(r, sred);
    if (threadIdx.x == 0) { red[4 * k] = r.a; red[4 * k + 1] = 0.0; red[4 * k + 2] = 0.0; red[4 * k + 3] = 0.0; }
}
// in: red = {#negative} (global): 0 finishes the vector (projfunc.m:40-44); else Z = {v <= 0}, v(Z) = 0; out: red = {sum(v), |Z|} (49-51)
__global__ __launch_bounds__(PF_THREADS) void pfd_zero_kernel(double *V, unsigned char *F, long len, double *red, double *state) {
    __shared__ double sred[PF_WAVES * 4];
    const long k = blockIdx.x;
    const bool was_done = state[2 * k + 1] != 0.0;
    const bool done = was_done || red[4 * k] == 0.0;
    Red4 r = {0.0, 0.0, 0.0, 0.0};
    if (!done) {
        double *v = V + len * k;
        unsigned char *fl = F + len * k;
        for (long i = threadIdx.x; i < len; i += PF_THREADS) {
            double vi = v[i];
            unsigned char f = fl[i] & 2;
            if (vi <= 0.0) { f |= 1; vi = 0.0; v[i] = 0.0; r.b += 1.0; }
            fl[i] = f;
            r.a += vi;
        }
    }
    r = block_red4(r, sred);
    if (threadIdx.x == 0) {
        if (done) state[2 * k + 1] = 1.0;
        red[4 * k] = r.a; red[4 * k + 1] = r.b; red[4 * k + 2] = 0.0; red[4 * k + 3] = 0.0;
    }
}
__global__ __launch_bounds__(PF_THREADS) void pfd_store_kernel(float *X, const double *V, const unsigned char *F, long len) {
    const long k = blockIdx.x;
    for (long i = threadIdx.x; i < len; i += PF_THREADS) X[len * k + i] = (float)((F[len * k + i] & 2) ? -V[len * k + i] : V[len * k + i]);   // projfunc.m:58-60
}

nmfx_status projfunc_cols_dist(hipStream_t st, float *X, long len, int count, long N_total, double k1, double k2, int nn, const Comm &comm,
                               double *v_scratch, unsigned char *flags, double *red) {
    if (count <= 0 || len <= 0) return NMFX_OK;
    const dim3 g(count), b(PF_THREADS);
    double *state = red + 4L * count;
    const double N = (double)N_total;
    std::vector<double> host(4 * (size_t)count);
    hipLaunchKernelGGL(pfd_init_kernel, g, b, 0, st, X, len, nn, v_scratch, flags, red, state);
    NMFX_HIP(hipGetLastError());
    nmfx_status rc = comm.allreduce(red, 4L * count, NMFX_F64, NMFX_REDUCE_SUM);
    if (rc != NMFX_OK) return rc;
    for (int j = 0; j <= PF_MAX_ITERS; ++j) {
        hipLaunchKernelGGL(pfd_shift_sums_kernel, g, b, 0, st, v_scratch, flags, len, N, k1, red, state);
        NMFX_HIP(hipGetLastError());
        if ((rc = comm.allreduce(red, 4L * count, NMFX_F64, NMFX_REDUCE_SUM)) != NMFX_OK) return rc;
        hipLaunchKernelGGL(pfd_step_kernel, g, b, 0, st, v_scratch, flags, len, N, k1, k2, red, state);
        NMFX_HIP(hipGetLastError());
        if ((rc = comm.allreduce(red, 4L * count, NMFX_F64, NMFX_REDUCE_SUM)) != NMFX_OK) return rc;
        NMFX_HIP(hipMemcpyAsync(host.data(), red, sizeof(double) * host.size(), hipMemcpyDeviceToHost, st));
        NMFX_HIP(hipStreamSynchronize(st));
        bool all_done = true;                                                    // identical on every rank: red is the all-reduced copy
        for (int k = 0; k < count; ++k) all_done &= host[4 * (size_t)k] == 0.0;
        if (all_done) break;
        hipLaunchKernelGGL(pfd_zero_kernel, g, b, 0, st, v_scratch, flags, len, red, state);
        NMFX_HIP(hipGetLastError());
        if ((rc = comm.allreduce(red, 4L * count, NMFX_F64, NMFX_REDUCE_SUM)) != NMFX_OK) return rc;
    }
    hipLaunchKernelGGL(pfd_store_kernel, g, b, 0, st, X, v_scratch, flags, len);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

template <typename TIO>
static nmfx_status projfunc_cols_t(hipStream_t st, TIO *X, long len, int count, double k1, double k2, int nn, int *usediters_dev, const float *dir, float mu) {
    if (count <= 0 || len <= 0) return NMFX_OK;
    // elements per thread: registers first (<= 16 doubles: no spills under the 128-VGPR budget of a 1024-thread block), then LDS
    if (len <= 4L * PF_THREADS) return launch_pf<4, 0, TIO>(st, X, len, count, k1, k2, nn, usediters_dev, dir, mu);
    if (len <= 8L * PF_THREADS) return launch_pf<8, 0, TIO>(st, X, len, count, k1, k2, nn, usediters_dev, dir, mu);
    if (len <= 16L * PF_THREADS) return launch_pf<16, 0, TIO>(st, X, len, count, k1, k2, nn, usediters_dev, dir, mu);
    if (len <= 24L * PF_THREADS) return launch_pf<16, 8, TIO>(st, X, len, count, k1, k2, nn, usediters_dev, dir, mu);
    if (len <= 32L * PF_THREADS) return launch_pf<16, 16, TIO>(st, X, len, count, k1, k2, nn, usediters_dev, dir, mu);
    // longer than registers + LDS hold: global fp64 working rows (allocated per call; this is the rare path)
    double *scratch = nullptr;
    unsigned char *flags = nullptr;
    NMFX_HIP(hipMalloc(&scratch, sizeof(double) * (size_t)len * count));
    hipError_t e2 = hipMalloc(&flags, (size_t)len * count);
    if (e2 != hipSuccess) { (void)hipFree(scratch); set_error("projfunc: hipMalloc failed: %s", hipGetErrorString(e2)); return NMFX_ERR_NOMEM; }
    hipLaunchKernelGGL(projfunc_long_kernel<TIO>, dim3(count), dim3(PF_THREADS), 0, st, X, len, k1, k2, nn, usediters_dev, scratch, flags, dir, mu);
    hipError_t e3 = hipGetLastError();
    if (e3 == hipSuccess) e3 = hipStreamSynchronize(st);
    (void)hipFree(scratch); (void)hipFree(flags);
    if (e3 != hipSuccess) { set_error("projfunc_long: %s", hipGetErrorString(e3)); return NMFX_ERR_HIP; }
    return NMFX_OK;
}

nmfx_status projfunc_cols(hipStream_t st, float *X, long len, int count, double k1, double k2, int nn, int *usediters_dev, const float *dir, float mu) {
    return projfunc_cols_t<float>(st, X, len, count, k1, k2, nn, usediters_dev, dir, mu);
}
nmfx_status projfunc_cols_f64(hipStream_t st, double *X, long len, int count, double k1, double k2, int nn, int *usediters_dev) {
    return projfunc_cols_t<double>(st, X, len, count, k1, k2, nn, usediters_dev, nullptr, 0.0f);
}

}  // namespace nmfx

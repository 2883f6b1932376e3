// Hoyer's L1/L2 projection (projfunc.m:13-65) for `count` vectors stored as the columns of X
// (len x count, column-major), in place.  One 1024-thread workgroup per vector; the vector lives in
// registers (+ LDS beyond 16 elements per thread) as fp64 for the whole iteration (len <= 32768), every reduction (sum, w'w, w'v, v'v,
// |Z|, all(v>=0)) is a wave-shuffle + LDS tree in fp64 so the discrete branches (v<=0 sets,
// nmfsc.m:164 objective test downstream) follow the float64 reference.  Bandwidth-class: the only
// HBM traffic is one read and one write of the vector.
#include <vector>

#include <cstdlib>
#include "nmfx_internal.h"

namespace nmfx {

constexpr int PF_THREADS = 1024;
constexpr int PF_WAVES = PF_THREADS / 64;
constexpr int PF_MAX_ITERS = 100000;
#ifndef PF_GROUP
#define PF_GROUP 8
#endif  // safety cap: the reference loops forever on NaN input

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

// ---- block reduction of four fp64 sums without LDS shuffles -----------------------------------------------------------------
// A projection needs two or three dependent block reductions per inner iteration (projfunc.m:34-36, 40, 51), so their latency, not
// HBM, bounds the kernel.  The round-1 version (ds_bpermute butterflies + every thread re-reading all WAVES*4 partials from LDS)
// cost ~3.7 us per reduction with 16 waves; this one keeps the intra-wave part in the VALU (DPP row operations on the two halves
// of each double) and reads one partial per lane.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {   // lanes without a source lane, and rows outside ROW_MASK, receive 0
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi2, lo2);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double wave_sum_f64(double v) {   // sum over the 64 lanes, returned wave-uniform
    v += dpp_f64<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]      -> every lane: sum of its quad
    v += dpp_f64<0x141, 0xf>(v);   // row_half_mirror          -> sum of its 8 lanes
    v += dpp_f64<0x140, 0xf>(v);   // row_mirror               -> sum of its row of 16
    v += dpp_f64<0x142, 0xa>(v);   // row_bcast:15 into rows 1, 3: rows 0+1 | rows 2+3
    v += dpp_f64<0x143, 0xc>(v);   // row_bcast:31 into rows 2, 3: row 3 holds the wave total
    return readlane_f64(v, 63);
}

template <int WAVES>
__device__ __forceinline__ Red4 block_red4_w(Red4 v, double *red) {
    static_assert(WAVES * 4 <= 64 && (WAVES & (WAVES - 1)) == 0, "one partial per lane");
    const double a = wave_sum_f64(v.a), b = wave_sum_f64(v.b), c = wave_sum_f64(v.c), d = wave_sum_f64(v.d);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();                                            // the previous reduction's readers are done with red[]
    if (lane == 0) { red[wave * 4 + 0] = a; red[wave * 4 + 1] = b; red[wave * 4 + 2] = c; red[wave * 4 + 3] = d; }
    __syncthreads();
    // lane l holds partial (wave l/4, sum l%4); add the lanes that share l%4: strides 4, 8 inside a row, then rows
    double t = lane < WAVES * 4 ? red[lane] : 0.0;
    t += dpp_f64<0x114, 0xf>(t);   // row_shr:4
    t += dpp_f64<0x118, 0xf>(t);   // row_shr:8               -> lanes 12..15 of a row: the row's four sums
    if (WAVES > 4) t += __shfl_xor(t, 16);                      // rows pair up (plain permutes: one value, two steps at most)
    if (WAVES > 8) t += __shfl_xor(t, 32);
    Red4 r;
    r.a = readlane_f64(t, 12); r.b = readlane_f64(t, 13); r.c = readlane_f64(t, 14); r.d = readlane_f64(t, 15);
    return r;
}

// Working vector: element e of thread tid is x[tid + e*THREADS].  The first ER elements per thread live in registers (fp64, 2 VGPRs
// each), the next EL in LDS (fp64, [e][tid]: conflict-free b64 accesses).  <1024, 16, 0> covers len <= 16384 inside the 128-VGPR budget
// of a 1024-thread block; <512, 64, 0> covers len <= 32768 (a row of H at BASELINE config 5) inside the 256-VGPR budget of a
// 512-thread block -- the round-1 <1024 threads, 32 elements> variant spilled 212 VGPRs to scratch there.
// TIO = float (engine buffers) or double (nmfx_projfunc on float64 input: no fp32 rounding anywhere).
// dir != nullptr fuses the line-search step into the load: s = x + mu*dir in fp64 (nmfsc.m:154; rounding the stepped vector -- or mu -- to fp32
// first costs parity where the projection amplifies: H off by 4e-6 on a K = 3 problem from that rounding alone).  dir64: the direction as doubles.
// Two block reductions per inner iteration instead of the three a literal transcription needs: the pass that applies
// v = alpha*w + v (projfunc.m:38) also gathers what lines 49-51 would need if the loop goes on -- |{v <= 0}| and the sum of the
// entries that survive the zeroing (the zeros add exactly 0.0) -- and the zeroing + redistribution of lines 50-53 is applied
// element-wise at the top of the next sweep.
template <int THREADS, int ER, int EL, typename TIO, int DM>   // DM: 0 no step | 1 fp32 direction | 2 float64 direction -- compile-time: with run-time branches hipcc
__global__ __launch_bounds__(THREADS) void projfunc_kernel(TIO *X,   /* joins all EPT loaded values in PHIs and processes them after the join: twice the live set, 50 spilled VGPRs */ long len, double k1, double k2, int nn, int *usediters, const float *dir, double mu, const TIO *src, const double *dir64) {
    constexpr int WAVES = THREADS / 64;
    __shared__ double red[WAVES * 4];
    extern __shared__ __attribute__((aligned(16))) double vl[];   // [EL][THREADS]
    constexpr int EPT = ER + EL;
    // raw buffer accesses: one 32-bit lane offset + an immediate per element (no 64-bit address pair per element and pointer, which
    // is what pushed the unrolled loads over the register budget), and the hardware bounds check stands in for the tail predicate:
    // loads past the end of the vector return 0, stores there are dropped
    const unsigned vbytes = (unsigned)(len * (long)sizeof(TIO));
    const __amdgpu_buffer_rsrc_t xo_srd = __builtin_amdgcn_make_buffer_rsrc((void *)(X + len * blockIdx.x), 0, (int)vbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xi_srd = __builtin_amdgcn_make_buffer_rsrc((void *)((src ? src : X) + len * blockIdx.x), 0, (int)vbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t dx_srd = DM == 2 ? __builtin_amdgcn_make_buffer_rsrc((void *)(dir64 + len * blockIdx.x), 0, (int)(unsigned)(len * 8), 0x00020000)
                                              : __builtin_amdgcn_make_buffer_rsrc((void *)(dir ? dir + len * blockIdx.x : (const float *)X), 0, dir ? (int)(unsigned)(len * 4) : 0, 0x00020000);
    auto ld = [&](const __amdgpu_buffer_rsrc_t srd, int e) -> double {
        const int voff = (int)(threadIdx.x * sizeof(TIO)), ioff = e * THREADS * (int)sizeof(TIO);
        if (sizeof(TIO) == 4) return (double)__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd, voff, ioff, 0));
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(srd, voff, ioff, 0));
    };
    const int tid = threadIdx.x;
    const double N = (double)len;
    // element e of this thread exists iff e < nfull, or e == nfull and tid < rem (wave-uniform tests except the last)
    const int nfull = (int)(len / THREADS), rem = (int)(len - (long)nfull * THREADS);
    auto valid = [&](int e) -> bool { return e < nfull || (e == nfull && tid < rem); };
    double vr[ER > 0 ? ER : 1];
    unsigned zm[(EPT + 31) / 32], ngm[(EPT + 31) / 32];           // Z membership / original sign, one bit per element
#pragma unroll
    for (int q = 0; q < (EPT + 31) / 32; ++q) zm[q] = ngm[q] = 0u;
    auto get = [&](int e) -> double { return e < ER ? vr[e < ER ? e : 0] : vl[(e - ER) * THREADS + tid]; };
    auto put = [&](int e, double val) { if (e < ER) vr[e < ER ? e : 0] = val; else vl[(e - ER) * THREADS + tid] = val; };

    // Everything below is branch-free per element (selects, not ifs): with divergent control flow around each element hipcc emits
    // one exec-masked block per element and the 64 dependent chains of a thread run one after the other (measured: 280 cycles per
    // element and sweep); as straight-line code they interleave.  Slots past the end of the vector are permanent members of Z with
    // value 0: they add exact zeros to every sum and are taken out of the |Z| count (n_pad).
    const double n_pad = (double)((long)THREADS * EPT - len);
    Red4 r = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const bool ok = valid(e);
        double s = ld(xi_srd, e);
        if (DM == 2) {
            s = fma(mu, __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(dx_srd, (int)(threadIdx.x * 8), e * THREADS * 8, 0)), s);
        } else if (DM == 1) {
            s = fma(mu, (double)__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dx_srd, (int)(threadIdx.x * 4), e * THREADS * 4, 0)), s);
        }
        const bool neg = !nn && s < 0;                                     // projfunc.m:16-19
        ngm[e >> 5] |= neg ? (1u << (e & 31)) : 0u;
        zm[e >> 5] |= ok ? 0u : (1u << (e & 31));
        s = nn ? s : fabs(s);
        put(e, s);
        r.a += s;
        if ((e & (PF_GROUP - 1)) == PF_GROUP - 1) __builtin_amdgcn_sched_barrier(0);
    }
    // pin the sign bits NOW: left alone, hipcc re-derives them at the store phase from the loaded values, which then stay live through
    // the whole kernel (one VGPR per element for fp32 input, two -- 50 of them spilled -- once the values are doubles: stepped or fp64 input)
#pragma unroll
    for (int q = 0; q < (EPT + 31) / 32; ++q) asm volatile("" : "+v"(ngm[q]));
    r = block_red4_w<WAVES>(r, red);
    double shift = (k1 - r.a) / N;                                         // projfunc.m:22
    bool zero_first = false;                                               // the pending element-wise step: v += shift off Z (first: everywhere)
    double nz = 0.0;
    int j = 0;
    for (;;) {
        const double mid = k1 / (N - nz);                                  // projfunc.m:31-32
        r.a = r.b = r.c = r.d = 0.0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const unsigned bit = 1u << (e & 31);
            double ve = get(e);
            const bool z = zero_first ? (ve <= 0.0) : ((zm[e >> 5] & bit) != 0u);   // projfunc.m:49 (first sweep: only the padding)
            ve = z ? 0.0 : ve + shift;                                     // projfunc.m:50, 53 | 22, 52   (padding and old zeros are 0 already)
            zm[e >> 5] = (zm[e >> 5] & ~bit) | (z ? bit : 0u);
            put(e, ve);
            const double w = ve - (z ? 0.0 : mid);                         // projfunc.m:33
            r.a += w * w;                                                  // projfunc.m:34
            r.b += w * ve;                                                 // projfunc.m:35
            r.c += ve * ve;                                                // projfunc.m:36
            if ((e & (PF_GROUP - 1)) == PF_GROUP - 1) __builtin_amdgcn_sched_barrier(0);   // interleave PF_GROUP chains, not all EPT: bounds the live ranges
        }
        r = block_red4_w<WAVES>(r, red);
        const double a = r.a, b = 2.0 * r.b, c = r.c - k2;
        const double disc = b * b - 4.0 * a * c;
        const double alphap = (-b + (disc > 0.0 ? sqrt(disc) : 0.0)) / (2.0 * a);   // projfunc.m:37 real(sqrt(.))
        r.a = r.b = r.c = r.d = 0.0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const double ve = get(e);
            const double w = ve - ((zm[e >> 5] & (1u << (e & 31))) ? 0.0 : mid);
            const double vn = alphap * w + ve;                             // projfunc.m:38
            put(e, vn);
            r.a += (vn >= 0.0) ? 0.0 : 1.0;                                // projfunc.m:40 all(v>=0)  (NaN counts as a failure)
            r.b += (vn <= 0.0) ? 1.0 : 0.0;                                // |Z| of projfunc.m:49, should the loop go on
            r.c += (vn <= 0.0) ? 0.0 : vn;                                 // sum(v) after v(Z) = 0, projfunc.m:51
            if ((e & (PF_GROUP - 1)) == PF_GROUP - 1) __builtin_amdgcn_sched_barrier(0);
        }
        r = block_red4_w<WAVES>(r, red);
        if (r.a == 0.0 || j >= PF_MAX_ITERS) break;                        // projfunc.m:40-44
        ++j;                                                               // projfunc.m:46
        nz = r.b - n_pad;
        shift = (k1 - r.c) / (N - nz);                                     // projfunc.m:52
        zero_first = true;
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const double o = (ngm[e >> 5] & (1u << (e & 31))) ? -get(e) : get(e);             // projfunc.m:58-60
        const int voff = (int)(threadIdx.x * sizeof(TIO)), ioff = e * THREADS * (int)sizeof(TIO);
        if (sizeof(TIO) == 4) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)o), xo_srd, voff, ioff, 0);
        else {
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), xo_srd, voff, ioff, 0);
        }
    }
    if (usediters && tid == 0) usediters[blockIdx.x] = j + 1;
}

template <int THREADS, int ER, int EL, typename TIO, int DM>
static nmfx_status launch_pf_dm(hipStream_t st, TIO *X, long len, int count, double k1, double k2, int nn, int *usediters, const float *dir, double mu, const TIO *src, const double *dir64) {
    auto kern = projfunc_kernel<THREADS, ER, EL, TIO, DM>;
    const size_t ldsb = sizeof(double) * EL * THREADS;
    static LdsAttrOnce lds_attr;
    TRY(lds_attr.set(reinterpret_cast<const void *>(kern), (int)ldsb));
    hipLaunchKernelGGL(kern, dim3(count), dim3(THREADS), ldsb, st, X, len, k1, k2, nn, usediters, dir, mu, src, dir64);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}
template <int THREADS, int ER, int EL, typename TIO>
static nmfx_status launch_pf(hipStream_t st, TIO *X, long len, int count, double k1, double k2, int nn, int *usediters, const float *dir, double mu, const TIO *src, const double *dir64) {
    if constexpr (sizeof(TIO) == 4) {
        if (dir64) return launch_pf_dm<THREADS, ER, EL, TIO, 2>(st, X, len, count, k1, k2, nn, usediters, nullptr, mu, src, dir64);
        if (dir) return launch_pf_dm<THREADS, ER, EL, TIO, 1>(st, X, len, count, k1, k2, nn, usediters, dir, mu, src, nullptr);
    } else if (dir || dir64) {
        set_error("projfunc: the fused line-search step needs fp32 vectors");
        return NMFX_ERR_INVALID;
    }
    return launch_pf_dm<THREADS, ER, EL, TIO, 0>(st, X, len, count, k1, k2, nn, usediters, nullptr, 0.0, src, nullptr);
}

// Any length: the working vector lives in a global fp64 scratch row (L2-resident for realistic sizes) instead of registers.
template <typename TIO>
__global__ __launch_bounds__(PF_THREADS) void projfunc_long_kernel(TIO *X, long len, double k1, double k2, int nn, int *usediters, double *scratch,
                                                                  unsigned char *flags, const float *dir, double mu, const double *dir64) {
    __shared__ double red[PF_WAVES * 4];
    TIO *x = X + len * blockIdx.x;
    const float *dx = dir ? dir + len * blockIdx.x : nullptr;
    const double *dx64 = dir64 ? dir64 + len * blockIdx.x : nullptr;
    double *v = scratch + len * blockIdx.x;
    unsigned char *fl = flags + len * blockIdx.x;   // bit0: in Z, bit1: was negative
    const int tid = threadIdx.x;
    const double N = (double)len;
    Red4 r = {0.0, 0.0, 0.0, 0.0};
    for (long i = tid; i < len; i += PF_THREADS) {
        double s = (double)x[i];
        if (sizeof(TIO) == 4 && dx64) s = fma(mu, dx64[i], s);
        else if (sizeof(TIO) == 4 && dx) s = fma(mu, (double)dx[i], s);
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
__global__ __launch_bounds__(PF_THREADS) void pfd_init_kernel(const float *X, long len, int nn, double *V, unsigned char *F, double *red, double *state,
                                                              const float *dir, const double *dir64, double mu) {   // the vectors projected: X + mu*dir (fp64)
    __shared__ double sred[PF_WAVES * 4];
    const long k = blockIdx.x;
    const float *x = X + len * k;
    double *v = V + len * k;
    unsigned char *fl = F + len * k;
    Red4 r = {0.0, 0.0, 0.0, 0.0};
    for (long i = threadIdx.x; i < len; i += PF_THREADS) {
        double s = (double)x[i];
        if (dir64) s = fma(mu, dir64[len * k + i], s);
        else if (dir) s = fma(mu, (double)dir[len * k + i], s);
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
                               double *v_scratch, unsigned char *flags, double *red, const float *dir, double mu, const float *src, const double *dir64) {
    if (count <= 0 || len <= 0) return NMFX_OK;
    const dim3 g(count), b(PF_THREADS);
    double *state = red + 4L * count;
    const double N = (double)N_total;
    std::vector<double> host(4 * (size_t)count);
    StreamDrain drain_(st);   // the read-backs below land in `host`: nothing may still be in flight into it when it dies, whichever way this function is left
    hipLaunchKernelGGL(pfd_init_kernel, g, b, 0, st, src ? src : X, len, nn, v_scratch, flags, red, state, dir, dir64, mu);
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
static nmfx_status projfunc_cols_t(hipStream_t st, TIO *X, long len, int count, double k1, double k2, int nn, int *usediters_dev, const float *dir, double mu,
                                   const TIO *src, const double *dir64) {
    if (count <= 0 || len <= 0) return NMFX_OK;
    // elements per thread: registers first, then LDS (see projfunc_kernel)
    if (len <= 1024L) return launch_pf<256, 4, 0, TIO>(st, X, len, count, k1, k2, nn, usediters_dev, dir, mu, src, dir64);
    if (len <= 4096L) return launch_pf<1024, 4, 0, TIO>(st, X, len, count, k1, k2, nn, usediters_dev, dir, mu, src, dir64);
    if (len <= 8192L) return launch_pf<1024, 8, 0, TIO>(st, X, len, count, k1, k2, nn, usediters_dev, dir, mu, src, dir64);
    if (len <= 16384L) return launch_pf<1024, 16, 0, TIO>(st, X, len, count, k1, k2, nn, usediters_dev, dir, mu, src, dir64);
    if (len <= 24576L) return launch_pf<512, 48, 0, TIO>(st, X, len, count, k1, k2, nn, usediters_dev, dir, mu, src, dir64);
    // (measured at 128 x 32768: <512, 64, 0> 0.0795 ms, <512, 48 + 16 in LDS> 0.0897, <1024, 32, 0> 0.0892)
    if (len <= 32768L) return launch_pf<512, 64, 0, TIO>(st, X, len, count, k1, k2, nn, usediters_dev, dir, mu, src, dir64);
    if (len <= 40960L) return launch_pf<512, 48, 32, TIO>(st, X, len, count, k1, k2, nn, usediters_dev, dir, mu, src, dir64);
    // longer than registers + LDS hold: global fp64 working rows (allocated per call; this is the rare path)
    if (src && src != X) NMFX_HIP(hipMemcpyAsync(X, src, sizeof(TIO) * (size_t)len * count, hipMemcpyDeviceToDevice, st));
    double *scratch = nullptr;
    unsigned char *flags = nullptr;
    NMFX_HIP(hipMalloc(&scratch, sizeof(double) * (size_t)len * count));
    hipError_t e2 = hipMalloc(&flags, (size_t)len * count);
    if (e2 != hipSuccess) { (void)hipFree(scratch); set_error("projfunc: hipMalloc failed: %s", hipGetErrorString(e2)); return NMFX_ERR_NOMEM; }
    hipLaunchKernelGGL(projfunc_long_kernel<TIO>, dim3(count), dim3(PF_THREADS), 0, st, X, len, k1, k2, nn, usediters_dev, scratch, flags, dir, mu, dir64);
    hipError_t e3 = hipGetLastError();
    if (e3 == hipSuccess) e3 = hipStreamSynchronize(st);
    (void)hipFree(scratch); (void)hipFree(flags);
    if (e3 != hipSuccess) { set_error("projfunc_long: %s", hipGetErrorString(e3)); return NMFX_ERR_HIP; }
    return NMFX_OK;
}

nmfx_status projfunc_cols(hipStream_t st, float *X, long len, int count, double k1, double k2, int nn, int *usediters_dev, const float *dir, double mu,
                          const float *src, const double *dir64) {
    return projfunc_cols_t<float>(st, X, len, count, k1, k2, nn, usediters_dev, dir, mu, src, dir64);
}
nmfx_status projfunc_cols_f64(hipStream_t st, double *X, long len, int count, double k1, double k2, int nn, int *usediters_dev) {
    return projfunc_cols_t<double>(st, X, len, count, k1, k2, nn, usediters_dev, nullptr, 0.0, nullptr, nullptr);
}

}  // namespace nmfx

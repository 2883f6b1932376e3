// Fused two-stage MFMA kernel of the multiplicative update (gfx950), the product path for nmf().
//
// One pass over V computes, per 128-row block of the STATIONARY factor and per 64-wide tile of the
// STREAMED factor,
//     S^T = Y_tile * X_blk^T            (MFMA, contraction K)         -- the W*H product, never stored
//     R   = f(V_tile, S)                (VALU: V./S for KL, V for euclidean; cost terms accumulated)
//     O^T += Y_tile^T * R               (MFMA, contraction over the tile) -- (V./WH)*H'  or  W'*(V./WH)
// so V_hat never exists in HBM and V is streamed exactly once per half-iteration
// (nmf.m:152-153 + 168 for the W step, nmf.m:183-184 + 199 for the H step).
//
//   W step (K2):  X = W (rows i), Y rows = columns of H,   V tile read with lanes along i (contiguous)
//   H step (K3):  X = H^T (rows j), Y rows = rows of W (from the W^T copy), V tile read with lanes along j;
//                 the epilogue applies H <- H .* G ./ max(den + lambda, eps) in place (nmf.m:199)
//
// Register-stationary design for one wave per SIMD (256 threads, 1 workgroup / CU, 512 VGPRs):
//   xreg[K/2]   the wave's 32 rows of X as MFMA B-port operands        (128 VGPRs at K = 256)
//   acc[K/32]   32x32 accumulators of O^T                              (128 VGPRs)
//   sacc[2]     the 64x32 S^T tile; overwritten in place by R, whose C-layout registers ARE valid
//               B-port operands of the second contraction (no LDS round trip, no shuffles)
// The streamed tile arrives by LDS-DMA (global_load_lds_dwordx4: one 1-KiB row of K floats per
// wave-instruction, no VGPRs) into a double buffer with row stride K+4 floats: ds_read_b128 down a
// column (first contraction) and ds_read_b32 along a row (second contraction) are both conflict-free.
// v_mfma_f32_32x32x2_f32 issues every 64 cycles per SIMD; each MFMA needs at most one ds_read.
#include "nmfx_internal.h"

namespace nmfx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int FT_ROWS = 128;  // stationary rows per workgroup (32 per wave)
constexpr int FT_C = 64;      // streamed rows (contraction tile of the second product) per step

__device__ __forceinline__ int rowmap(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// FUNC: 0 R=V, no S | 1 R=V, S only for the euclidean cost | 2 R=V./S (KL) | 3 R=V./S + KL cost
template <int K, bool D_RC, int FUNC, bool DO_G2, int EPI>
__global__ __launch_bounds__(256, 1) void fused_kernel(const FusedParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int LDY = K + 4;
    constexpr int NKB = K / 32;
    constexpr int BUF = FT_C * LDY;
    constexpr bool NEED_S = FUNC != 0;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const long r0 = (long)blockIdx.x * FT_ROWS + 32 * w;   // this wave's rows r0 .. r0+31
    const long r = r0 + l31;
    const long cbeg = (long)blockIdx.y * p.c_per_split;
    const long cend = cbeg + p.c_per_split < p.Cn ? cbeg + p.c_per_split : p.Cn;
    const int ntiles = (int)((cend - cbeg) / FT_C);

    // stationary operand: B-port register s holds X(r, k = 8*(s>>2) + 4*h + (s&3))
    float xreg[NEED_S ? K / 2 : 1];
    if (NEED_S) {
#pragma unroll
        for (int s = 0; s < K / 2; ++s) xreg[s] = p.X[r * p.xs_r + (long)(8 * (s >> 2) + 4 * h + (s & 3)) * p.xs_k];
    }

    f32x16 acc[DO_G2 ? NKB : 1];
#pragma unroll
    for (int kb = 0; kb < (DO_G2 ? NKB : 1); ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[kb][e] = 0.0f;

    // LDS-DMA of streamed tile t into buffer b: wave w moves rows w, w+4, ... (K floats = K/4 lanes x 16 B each)
    auto dma_tile = [&](int t, int b) {
        const float *src = p.Y + (cbeg + (long)t * FT_C) * K + lane * 4;
        float *dst = lds + b * BUF;
        if (lane < K / 4) {
#pragma unroll
            for (int c = 0; c < FT_C / 4; ++c) {
                const int row = w + 4 * c;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (long)row * K),
                                                 (__attribute__((address_space(3))) void *)(dst + row * LDY), 16, 0, 0);
            }
        }
    };
    // V tile of step t: d[jb*16 + reg] = V(r, c = c0 + 32*jb + rowmap(reg, h)).  Addresses are split into a wave-uniform
    // part (scalar registers / immediates) and one per-lane offset computed once.
    float d[32];
    const int lane_off = (int)(r + 4 * h * p.ldd);                       // D_RC: element offset inside a column block
    const float *lane_ptr = p.D + p.ldd * r + 4 * h + cbeg;              // !D_RC: this lane's row of V', advanced per tile
    auto load_d = [&](int t) {
        if (D_RC) {
            const float *ub = p.D + p.ldd * (cbeg + (long)t * FT_C);      // wave-uniform
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) d[jb * 16 + reg] = (ub + p.ldd * (32 * jb + (reg & 3) + 8 * (reg >> 2)))[lane_off];
        } else {
            const float *lp = lane_ptr + (long)t * FT_C;
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4 *>(lp + 32 * jb + 8 * q);
                    d[jb * 16 + 4 * q + 0] = v.x; d[jb * 16 + 4 * q + 1] = v.y; d[jb * 16 + 4 * q + 2] = v.z; d[jb * 16 + 4 * q + 3] = v.w;
                }
        }
    };

    double cost = 0.0;
    if (ntiles > 0) {
        dma_tile(0, 0);
        load_d(0);
    }
    for (int t = 0; t < ntiles; ++t) {
        const int b = t & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own DMA rows of tile t (and d) have landed
        __syncthreads();                                     // everyone's rows landed; buffer b^1 is free again
        if (t + 1 < ntiles) dma_tile(t + 1, b ^ 1);
        const float *Yt = lds + b * BUF;

        f32x16 sacc[2];
        if (NEED_S) {
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int e = 0; e < 16; ++e) sacc[jb][e] = 0.0f;
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
#pragma unroll
                for (int g = 0; g < K / 8; ++g) {
                    const float4 a = *reinterpret_cast<const float4 *>(Yt + (32 * jb + l31) * LDY + 8 * g + 4 * h);
                    sacc[jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, xreg[4 * g + 0], sacc[jb], 0, 0, 0);
                    sacc[jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, xreg[4 * g + 1], sacc[jb], 0, 0, 0);
                    sacc[jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, xreg[4 * g + 2], sacc[jb], 0, 0, 0);
                    sacc[jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, xreg[4 * g + 3], sacc[jb], 0, 0, 0);
                }
            }
        }
        // element map: R (the B-port operand of the second product) and the divergence terms of nmf.m:206-215
        float tc = 0.0f;
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const float v = d[jb * 16 + reg];
                if (FUNC >= 2) {
                    const float sv = sacc[jb][reg];
                    const float q = v / sv;                          // V ./ V_hat            nmf.m:152
                    if (FUNC == 3) tc += v * logf(q) - v + sv;       // nmf.m:210
                    sacc[jb][reg] = q;
                } else {
                    if (FUNC == 1) { const float e = v - sacc[jb][reg]; tc = fmaf(e, e, tc); }   // nmf.m:208
                    sacc[jb][reg] = v;
                }
            }
        cost += (double)tc;
        if (t + 1 < ntiles) load_d(t + 1);   // in flight under the second product
        if (DO_G2) {
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const float *yrow = Yt + (32 * jb + rowmap(reg, h)) * LDY + l31;
                    const float rr = sacc[jb][reg];
#pragma unroll
                    for (int kb = 0; kb < NKB; ++kb) acc[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(yrow[32 * kb], rr, acc[kb], 0, 0, 0);
                }
        }
    }

    // epilogue: acc[kb][reg] = O(k = 32*kb + rowmap(reg,h), r)
    if (DO_G2) {
        if (EPI == 0) {
            float *out = p.out + (long)blockIdx.y * p.slab_stride;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) out[r * p.os_r + (long)(32 * kb + rowmap(reg, h)) * p.os_k] = acc[kb][reg];
        } else {
            // H(k, j=r) <- H .* (G ./ max(den + lambda, eps))      nmf.m:199   (den: matrix K x n, or per-row vector for KL)
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int k = 32 * kb + rowmap(reg, h);
                    if (p.fix && p.fix[k]) continue;
                    const long idx = (long)k + (long)K * r;
                    const float den = p.den ? p.den[idx] : (float)p.denvec[k];
                    const float lam = p.lam ? p.lam[k] : 0.0f;
                    p.Hio[idx] = p.Hio[idx] * (acc[kb][reg] / fmaxf(den + lam, NMFX_EPS_F));
                }
        }
    }
    if (p.cost_partials) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cost += __shfl_xor(cost, o);
        double *red = reinterpret_cast<double *>(lds);
        __syncthreads();
        if (lane == 0) red[w] = cost;
        __syncthreads();
        if (tid == 0) p.cost_partials[(long)blockIdx.y * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    }
}

template <int K, bool D_RC, int FUNC, bool DO_G2, int EPI>
static nmfx_status launch_one(hipStream_t st, const FusedParams &p, int nsplit) {
    const size_t ldsb = sizeof(float) * 2 * FT_C * (K + 4);
    auto kern = fused_kernel<K, D_RC, FUNC, DO_G2, EPI>;
    static bool attr_done = false;
    if (!attr_done) {
        NMFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
        attr_done = true;
    }
    dim3 grid((unsigned)(p.R / FT_ROWS), (unsigned)nsplit);
    hipLaunchKernelGGL(kern, grid, dim3(256), ldsb, st, p);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

template <int K, bool D_RC, bool DO_G2, int EPI>
static nmfx_status launch_f(hipStream_t st, const FusedParams &p, int nsplit, int func) {
    switch (func) {
    case 0: if (DO_G2) return launch_one<K, D_RC, 0, DO_G2, EPI>(st, p, nsplit); break;
    case 1: return launch_one<K, D_RC, 1, DO_G2, EPI>(st, p, nsplit);
    case 2: if (DO_G2) return launch_one<K, D_RC, 2, DO_G2, EPI>(st, p, nsplit); break;
    case 3: return launch_one<K, D_RC, 3, DO_G2, EPI>(st, p, nsplit);
    }
    set_error("launch_fused: unsupported functor %d", func);
    return NMFX_ERR_UNSUPPORTED;
}

template <int K>
static nmfx_status launch_k(hipStream_t st, const FusedParams &p, int nsplit, bool d_rc, int func, bool do_g2, int epi) {
    if (!do_g2) return d_rc ? launch_f<K, true, false, 0>(st, p, nsplit, func) : NMFX_ERR_UNSUPPORTED;   // cost-only pass
    if (d_rc) return launch_f<K, true, true, 0>(st, p, nsplit, func);                                    // W step: slabs out
    if (epi == 1) return launch_f<K, false, true, 1>(st, p, nsplit, func);                               // H step, fused update
    return launch_f<K, false, true, 0>(st, p, nsplit, func);                                             // H step, slabs out
}

bool fused_supported(int K) { return K == 64 || K == 128 || K == 256; }

// nsplit: number of contraction ranges (grid.y); c_per_split must be a multiple of FT_C and R of FT_ROWS
nmfx_status launch_fused(hipStream_t st, const FusedParams &p, int nsplit, bool d_rc, int func, bool do_g2, int epi) {
    if (p.R % FT_ROWS || p.Cn % FT_C || p.c_per_split % FT_C || nsplit < 1) {
        set_error("launch_fused: shape not tileable (R=%ld Cn=%ld c_per_split=%ld)", p.R, p.Cn, p.c_per_split);
        return NMFX_ERR_INVALID;
    }
    switch (p.K) {
    case 64: return launch_k<64>(st, p, nsplit, d_rc, func, do_g2, epi);
    case 128: return launch_k<128>(st, p, nsplit, d_rc, func, do_g2, epi);
    case 256: return launch_k<256>(st, p, nsplit, d_rc, func, do_g2, epi);
    default: set_error("launch_fused: K=%d not supported", p.K); return NMFX_ERR_UNSUPPORTED;
    }
}

}  // namespace nmfx

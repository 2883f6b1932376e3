// Shared by gemm.hip (general kernel + host dispatch) and gemm_pipe.hip (pipelined kernel): view decoding, element maps,
// divergence terms.
#pragma once
#include "nmfx_internal.h"

namespace nmfx {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int NTHREADS = 256;

__device__ __forceinline__ void dec_r(const OpView &v, int r, long &off, int &g) {
    switch (v.mode) {
    case VIEW_RC: off = r; g = 0; break;
    case VIEW_HSTACK_RC: { int rr = r + v.lim; int t = rr / v.blk; int k = rr - t * v.blk; off = (long)k - v.ld * t; g = -t; } break;
    case VIEW_HSTACK_KC: off = v.ld * r; g = (v.tstride > 0 && r >= v.tstride) ? -(1 << 30) : r + v.lim + v.goff; break;
    case VIEW_XSHIFT_KC: off = v.ld * r; g = v.lim - 1 - r; break;
    default: off = v.ld * r; g = 0; break;  // VIEW_KC, VIEW_WSTACK_KC
    }
}
__device__ __forceinline__ void dec_k(const OpView &v, int kc, long &off, int &g) {
    switch (v.mode) {
    case VIEW_RC: off = v.ld * kc; g = 0; break;
    case VIEW_HSTACK_RC: off = v.ld * kc; g = kc + v.goff; break;
    case VIEW_HSTACK_KC: { int t = kc / v.blk; int k = kc - t * v.blk; off = (long)k - v.ld * t; g = -t; } break;
    case VIEW_WSTACK_KC: { int t = kc / v.blk; int i = kc - t * v.blk; off = (long)i + v.tstride * t; g = 0; } break;
    case VIEW_XSHIFT_KC: { int t = kc / v.blk; int i = kc - t * v.blk; off = (long)i + v.ld * t; g = -t; } break;
    default: off = kc; g = 0; break;  // VIEW_KC
    }
}

__device__ __forceinline__ float mpow(float x, float e) {   // MATLAB x.^e for the exponents that occur: exact for 0 and 1
    if (e == 0.0f) return 1.0f;
    if (e == 1.0f) return x;
    if (e == -1.0f) return 1.0f / x;
    return powf(x, e);
}
template <bool HEAVY>
__device__ __forceinline__ float pro1(int func, float x, float y, float e1 = 0.f, float e2 = 0.f) {
    if (HEAVY && func == NMFX_PRO_POWPROD) return mpow(x, e1) * mpow(y, e2);
    // v_rcp_f32 (1 ulp) instead of the IEEE division sequence, as in the fused kernels' element maps: these maps run once per element and
    // per output tile column while the operand is staged, and the correctly rounded division was a third of such a GEMM (K = 320 KL:
    // numerator product 3.29 ms against 2.37 ms for the plain contraction)
    switch (func) {
    case NMFX_PRO_RATIO: return x * __builtin_amdgcn_rcpf(y);
    case NMFX_PRO_RATIO_SQ: { const float r = __builtin_amdgcn_rcpf(y); return x * r * r; }
    case NMFX_PRO_RECIP2: return __builtin_amdgcn_rcpf(y);
    case NMFX_PRO_DIFF: return y - x;
    default: return x;
    }
}
template <bool HEAVY>
__device__ __forceinline__ float4 pro4(int func, float4 x, float4 y, float e1, float e2) {
    return make_float4(pro1<HEAVY>(func, x.x, y.x, e1, e2), pro1<HEAVY>(func, x.y, y.y, e1, e2), pro1<HEAVY>(func, x.z, y.z, e1, e2),
                       pro1<HEAVY>(func, x.w, y.w, e1, e2));
}

template <bool HEAVY>
__device__ __forceinline__ double div_term(int div, float v, float s, float al, float be) {
    if (HEAVY && div == NMFX_DIV_AB)   // nmf.m:214 (the trailing "+ beta" is the reference's)
        return (double)(powf(v, al) * powf(s, be)) - ((double)al * powf(v, al + be) + (double)be * powf(s, al + be) + (double)be) / ((double)al + (double)be);
    // v_rcp_f32 / v_log_f32 as in the fused kernels (same NaN / Inf pattern as the reference's expressions: 0*log(0) = NaN, x/0 = Inf): the
    // libm logf + IEEE division epilogue cost 1.2 ms on top of a 1.75 ms V_hat product (8192 x 32768, K = 320)
    switch (div) {
    case NMFX_DIV_KL: { const float q = v * __builtin_amdgcn_rcpf(s); return (double)(v * (0.6931471805599453f * __builtin_amdgcn_logf(q))) - (double)v + (double)s; }   // nmf.m:210
    case NMFX_DIV_IS: { const float q = v * __builtin_amdgcn_rcpf(s); return (double)(q - 0.6931471805599453f * __builtin_amdgcn_logf(q)) - 1.0; }                    // nmf.m:212  log(s/v) = -log(q)
    default: { float d = v - s; return (double)d * (double)d; }                     // nmf.m:208 (0.5 applied later)
    }
}

// the same terms in fp32, for the pipelined kernel's epilogue: 16 of them are summed in fp32 (one accumulator block of a thread) before the
// sum is promoted -- per-element conversions and fp64 adds made the KL cost epilogue cost two thirds of the product it follows
__device__ __forceinline__ float div_term_f32(int div, float v, float s) {
    switch (div) {
    case NMFX_DIV_KL: { const float q = v * __builtin_amdgcn_rcpf(s); return fmaf(v, 0.6931471805599453f * __builtin_amdgcn_logf(q), s - v); }
    case NMFX_DIV_IS: { const float q = v * __builtin_amdgcn_rcpf(s); return fmaf(-0.6931471805599453f, __builtin_amdgcn_logf(q), q) - 1.0f; }
    default: { const float d = v - s; return d * d; }
    }
}

inline bool is_kc(int mode) { return mode == VIEW_KC || mode == VIEW_HSTACK_KC || mode == VIEW_WSTACK_KC || mode == VIEW_XSHIFT_KC; }
// launch gemm_pipe_kernel<BM,BN,...> for (bm, bn) in {(128,128), (64,128), (128,64)}
nmfx_status dispatch_pipe_whole(hipStream_t st, const GemmParams &p, int bm, int bn);            // gemm_pipe.hip
nmfx_status dispatch_pipe_edge(hipStream_t st, const GemmParams &p, int bm, int bn, bool vec);   // gemm_pipe_edge.hip

}  // namespace nmfx

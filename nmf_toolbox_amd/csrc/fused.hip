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
#include "fused_kernel.h"

namespace nmfx {

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

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

bool fused_supported(int K) { return K >= 32 && K <= 256 && K % 32 == 0; }
// (Kh, T) pairs instantiated in fused_cnmf_*.hip: Kh*T <= 512 (the register-stationary operand / the accumulators take Kh*T/2 VGPRs)
bool fused_supported_T(int Kh, int T) {
    static const int ok[][2] = {{64, 8}, {64, 4}, {32, 8}, {32, 16}, {64, 2}, {32, 4}, {128, 2}, {128, 4},
                                {32, 3}, {32, 5}, {32, 6}, {64, 3}, {32, 10}, {32, 12}, {64, 5}, {64, 6},    // round 3: context lengths 3, 5, 6, 10, 12 ...
                                {32, 7}, {32, 9}, {32, 11}, {64, 7}, {32, 13}, {32, 14}, {32, 15}, {128, 3}, {256, 2}};   // ... and the rest up to 16 (K <= 32), 8 (K <= 64)
    for (const auto &c : ok) if (c[0] == Kh && c[1] == T) return true;
    return false;
}

// ... and the pairs whose S pass also exists in the IS / alpha-beta form that stores both element maps (functors 11 / 13, cost-only form): every pair since
// round 6 (cnmf.m:179-194,227-231 on the fused passes wherever the euclidean / KL passes run)
bool fused_supported_T_dual(int Kh, int T) { return fused_supported_T(Kh, T); }

// one translation unit per K group and per extent kind (fused_k*.hip, fused_rag_k*.hip): the instantiations compile in parallel
#define NMFX_DECL(name) nmfx_status name(hipStream_t st, const FusedParams &p, int nsplit, bool d_rc, int func, bool do_g2, int epi)
NMFX_DECL(launch_fused_k32_96); NMFX_DECL(launch_fused_k128_192); NMFX_DECL(launch_fused_k224_256);
NMFX_DECL(launch_fused_rag_k32_96); NMFX_DECL(launch_fused_rag_k128_192); NMFX_DECL(launch_fused_rag_k224_256);
#undef NMFX_DECL
#define NMFX_DECL_T(name) nmfx_status name(hipStream_t st, const FusedParams &p, int nsplit, int func, bool do_g2)
NMFX_DECL_T(launch_fused_cnmf_a); NMFX_DECL_T(launch_fused_cnmf_b); NMFX_DECL_T(launch_fused_cnmf_c); NMFX_DECL_T(launch_fused_cnmf_d); NMFX_DECL_T(launch_fused_cnmf_e); NMFX_DECL_T(launch_fused_cnmf_f); NMFX_DECL_T(launch_fused_cnmf_g);
#undef NMFX_DECL_T

// nsplit: number of contraction ranges (grid.y); c_per_split must be a multiple of 64.  R (stationary rows) and Cn (streamed extent)
// that are not multiples of 128 / 64 select the RAG instantiations (masked edges).
nmfx_status launch_fused(hipStream_t st, const FusedParams &p, int nsplit, bool d_rc, int func, bool do_g2, int epi) {
    constexpr int FT_ROWS = 128, FT_C = 64;   // fused_kernel.h: stationary rows per workgroup, streamed rows per tile
    if (p.R <= 0 || p.Cn <= 0 || p.c_per_split % FT_C || p.c_per_split <= 0 || nsplit < 1 || (long)(nsplit - 1) * p.c_per_split >= p.Cn) {
        set_error("launch_fused: bad split (R=%ld Cn=%ld c_per_split=%ld nsplit=%d)", p.R, p.Cn, p.c_per_split, nsplit);
        return NMFX_ERR_INVALID;
    }
    if (p.T > 1) {   // cnmf: W-step form only
        if (!d_rc || epi != 0 || p.K % p.T != 0 || !fused_supported_T(p.K / p.T, p.T)) {
            set_error("launch_fused: cnmf pass with (K = %d, T = %d) is not instantiated", p.K, p.T);
            return NMFX_ERR_UNSUPPORTED;
        }
        const int kh = p.K / p.T;
        if ((kh == 64 && (p.T == 8 || p.T == 4))) return launch_fused_cnmf_a(st, p, nsplit, func, do_g2);
        if (kh == 32 && (p.T == 8 || p.T == 16)) return launch_fused_cnmf_b(st, p, nsplit, func, do_g2);
        if ((kh == 32 && (p.T == 3 || p.T == 5 || p.T == 6)) || (kh == 64 && p.T == 3)) return launch_fused_cnmf_d(st, p, nsplit, func, do_g2);
        if ((kh == 32 && (p.T == 10 || p.T == 12)) || (kh == 64 && (p.T == 5 || p.T == 6))) return launch_fused_cnmf_e(st, p, nsplit, func, do_g2);
        if ((kh == 32 && (p.T == 7 || p.T == 9 || p.T == 11)) || (kh == 64 && p.T == 7)) return launch_fused_cnmf_f(st, p, nsplit, func, do_g2);
        if ((kh == 32 && (p.T == 13 || p.T == 14 || p.T == 15)) || (kh == 128 && p.T == 3) || (kh == 256 && p.T == 2)) return launch_fused_cnmf_g(st, p, nsplit, func, do_g2);
        return launch_fused_cnmf_c(st, p, nsplit, func, do_g2);
    }
    if (!fused_supported(p.K)) { set_error("launch_fused: K=%d not supported (multiples of 32 up to 256)", p.K); return NMFX_ERR_UNSUPPORTED; }
    // the no-first-product pass of the W-step form writing a K-CONTIGUOUS output (os_k == 1: the transposed products (V'*W)' and (V'*W_flat)' of the euclidean H
    // steps): the accumulator transposed, so that lanes run along k (fused_kernel.h, SWAP).  Anything else keeps the row-contiguous form
    if (d_rc && do_g2 && func == 0 && epi == 0 && p.os_k == 1 && p.os_r != 1 && NMFX_G2_VEC) epi = 2;
    const bool rag = p.R % FT_ROWS != 0 || p.Cn % FT_C != 0;
    if (p.K <= 96) return rag ? launch_fused_rag_k32_96(st, p, nsplit, d_rc, func, do_g2, epi) : launch_fused_k32_96(st, p, nsplit, d_rc, func, do_g2, epi);
    if (p.K <= 192) return rag ? launch_fused_rag_k128_192(st, p, nsplit, d_rc, func, do_g2, epi) : launch_fused_k128_192(st, p, nsplit, d_rc, func, do_g2, epi);
    return rag ? launch_fused_rag_k224_256(st, p, nsplit, d_rc, func, do_g2, epi) : launch_fused_k224_256(st, p, nsplit, d_rc, func, do_g2, epi);
}

}  // namespace nmfx

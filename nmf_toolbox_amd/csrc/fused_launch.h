// Launch templates of the fused kernel, shared by the per-K-group translation units fused_k*.hip.
#pragma once
#include "fused_kernel.h"

namespace nmfx {

template <int K, bool D_RC, int FUNC, bool DO_G2, int EPI, bool RAG>
static nmfx_status launch_one(hipStream_t st, const FusedParams &p, int nsplit) {
    const size_t ldsb = sizeof(float) * 2 * FT_C * (K + 4);
    auto kern = fused_kernel<K, D_RC, FUNC, DO_G2, EPI, RAG>;
    static LdsAttrOnce lds_attr;
    TRY(lds_attr.set(reinterpret_cast<const void *>(kern), (int)ldsb));
    dim3 grid((unsigned)((p.R + FT_ROWS - 1) / FT_ROWS), (unsigned)nsplit, (unsigned)(p.nz > 1 ? p.nz : 1));
    hipLaunchKernelGGL(kern, grid, dim3(256), ldsb, st, p);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

// cnmf (TT > 1), W-step form: numerator pass (FUNC 0: N_t = V * rshift_t(H)', all t at once) and cost-only pass (FUNC 1).  The masked-edge
// (RAG) instantiation serves every shape: its two extra VALU ops per element are nothing next to K = Kh*TT MFMAs per streamed column.
template <int K, int FUNC, bool DO_G2, int TT>
static nmfx_status launch_one_T(hipStream_t st, const FusedParams &p, int nsplit) {
    const size_t ldsb = sizeof(float) * 2 * (FT_C + TT - 1) * (K / TT + 4);
    auto kern = fused_kernel<K, true, FUNC, DO_G2, 0, true, TT>;
    static LdsAttrOnce lds_attr;
    TRY(lds_attr.set(reinterpret_cast<const void *>(kern), (int)ldsb));
    dim3 grid((unsigned)((p.R + FT_ROWS - 1) / FT_ROWS), (unsigned)nsplit);
    hipLaunchKernelGGL(kern, grid, dim3(256), ldsb, st, p);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}
template <int KH, int TT, bool DUALT = true>   // (round 5 instantiated the IS / alpha-beta S pass for eight pairs only; round 6: for every pair)
static nmfx_status launch_T(hipStream_t st, const FusedParams &p, int nsplit, int func, bool do_g2) {
    if (func == 0 && do_g2) return launch_one_T<KH * TT, 0, true, TT>(st, p, nsplit);
    if (func == 1 && !do_g2) return launch_one_T<KH * TT, 1, false, TT>(st, p, nsplit);
    if (func == 3 && !do_g2) return launch_one_T<KH * TT, 3, false, TT>(st, p, nsplit);   // KL: S pass (cost, optionally R = V./S to HBM)
    if constexpr (KH <= 128) {   // cnmfsc (K <= 128): the objective pass that leaves the residual S - V in HBM
        if (func == 21 && !do_g2) return launch_one_T<KH * TT, 21, false, TT>(st, p, nsplit);
    }
    if constexpr (DUALT) {   // IS / alpha-beta: S pass that stores both element maps' values (functors 11 / 13, cost-only form)
        if (func == 11 && !do_g2) return launch_one_T<KH * TT, 11, false, TT>(st, p, nsplit);
        if (func == 13 && !do_g2) return launch_one_T<KH * TT, 13, false, TT>(st, p, nsplit);
    }
    set_error("launch_fused_T: unsupported pass (func %d, do_g2 %d)", func, (int)do_g2);
    return NMFX_ERR_UNSUPPORTED;
}

template <int K, bool D_RC, bool DO_G2, int EPI, bool RAG>
static nmfx_status launch_f(hipStream_t st, const FusedParams &p, int nsplit, int func) {
    switch (func) {
    case 0: if (DO_G2) return launch_one<K, D_RC, 0, DO_G2, EPI, RAG>(st, p, nsplit); break;
    case 1: return launch_one<K, D_RC, 1, DO_G2, EPI, RAG>(st, p, nsplit);
    case 2: if (DO_G2) return launch_one<K, D_RC, 2, DO_G2, EPI, RAG>(st, p, nsplit); break;
    case 3: return launch_one<K, D_RC, 3, DO_G2, EPI, RAG>(st, p, nsplit);
    case 4: if constexpr (K <= 192) return launch_one<K, D_RC, 4, DO_G2, EPI, RAG>(st, p, nsplit); break;   // dual-map divergences: two accumulator sets
    case 5: if constexpr (K <= 192) return launch_one<K, D_RC, 5, DO_G2, EPI, RAG>(st, p, nsplit); break;
    case 6: if constexpr (DO_G2 && EPI == 0) return launch_one<K, D_RC, 6, DO_G2, EPI, RAG>(st, p, nsplit); break;   // residual-form gradients (nmfsc)
    case 7: if constexpr (!DO_G2 && D_RC && K >= 128) return launch_one<K, D_RC, 7, DO_G2, EPI, RAG>(st, p, nsplit); break;   // S over column blocks of a factor wider than 256
    case 8: if constexpr (!DO_G2 && D_RC && K >= 128) return launch_one<K, D_RC, 8, DO_G2, EPI, RAG>(st, p, nsplit); break;
    case 10: if constexpr (!DO_G2 && D_RC && K >= 128) return launch_one<K, D_RC, 10, DO_G2, EPI, RAG>(st, p, nsplit); break;   // ... of a euclidean chain: residual cost
    case 19: if constexpr (!DO_G2 && D_RC && K >= 128) return launch_one<K, D_RC, 19, DO_G2, EPI, RAG>(st, p, nsplit); break;   // ... of an IS chain: both element maps stored
    case 20: if constexpr (!DO_G2 && D_RC && K >= 128) return launch_one<K, D_RC, 20, DO_G2, EPI, RAG>(st, p, nsplit); break;   // ... of an alpha-beta chain
    // the dual-map divergences above K = 192 as two single-map passes (K = 224, 256 only): 11 / 13 also in the cost-only form, 12 / 14 with the second product only
    case 11: if constexpr (K >= 224 && EPI == 0) return launch_one<K, D_RC, 11, DO_G2, EPI, RAG>(st, p, nsplit); break;
    case 13: if constexpr (K >= 224 && EPI == 0) return launch_one<K, D_RC, 13, DO_G2, EPI, RAG>(st, p, nsplit); break;
    case 12: if constexpr (K >= 224 && EPI == 0 && DO_G2) return launch_one<K, D_RC, 12, DO_G2, EPI, RAG>(st, p, nsplit); break;
    case 14: if constexpr (K >= 224 && EPI == 0 && DO_G2) return launch_one<K, D_RC, 14, DO_G2, EPI, RAG>(st, p, nsplit); break;
    case 17: if constexpr (EPI == 0 && DO_G2) return launch_one<K, D_RC, 17, DO_G2, EPI, RAG>(st, p, nsplit); break;   // alpha-beta dual form (alpha == 0): numerators; any K
    case 15: if constexpr (K >= 224 && EPI == 0 && DO_G2 && D_RC) return launch_one<K, D_RC, 15, DO_G2, EPI, RAG>(st, p, nsplit); break;   // 11 / 13 + the second map's values to p.Rout
    case 16: if constexpr (K >= 224 && EPI == 0 && DO_G2 && D_RC) return launch_one<K, D_RC, 16, DO_G2, EPI, RAG>(st, p, nsplit); break;
    }
    set_error("launch_fused: unsupported functor %d", func);
    return NMFX_ERR_UNSUPPORTED;
}

template <int K, bool RAG>
static nmfx_status launch_k(hipStream_t st, const FusedParams &p, int nsplit, bool d_rc, int func, bool do_g2, int epi) {
    if (!do_g2) return d_rc ? launch_f<K, true, false, 0, RAG>(st, p, nsplit, func) : NMFX_ERR_UNSUPPORTED;   // cost-only pass
    if (d_rc && epi == 2) {   // W-step form, K-contiguous output: transposed accumulator (functor 0 only)
        if (func == 0) return launch_one<K, true, 0, true, 2, RAG>(st, p, nsplit);
        return NMFX_ERR_UNSUPPORTED;
    }
    if (d_rc) return launch_f<K, true, true, 0, RAG>(st, p, nsplit, func);                                    // W step: slabs out
    if (epi == 1) return launch_f<K, false, true, 1, RAG>(st, p, nsplit, func);                               // H step, fused update
    return launch_f<K, false, true, 0, RAG>(st, p, nsplit, func);                                             // H step, slabs out
}


}  // namespace nmfx

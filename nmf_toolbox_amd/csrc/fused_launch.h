// Launch templates of the fused kernel, shared by the per-K-group translation units fused_k*.hip.
#pragma once
#include "fused_kernel.h"

namespace nmfx {

template <int K, bool D_RC, int FUNC, bool DO_G2, int EPI, bool RAG>
static nmfx_status launch_one(hipStream_t st, const FusedParams &p, int nsplit) {
    const size_t ldsb = sizeof(float) * 2 * FT_C * (K + 4);
    auto kern = fused_kernel<K, D_RC, FUNC, DO_G2, EPI, 0, RAG>;
    static bool attr_done = false;
    if (!attr_done) {
        NMFX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
        attr_done = true;
    }
    dim3 grid((unsigned)((p.R + FT_ROWS - 1) / FT_ROWS), (unsigned)nsplit);
    hipLaunchKernelGGL(kern, grid, dim3(256), ldsb, st, p);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

template <int K, bool D_RC, bool DO_G2, int EPI, bool RAG>
static nmfx_status launch_f(hipStream_t st, const FusedParams &p, int nsplit, int func) {
    switch (func) {
    case 0: if (DO_G2) return launch_one<K, D_RC, 0, DO_G2, EPI, RAG>(st, p, nsplit); break;
    case 1: return launch_one<K, D_RC, 1, DO_G2, EPI, RAG>(st, p, nsplit);
    case 2: if (DO_G2) return launch_one<K, D_RC, 2, DO_G2, EPI, RAG>(st, p, nsplit); break;
    case 3: return launch_one<K, D_RC, 3, DO_G2, EPI, RAG>(st, p, nsplit);
    }
    set_error("launch_fused: unsupported functor %d", func);
    return NMFX_ERR_UNSUPPORTED;
}

template <int K, bool RAG>
static nmfx_status launch_k(hipStream_t st, const FusedParams &p, int nsplit, bool d_rc, int func, bool do_g2, int epi) {
    if (!do_g2) return d_rc ? launch_f<K, true, false, 0, RAG>(st, p, nsplit, func) : NMFX_ERR_UNSUPPORTED;   // cost-only pass
    if (d_rc) return launch_f<K, true, true, 0, RAG>(st, p, nsplit, func);                                    // W step: slabs out
    if (epi == 1) return launch_f<K, false, true, 1, RAG>(st, p, nsplit, func);                               // H step, fused update
    return launch_f<K, false, true, 0, RAG>(st, p, nsplit, func);                                             // H step, slabs out
}


}  // namespace nmfx

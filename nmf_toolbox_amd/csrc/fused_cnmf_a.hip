// Fused-kernel instantiations for cnmf (W-step form with T time-shifted bases), (Kh, T) in [(64, 8), (64, 4)] (see fused_kernel.h).
#include "fused_launch.h"

namespace nmfx {

nmfx_status launch_fused_cnmf_a(hipStream_t st, const FusedParams &p, int nsplit, int func, bool do_g2) {
    if (p.K == 512 && p.T == 8) return launch_T<64, 8, true>(st, p, nsplit, func, do_g2);
    if (p.K == 256 && p.T == 4) return launch_T<64, 4, true>(st, p, nsplit, func, do_g2);
    set_error("launch_fused_T: (K = %d, T = %d) not in this group", p.K, p.T);
    return NMFX_ERR_UNSUPPORTED;
}

}  // namespace nmfx

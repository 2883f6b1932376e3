// Fused-kernel instantiations for cnmf (W-step form with T time-shifted bases), (Kh, T) in [(32, 8), (32, 16)] (see fused_kernel.h).
#include "fused_launch.h"

namespace nmfx {

nmfx_status launch_fused_cnmf_b(hipStream_t st, const FusedParams &p, int nsplit, int func, bool do_g2) {
    if (p.K == 256 && p.T == 8) return launch_T<32, 8, true>(st, p, nsplit, func, do_g2);
    if (p.K == 512 && p.T == 16) return launch_T<32, 16, true>(st, p, nsplit, func, do_g2);
    set_error("launch_fused_T: (K = %d, T = %d) not in this group", p.K, p.T);
    return NMFX_ERR_UNSUPPORTED;
}

}  // namespace nmfx

// Fused-kernel instantiations for cnmf (W-step form with T time-shifted bases), (Kh, T) in [(32, 10), (32, 12), (64, 5), (64, 6)] (see fused_kernel.h).
#include "fused_launch.h"

namespace nmfx {

nmfx_status launch_fused_cnmf_e(hipStream_t st, const FusedParams &p, int nsplit, int func, bool do_g2) {
    if (p.K == 320 && p.T == 10) return launch_T<32, 10>(st, p, nsplit, func, do_g2);
    if (p.K == 384 && p.T == 12) return launch_T<32, 12>(st, p, nsplit, func, do_g2);
    if (p.K == 320 && p.T == 5) return launch_T<64, 5>(st, p, nsplit, func, do_g2);
    if (p.K == 384 && p.T == 6) return launch_T<64, 6>(st, p, nsplit, func, do_g2);
    set_error("launch_fused_T: (K = %d, T = %d) not in this group", p.K, p.T);
    return NMFX_ERR_UNSUPPORTED;
}

}  // namespace nmfx

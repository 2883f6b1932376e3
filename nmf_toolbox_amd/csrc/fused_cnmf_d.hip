// Fused-kernel instantiations for cnmf (W-step form with T time-shifted bases), (Kh, T) in [(32, 3), (32, 5), (32, 6), (64, 3)] (see fused_kernel.h).
#include "fused_launch.h"

namespace nmfx {

nmfx_status launch_fused_cnmf_d(hipStream_t st, const FusedParams &p, int nsplit, int func, bool do_g2) {
    if (p.K == 96 && p.T == 3) return launch_T<32, 3>(st, p, nsplit, func, do_g2);
    if (p.K == 160 && p.T == 5) return launch_T<32, 5>(st, p, nsplit, func, do_g2);
    if (p.K == 192 && p.T == 6) return launch_T<32, 6>(st, p, nsplit, func, do_g2);
    if (p.K == 192 && p.T == 3) return launch_T<64, 3>(st, p, nsplit, func, do_g2);
    set_error("launch_fused_T: (K = %d, T = %d) not in this group", p.K, p.T);
    return NMFX_ERR_UNSUPPORTED;
}

}  // namespace nmfx

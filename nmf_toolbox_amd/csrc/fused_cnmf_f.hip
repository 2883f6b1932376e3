// Fused-kernel instantiations for cnmf (W-step form with T time-shifted bases), (Kh, T) in [(32, 7), (32, 9), (32, 11), (64, 7)] (see fused_kernel.h).
#include "fused_launch.h"

namespace nmfx {

nmfx_status launch_fused_cnmf_f(hipStream_t st, const FusedParams &p, int nsplit, int func, bool do_g2) {
    if (p.K == 224 && p.T == 7) return launch_T<32, 7>(st, p, nsplit, func, do_g2);
    if (p.K == 288 && p.T == 9) return launch_T<32, 9>(st, p, nsplit, func, do_g2);
    if (p.K == 352 && p.T == 11) return launch_T<32, 11>(st, p, nsplit, func, do_g2);
    if (p.K == 448 && p.T == 7) return launch_T<64, 7>(st, p, nsplit, func, do_g2);
    set_error("launch_fused_T: (K = %d, T = %d) not in this group", p.K, p.T);
    return NMFX_ERR_UNSUPPORTED;
}

}  // namespace nmfx

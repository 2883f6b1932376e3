// Fused-kernel instantiations for cnmf (W-step form with T time-shifted bases), (Kh, T) in [(32, 13), (32, 14), (32, 15), (128, 3), (256, 2)] (see fused_kernel.h).
#include "fused_launch.h"

namespace nmfx {

nmfx_status launch_fused_cnmf_g(hipStream_t st, const FusedParams &p, int nsplit, int func, bool do_g2) {
    if (p.K == 416 && p.T == 13) return launch_T<32, 13>(st, p, nsplit, func, do_g2);
    if (p.K == 448 && p.T == 14) return launch_T<32, 14>(st, p, nsplit, func, do_g2);
    if (p.K == 480 && p.T == 15) return launch_T<32, 15>(st, p, nsplit, func, do_g2);
    if (p.K == 384 && p.T == 3) return launch_T<128, 3>(st, p, nsplit, func, do_g2);
    if (p.K == 512 && p.T == 2) return launch_T<256, 2>(st, p, nsplit, func, do_g2);
    set_error("launch_fused_T: (K = %d, T = %d) not in this group", p.K, p.T);
    return NMFX_ERR_UNSUPPORTED;
}

}  // namespace nmfx

// Fused-kernel instantiations for K in {128, 160, 192} (see fused.hip / fused_kernel.h).
#include "fused_launch.h"

namespace nmfx {

nmfx_status launch_fused_k128_192(hipStream_t st, const FusedParams &p, int nsplit, bool d_rc, int func, bool do_g2, int epi) {
    switch (p.K) {
    case 128: return launch_k<128, false>(st, p, nsplit, d_rc, func, do_g2, epi);
    case 160: return launch_k<160, false>(st, p, nsplit, d_rc, func, do_g2, epi);
    case 192: return launch_k<192, false>(st, p, nsplit, d_rc, func, do_g2, epi);
    default: set_error("launch_fused: K=%d not in this group", p.K); return NMFX_ERR_UNSUPPORTED;
    }
}

}  // namespace nmfx

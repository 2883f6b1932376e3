// Fused-kernel instantiations for K in {32, 64, 96} (see fused.hip / fused_kernel.h).
#include "fused_launch.h"

namespace nmfx {

nmfx_status launch_fused_k32_96(hipStream_t st, const FusedParams &p, int nsplit, bool d_rc, int func, bool do_g2, int epi) {
    switch (p.K) {
    case 32: return launch_k<32, false>(st, p, nsplit, d_rc, func, do_g2, epi);
    case 64: return launch_k<64, false>(st, p, nsplit, d_rc, func, do_g2, epi);
    case 96: return launch_k<96, false>(st, p, nsplit, d_rc, func, do_g2, epi);
    default: set_error("launch_fused: K=%d not in this group", p.K); return NMFX_ERR_UNSUPPORTED;
    }
}

}  // namespace nmfx

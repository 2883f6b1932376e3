// Pipelined GEMM instantiations with guards: edge tiles / short last k-tile (float4 loads), and the dword-load variant for views
// with odd leading dimensions.
#include "gemm_pipe.h"

namespace nmfx {

nmfx_status dispatch_pipe_edge(hipStream_t st, const GemmParams &p, int bm, int bn, bool vec) {
    if (!vec) return dispatch_pipe_t<128, 128, false, true>(st, p);   // unaligned views: dword loads, one tile shape
    if (bm == 64) return dispatch_pipe_t<64, 128, true, true>(st, p);
    if (bn == 64) return dispatch_pipe_t<128, 64, true, true>(st, p);
    return dispatch_pipe_t<128, 128, true, true>(st, p);
}

}  // namespace nmfx

// Pipelined GEMM instantiations for problems made of whole tiles (M, N multiples of the tile, k ranges multiples of 32, float4 loads).
#include "gemm_pipe.h"

namespace nmfx {

nmfx_status dispatch_pipe_whole(hipStream_t st, const GemmParams &p, int bm, int bn) {
    if (bm == 64) return dispatch_pipe_t<64, 128, true, false>(st, p);
    if (bn == 64) return dispatch_pipe_t<128, 64, true, false>(st, p);
    return dispatch_pipe_t<128, 128, true, false>(st, p);
}

}  // namespace nmfx

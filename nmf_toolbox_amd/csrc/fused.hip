// Fused two-stage kernels (S = W*H in registers -> element map -> second MFMA contraction); see DESIGN.md.
#include "nmfx_internal.h"
namespace nmfx {
}

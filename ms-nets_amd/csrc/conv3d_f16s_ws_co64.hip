// Stride-1 layers with Co = 64 / 128 (two N-blocks per workgroup, swizzled 128-byte records).
#include "conv_f16s_ws.h"

namespace msnet {
// w16: widths that are 16 mod 32 (240, 120, ...): 16-wide M-block rows leave no half-empty edge tile and a smaller halo
int ws_launch_co64(bool w16, const char* name, ConvArgs a, hipStream_t s) {
    if (w16) return launch_f16s<2, 8, 16, 16, 2, 2, true, 2, false>(name, a, s);
    return launch_f16s<2, 4, 32, 32, 2, 2, true, 2, false>(name, a, s);
}
}  // namespace msnet

// 16-channel input (the left+right volume, cbmv_in_planes = 16): a single 16-channel chunk, streamed weight groups of 3 K-steps.
#include "conv_f16s_ws.h"

namespace msnet {
int ws_launch_c16(bool co64, const char* name, ConvArgs a, hipStream_t s) {
    if (co64) return launch_f16s<2, 4, 32, 32, 2, 2, false, 1, false>(name, a, s);
    return launch_f16s<2, 4, 32, 32, 2, 1, false, 1, false>(name, a, s);
}
}  // namespace msnet

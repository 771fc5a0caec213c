// Stride-2 layers (32 -> 64, 64 -> 64, 64 -> 128): conv3d_k3s1_f16s_ws with STRIDE = 2.
#include "conv_f16s_ws.h"

namespace msnet {
// 2x2x32 output tile <- 5x5x65 input voxels x 16 channels (104 KB of 64-byte swizzled records); 4 M-blocks, one per MFMA wave
int ws_launch_s2(const char* name, ConvArgs a, hipStream_t s) {
    return launch_f16s<2, 2, 32, 32, 1, 2, S2_SWZ, 1, false, 2, S2_LOADER_WAVES>(name, a, s);
}
}  // namespace msnet

// Stride-2 layers (32 -> 64, 64 -> 64, 64 -> 128): conv3d_k3s1_f16s_ws with STRIDE = 2.
#include "conv_f16s_ws.h"

namespace msnet {
// 2x2x32 output tile <- 5x5x65 input voxels x 16 channels (104 KB of 64-byte swizzled records); 4 M-blocks, one per MFMA wave.
// Output widths that are 16 mod 32 (240, 120: configs #2 / #4) with rows in fours take a 2x4x16 tile of two-row 16-wide M-blocks
// <- 5x9x33 input voxels instead: 8.6 % fewer staged voxels per output and no half-empty edge tile (7.5 tiles per 240-wide row).
// Same products in the same order: bit-identical.  Round 4, one box, interleaved: 32 -> 64 at 96x272x480 0.995 -> 0.941 ms,
// step 138.5 -> 139.3 maps/s (gpurun_out/r04_s2w16).
int ws_launch_s2(const char* name, ConvArgs a, hipStream_t s) {
    if (a.OW % 32 == 16 && a.OH % 4 == 0) return launch_f16s<2, 4, 16, 16, 1, 2, S2_SWZ, 1, false, 2, S2_LOADER_WAVES>(name, a, s);
    return launch_f16s<2, 2, 32, 32, 1, 2, S2_SWZ, 1, false, 2, S2_LOADER_WAVES>(name, a, s);
}
}  // namespace msnet

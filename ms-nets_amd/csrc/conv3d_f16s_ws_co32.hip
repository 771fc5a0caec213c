// Stride-1 layers with Co = 32: plain 2x4x32 tiles and the sliding window along d for single-chunk layers.
#include "conv_f16s_ws.h"

namespace msnet {
// (2x8x16 tiles and swizzled 128-byte records were measured for this layer too: both 4 % slower than padded 2x4x32)
int ws_launch_co32(const char* name, ConvArgs a, hipStream_t s) {
    return launch_f16s<2, 4, 32, 32, 2, 1, false, 2, false>(name, a, s);
}
// the same kernel on a MODULE INPUT: the loaders carry the input's fp16-range check (conv_f16s_ws.h INCHK)
int ws_launch_co32_inchk(const char* name, ConvArgs a, hipStream_t s) {
    return launch_f16s<2, 4, 32, 32, 2, 1, false, 2, false, 1, 4, true>(name, a, s);
}
int ws_launch_co32_slide(const char* name, ConvArgs a, hipStream_t s) {
    return launch_f16s_slide<4, 32, 2, 1>(name, a, s);
}
}  // namespace msnet

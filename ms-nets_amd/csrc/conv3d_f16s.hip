// Entry points of the split-fp16 conv path that are not tied to one kernel family: weight packing for the tiled / direct
// kernels and msnet_conv3d_k3_f16s, which picks the kernel for a shape.  The kernels live in conv3d_f16s_{ws_*,c8,deconv,
// direct,wd}.hip (one translation unit per family, so a change to one family rebuilds one unit); conv_f16s.h holds what they share.
#include "conv_f16s.h"

namespace msnet {
// Packed split weights, in 16-byte units (KS = 16-channel K-steps per chunk: 2 for Ci % 32 == 0, 1 for Ci = 8 which
// is zero-padded to 16 channels):
//   idx = (((((((cg*nchunks + chunk)*9 + grp)*3 + t)*KS + ks)*NBG + nbl)*2 + hl)*64 + lane
//   element j of lane (r = lane&31, h = lane>>5):
//       W[co = (cg*NBG + nbl)*32 + r][ci = chunk*16*KS + ks*16 + h*8 + j][tap = grp*3 + t]
//   hl = 0: fp16(w);  hl = 1: fp16((w - hi) * 2^11).   NBG = N-blocks per output-channel group (a workgroup handles one
//   group: 1 block for Co = 32, 2 for Co = 64 / 128).  One weight group = 6*KS*NBG KiB, contiguous.
template <bool TRANSPOSED>
__global__ void pack_weight_f16s_kernel(const float* __restrict__ w, _Float16* __restrict__ out, int Ci, int Co, int KS,
                                        int NBG) {
    const int cip = Ci < 16 * KS ? 16 * KS : Ci;        // padded input channels
    const size_t total = (size_t)27 * cip * Co * 2;
    const int nbt = NBG;
    const int nchunks = cip / (16 * KS);
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        size_t i = o;
        const int j = i & 7; i >>= 3;
        const int lane = i & 63; i >>= 6;
        const int hl = i & 1; i >>= 1;
        const int nb = i % nbt; i /= nbt;
        const int ks = i % KS; i /= KS;
        const int t = i % 3; i /= 3;
        const int grp = i % 9; i /= 9;
        const int chunk = i % nchunks;
        const int cg = (int)(i / nchunks);
        const int co = (cg * NBG + nb) * 32 + (lane & 31);
        const int ci = chunk * 16 * KS + ks * 16 + (lane >> 5) * 8 + j;
        const int tap = grp * 3 + t;
        float v = 0.f;
        if (ci < Ci) v = TRANSPOSED ? w[((size_t)ci * Co + co) * 27 + tap] : w[((size_t)co * Ci + ci) * 27 + tap];
        const _Float16 h = (_Float16)v;
        out[o] = hl ? (_Float16)((v - (float)h) * kLoScale) : h;
    }
}

}  // namespace msnet

using namespace msnet;

extern "C" int msnet_conv3d_k3_f16s_supported(int Ci, int Co, int stride);

// Split-fp16 packed size: 27 * max(Ci,16) * Co floats (2 halves per weight; Ci = 8 is zero-padded to 16 channels).
extern "C" int msnet_pack_conv_weight_f16s(const float* w, void* packed, int Ci, int Co, int stride,
                                           msnet_stream_t stream) {
    if (!w || !packed) return fail("msnet_pack_conv_weight_f16s: null pointer");
    if (!msnet_conv3d_k3_f16s_supported(Ci, Co, stride))
        return fail("msnet_pack_conv_weight_f16s: unsupported Ci=%d Co=%d stride=%d", Ci, Co, stride);
    const int KS = (Ci == 8 || Ci == 16 || stride == 2) ? 1 : 2;    // 16-channel K-steps per staged chunk
    if (Co <= 0 || Co % 32 != 0) return fail("msnet_pack_conv_weight_f16s: Co=%d must be a positive multiple of 32", Co);
    if (Ci == 8) {                                      // first-layer kernel: two taps per K-step
        return c8_pack_launch(w, (_Float16*)packed, Co, (hipStream_t)stream);
    }
    const size_t total = (size_t)27 * (Ci < 16 ? 16 : Ci) * Co * 2;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipStream_t s = (hipStream_t)stream;
    LaunchScope ls("pack_weight_f16s", s, 0, 6.0 * total);
    hipLaunchKernelGGL(pack_weight_f16s_kernel<false>, dim3(blocks), dim3(256), 0, s, w, (_Float16*)packed, Ci, Co, KS, Co == 32 ? 1 : 2);
    return check_launch("msnet_pack_conv_weight_f16s");
}

extern "C" int msnet_conv3d_k3_f16s_supported(int Ci, int Co, int stride) {
    if (stride == 2) return (Ci > 0 && Ci % 16 == 0 && Co > 0 && Co % 64 == 0) ? 1 : 0;
    // Ci = 16: the left+right matching-space volume (cbmv_in_planes = 16, gcnet_3dcnn.py:58-65) -- one 16-channel K-step per tap
    return (stride == 1 && (Ci == 8 || Ci == 16 || (Ci > 0 && Ci % 32 == 0)) && (Co == 32 || (Co > 0 && Co % 64 == 0)) &&
            !((Ci == 8 || Ci == 16) && Co > 64)) ? 1 : 0;
}

extern "C" int msnet_conv3d_k3_f16s(const float* x, const void* wpk_f16s, const float* scale, const float* shift,
                                    const float* residual, float* y, int N, int D, int H, int W, int Ci, int Co,
                                    int stride, int relu, msnet_stream_t stream) {
    if (!x || !wpk_f16s || !y) return fail("msnet_conv3d_k3_f16s: null pointer");
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0) return fail("msnet_conv3d_k3_f16s: empty input");
    if (!msnet_conv3d_k3_f16s_supported(Ci, Co, stride))
        return fail("msnet_conv3d_k3_f16s: unsupported shape Ci=%d Co=%d stride=%d", Ci, Co, stride);
    ConvArgs a{};
    a.x = x; a.wpk = reinterpret_cast<const f32x4*>(wpk_f16s); a.scale = scale; a.shift = shift; a.res = residual; a.y = y;
    a.N = N; a.D = D; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co; a.relu = relu; a.oflag = overflow_flag();
    a.OD = (D - 1) / stride + 1; a.OH = (H - 1) / stride + 1; a.OW = (W - 1) / stride + 1;
    hipStream_t s = (hipStream_t)stream;
    {   // small layers: one workgroup per 32x32 output block instead of a handful of persistent tile walkers
        // items of ONE sample: the choice (and with it the summation order) must not depend on the batch size
        const size_t items = stride == 2 ? (size_t)cdiv(a.OD, 2) * cdiv(a.OH, 2) * cdiv(a.OW, 32) * (Co / 64)
                                         : (size_t)cdiv(a.OD, 2) * cdiv(a.OH, 4) * cdiv(a.OW, 32) * (Co == 32 ? 1 : Co / 64);
        if (direct_eligible(a, items))
            return direct_launch(false, stride == 2 ? "conv3d_s2_f16s" : (Co == 32 ? "conv3d_s1_f16s_co32" : "conv3d_s1_f16s_co64"),
                                 a, stride, stride == 2 ? 1 : 2, Co == 32 ? 1 : 2, s);
    }
    if (stride == 2) return ws_launch_s2("conv3d_s2_f16s", a, s);
    if (Ci == 8) return c8_launch(Co == 64 ? 2 : 1, false, false, "conv3d_s1_c8_f16s", a, s);
    if (Ci == 16) return ws_launch_c16(Co == 64, "conv3d_s1_c16_f16s", a, s);
    if (Co % 64 == 0) return ws_launch_co64(W % 32 == 16 && H % 8 == 0, "conv3d_s1_f16s_co64", a, s);
    if (Ci == 32) {                                     // single 32-channel chunk: sliding window along d
        const int rc = ws_launch_co32_slide("conv3d_s1_f16s_co32", a, s);
        if (rc >= 0) return rc;
    }
    return ws_launch_co32("conv3d_s1_f16s_co32", a, s);
}

// Stride-1 conv (no residual) on a channels-last MODULE INPUT x: f32[N][D][H][W][Ci] with the fp16-range check of that input
// (bit 1 of the calling thread's overflow word) -- PSMNet_CostVolumeAggre.forward_ndhwc's dres0.0 on the reference's 64-plane
// volume (psmnet_3dcnn.py:96-99,126-131).  Shapes the tiled Co = 32 kernel takes carry the check in their loaders (round 6: no
// separate pass over the volume); every other shape runs msnet_check_input_range in front of msnet_conv3d_k3_f16s.  Same bits as
// msnet_conv3d_k3_f16s either way.
extern "C" int msnet_check_input_range(const float* x, size_t count, msnet_stream_t stream);
extern "C" int msnet_conv3d_k3_in_f16s(const float* x_ndhwc, const void* wpk_f16s, const float* scale, const float* shift, float* y,
                                       int N, int D, int H, int W, int Ci, int Co, int relu, msnet_stream_t stream) {
    if (!x_ndhwc || !wpk_f16s || !y) return fail("msnet_conv3d_k3_in_f16s: null pointer");
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0) return fail("msnet_conv3d_k3_in_f16s: empty input");
    if (!msnet_conv3d_k3_f16s_supported(Ci, Co, 1)) return fail("msnet_conv3d_k3_in_f16s: unsupported shape Ci=%d Co=%d", Ci, Co);
    ConvArgs a{};
    a.x = x_ndhwc; a.wpk = reinterpret_cast<const f32x4*>(wpk_f16s); a.scale = scale; a.shift = shift; a.res = nullptr; a.y = y;
    a.N = N; a.D = D; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co; a.relu = relu; a.oflag = overflow_flag();
    a.OD = D; a.OH = H; a.OW = W;
    const size_t items = (size_t)cdiv(D, 2) * cdiv(H, 4) * cdiv(W, 32);       // of ONE sample, as msnet_conv3d_k3_f16s counts them
    if (Co == 32 && Ci % 32 == 0 && Ci > 32 && !direct_eligible(a, items))
        return ws_launch_co32_inchk("conv3d_s1_f16s_co32", a, (hipStream_t)stream);
    const int rc = msnet_check_input_range(x_ndhwc, (size_t)N * D * H * W * Ci, stream);
    if (rc) return rc;
    return msnet_conv3d_k3_f16s(x_ndhwc, wpk_f16s, scale, shift, nullptr, y, N, D, H, W, Ci, Co, 1, relu, stream);
}

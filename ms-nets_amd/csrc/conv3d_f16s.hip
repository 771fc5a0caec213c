// Split-fp16 3x3x3 convolution on the fp16 MFMA (v_mfma_f32_32x32x16_f16, 16x the fp32-MFMA rate).
//
// Every fp32 operand is written as  x = hi + lo * 2^-11  with  hi = fp16(x),  lo = fp16((x - hi) * 2^11)
// (22 significand bits; storing lo pre-scaled keeps it a NORMAL fp16 whenever hi is), and a product is
//     a*w  ~=  ah*wh  +  2^-11 * (al*wh + ah*wl)                (the dropped al*wl term is 2^-22 relative)
// i.e. three fp16 MFMAs into two fp32 accumulators (acc0: ah*wh, acc1: al*wh + ah*wl), combined once in the
// epilogue.  fp16 x fp16 products are exact in fp32, accumulation is fp32, so the result differs from the exact
// fp32 conv by ~3*2^-23 per product -- measured end to end on the parity fixtures this is below the fp32
// reference's own rounding noise (DESIGN.md "Numerics"); plain fp16 / bf16 / tf32 inputs are NOT (1e-2..1e-1).
// Requirement: |activation| < 65504 (fp16 range of `hi`); BN+ReLU activations of these nets are O(1..100).
//
// HBM layout is unchanged (fp32, channels-last): the LOADER waves split each staged fp32 voxel into the LDS image
//   [voxel][ hi c0..c31 (64 B) | lo c0..c31 (64 B) | 16 B pad ]
// while the MFMA waves work, so no other kernel sees the fp16 form.  Weights are split once at pack time.
//
// Work distribution is the wave-specialised persistent scheme of conv3d.hip (4 MFMA waves + 4 loader waves per
// workgroup, one workgroup per CU, work items = (tile, 32-channel chunk)).  At 5.3x the MFMA rate the weight
// stream can no longer come per-wave from L2 (it would need ~40 B/clk/CU), so the loaders also stream the B
// operand through LDS, one (kd,kh) row of three taps ("group") at a time into a double buffer:
//     loader :  |b1| write A_it, B_(it,0) |b2|  write B_1   |g0|  write B_2   |g1| ...   |g7|
//     compute:  |b1| epilogue(it-1)       |b2|  MFMA grp 0  |g0|  MFMA grp 1  |g1| ...   |g7| MFMA grp 8
// Group g+1's weights are written while group g is being multiplied; the barrier that ends group g publishes them.
#include "conv_common.h"

namespace msnet {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma16(half8 a, half8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

constexpr float kLoScale = 2048.f;          // 2^11
constexpr float kLoInv = 1.f / 2048.f;

__device__ __forceinline__ void split4(const f32x4 v, half4& hi, half4& lo) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const _Float16 h = (_Float16)v[k];
        hi[k] = h;
        lo[k] = (_Float16)((v[k] - (float)h) * kLoScale);
    }
}

// Packed split weights, in 16-byte units:
//   idx = ((((((chunk*9 + grp)*3 + t)*2 + ks)*NBT + nb)*2 + hl)*64 + lane
//   element j of lane (r = lane&31, h = lane>>5):  W[co = nb*32 + r][ci = chunk*32 + ks*16 + h*8 + j][tap = grp*3 + t]
//   hl = 0: fp16(w);  hl = 1: fp16((w - hi) * 2^11).           One group = 12*NBT KiB, contiguous.
template <bool TRANSPOSED>
__global__ void pack_weight_f16s_kernel(const float* __restrict__ w, _Float16* __restrict__ out, int Ci, int Co) {
    const size_t total = (size_t)27 * Ci * Co * 2;
    const int nbt = Co >> 5;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        size_t i = o;
        const int j = i & 7; i >>= 3;
        const int lane = i & 63; i >>= 6;
        const int hl = i & 1; i >>= 1;
        const int nb = i % nbt; i /= nbt;
        const int ks = i & 1; i >>= 1;
        const int t = i % 3; i /= 3;
        const int grp = i % 9;
        const int chunk = (int)(i / 9);
        const int co = nb * 32 + (lane & 31);
        const int ci = chunk * 32 + ks * 16 + (lane >> 5) * 8 + j;
        const int tap = grp * 3 + t;
        const float v = TRANSPOSED ? w[((size_t)ci * Co + co) * 27 + tap] : w[((size_t)co * Ci + ci) * 27 + tap];
        const _Float16 h = (_Float16)v;
        out[o] = hl ? (_Float16)((v - (float)h) * kLoScale) : h;
    }
}

template <int TD, int TH, int TW, int BW, int MB, int NB>
__global__ __launch_bounds__(512, 2) void conv3d_k3s1_f16s_ws(ConvArgs a) {
    constexpr int CC = 32;
    constexpr int BH = 32 / BW;
    constexpr int ID = TD + 2, IH = TH + 2, IW = TW + 2;
    constexpr int RB = 144;                             // bytes per voxel record in LDS (64 hi + 64 lo + 16 pad)
    constexpr int MW = TW / BW, MH = TH / BH;
    constexpr int V = CC / 4;
    constexpr int NPOS = ID * IH * IW;
    constexpr int NSLOT = NPOS * V;
    constexpr int NL = (NSLOT + 255) / 256;             // fp32 float4 per loader thread per tile
    constexpr int GB = 3 * 2 * NB * 2 * 1024;           // bytes of one weight group
    constexpr int NLB = GB / 16 / 256;                  // 16-byte pieces per loader thread per group
    static_assert(TD * MH * MW == 4 * MB, "M-block count mismatch");
    static_assert(GB % (16 * 256) == 0, "group must split evenly over the loader threads");
    __shared__ __attribute__((aligned(16))) unsigned char lds[NPOS * RB + 2 * GB];
    unsigned char* const lds_b = lds + NPOS * RB;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const unsigned G = gridDim.x;
    const unsigned lb = xcd_remap(blockIdx.x, G);
    const int nchunks = a.Ci / CC;
    const unsigned T = (unsigned)a.N * a.ntd * a.nth * a.ntw;
    const int my_tiles = (T > lb) ? (int)((T - lb + G - 1) / G) : 0;
    const int nitems = my_tiles * nchunks;
    if (nitems == 0) return;
    const uint4* wg = reinterpret_cast<const uint4*>(a.wpk);     // split-fp16 packed weights

    auto decode = [&](int it, int& n, int& od0, int& oh0, int& ow0, int& chunk) {
        unsigned t = lb + (unsigned)(it / nchunks) * G;
        chunk = it % nchunks;
        ow0 = (t % a.ntw) * TW; t /= a.ntw;
        oh0 = (t % a.nth) * TH; t /= a.nth;
        od0 = (t % a.ntd) * TD;
        n = t / a.ntd;
    };

    if (wave >= 4) {
        // ------------------------------ loader waves ------------------------------
        const int lt = tid - 256;
        f32x4 av[NL];
        uint4 bw[NLB];
        auto issue_a = [&](int it) {
            int n, od0, oh0, ow0, chunk;
            decode(it, n, od0, oh0, ow0, chunk);
            const float* xc = a.x + chunk * CC;
#pragma unroll
            for (int u = 0; u < NL; ++u) {
                const int slot = u * 256 + lt;
                const int pos = slot / V, c4 = slot % V;
                const int iw = pos % IW, ih = (pos / IW) % IH, id = pos / (IW * IH);
                const int gd = od0 - 1 + id, gh = oh0 - 1 + ih, gw = ow0 - 1 + iw;
                av[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (slot < NSLOT && (unsigned)gd < (unsigned)a.D && (unsigned)gh < (unsigned)a.H &&
                    (unsigned)gw < (unsigned)a.W) {
                    const size_t vox = (((size_t)n * a.D + gd) * a.H + gh) * a.W + gw;
                    av[u] = *reinterpret_cast<const f32x4*>(xc + vox * a.Ci + c4 * 4);
                }
            }
        };
        auto write_a = [&]() {
#pragma unroll
            for (int u = 0; u < NL; ++u) {
                const int slot = u * 256 + lt;
                if (slot < NSLOT) {
                    half4 hi, lo;
                    split4(av[u], hi, lo);
                    unsigned char* rec = lds + (slot / V) * RB + (slot % V) * 8;
                    *reinterpret_cast<half4*>(rec) = hi;
                    *reinterpret_cast<half4*>(rec + 64) = lo;
                }
            }
        };
        auto issue_b = [&](int chunk, int grp) {
            const uint4* src = wg + (size_t)(chunk * 9 + grp) * (GB / 16) + lt;
#pragma unroll
            for (int u = 0; u < NLB; ++u) bw[u] = src[u * 256];
        };
        auto write_b = [&](int buf) {
            uint4* dst = reinterpret_cast<uint4*>(lds_b + buf * GB) + lt;
#pragma unroll
            for (int u = 0; u < NLB; ++u) dst[u * 256] = bw[u];
        };

        issue_a(0);
        issue_b(0, 0);
        for (int it = 0; it < nitems; ++it) {
            const int chunk = it % nchunks;
            const int gg0 = it * 9;
            MSNET_LDS_BARRIER();                        // b1: MFMA waves are done with the previous tile
            write_a();
            write_b(gg0 & 1);
            MSNET_LDS_BARRIER();                        // b2: tile and group 0 are in LDS
            issue_b(chunk, 1);
            if (it + 1 < nitems) issue_a(it + 1);
#pragma unroll 1
            for (int g = 0; g < 8; ++g) {
                write_b((gg0 + g + 1) & 1);             // group g+1, while group g is multiplied
                if (g < 7) issue_b(chunk, g + 2);
                else if (it + 1 < nitems) issue_b((it + 1) % nchunks, 0);
                MSNET_LDS_BARRIER();                    // g_g
            }
        }
        return;
    }

    // ------------------------------ MFMA waves ------------------------------
    const int wm = wave;                                // WM = 4, WN = 1
    const int r = lane & 31, hh = lane >> 5;
    int abase[MB];                                      // byte offsets
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int mb = wm * MB + i;
        const int bw_ = mb % MW, bh = (mb / MW) % MH, bd = mb / (MW * MH);
        const int lh = bh * BH + r / BW, lw = bw_ * BW + r % BW;
        abase[i] = ((bd * IH + lh) * IW + lw) * RB + 16 * hh;
    }
    const int stride_w = a.Co, stride_h = a.OW * a.Co;

    f32x16 acc0[MB][NB], acc1[MB][NB];
    int pn = 0, pod0 = 0, poh0 = 0, pow0 = 0;           // coordinates of the item whose epilogue is pending
    bool pending = false;

    auto epilogue = [&](int n, int od0, int oh0, int ow0) {
        const bool full_hw = (oh0 + TH <= a.OH) && (ow0 + TW <= a.OW);
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const int mb = wm * MB + i;
            const int bw_ = mb % MW, bh = (mb / MW) % MH, bd = mb / (MW * MH);
            const int od = od0 + bd;
            if (od >= a.OD) continue;
            const int ohb = oh0 + bh * BH, owb = ow0 + bw_ * BW + 4 * hh;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int co = j * 32 + r;
                const float sc = a.scale ? a.scale[co] : 1.f;
                const float sh = a.shift ? a.shift[co] : 0.f;
                f32x16 v;
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = acc0[i][j][e] + acc1[i][j][e] * kLoInv;
                const size_t base = ((((size_t)n * a.OD + od) * a.OH + ohb) * a.OW + owb) * a.Co + co;
                epilogue_block<BW>(v, sc, sh, a.res, a.y, base, stride_h, stride_w, a.relu, full_hw,
                                   [&](int lh, int lw) { return ohb + lh < a.OH && owb + lw < a.OW; });
            }
        }
    };

    for (int it = 0; it < nitems; ++it) {
        int n, od0, oh0, ow0, chunk;
        decode(it, n, od0, oh0, ow0, chunk);
        MSNET_LDS_BARRIER();                            // b1
        if (pending) { epilogue(pn, pod0, poh0, pow0); pending = false; }
        MSNET_LDS_BARRIER();                            // b2
        if (chunk == 0) {
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) { acc0[i][j][e] = 0.f; acc1[i][j][e] = 0.f; }
        }
        const int gg0 = it * 9;
#pragma unroll 1
        for (int g = 0; g < 9; ++g) {
            const unsigned char* bb = lds_b + ((gg0 + g) & 1) * GB + lane * 16;
            const int goff = ((g / 3) * IH + (g % 3)) * IW * RB;         // (kd, kh) row of this group
            // 6 steps (t = kw tap, ks = 16-channel K-step); fragments of step s+1 are read while step s multiplies
            half8 ah[2][MB], al[2][MB], bh_[2][NB], bl[2][NB];
            auto frag = [&](int s, int slot) {
                const int t = s >> 1, ks = s & 1;
#pragma unroll
                for (int i = 0; i < MB; ++i) {
                    const unsigned char* p = lds + abase[i] + goff + t * RB + ks * 32;
                    ah[slot][i] = *reinterpret_cast<const half8*>(p);
                    al[slot][i] = *reinterpret_cast<const half8*>(p + 64);
                }
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const unsigned char* p = bb + (((t * 2 + ks) * NB + j) * 2) * 1024;
                    bh_[slot][j] = *reinterpret_cast<const half8*>(p);
                    bl[slot][j] = *reinterpret_cast<const half8*>(p + 1024);
                }
            };
            frag(0, 0);
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                if (s + 1 < 6) frag(s + 1, (s + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        acc0[i][j] = mfma16(ah[s & 1][i], bh_[s & 1][j], acc0[i][j]);
                        acc1[i][j] = mfma16(al[s & 1][i], bh_[s & 1][j], acc1[i][j]);
                        acc1[i][j] = mfma16(ah[s & 1][i], bl[s & 1][j], acc1[i][j]);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (g < 8) MSNET_LDS_BARRIER();             // g_g
        }
        if (chunk == nchunks - 1) { pending = true; pn = n; pod0 = od0; poh0 = oh0; pow0 = ow0; }
    }
    if (pending) epilogue(pn, pod0, poh0, pow0);
}

template <int TD, int TH, int TW, int BW, int MB, int NB>
static int launch_f16s(const char* name, ConvArgs a, hipStream_t s) {
    a.ntd = cdiv(a.OD, TD); a.nth = cdiv(a.OH, TH); a.ntw = cdiv(a.OW, TW);
    a.ngroups = 1; a.nbtot = a.Co / 32;
    const size_t ntiles = (size_t)a.N * a.ntd * a.nth * a.ntw;
    if (ntiles == 0 || ntiles > 0x7fffffffu) return fail("%s: bad tile count %zu", name, ntiles);
    const size_t nblk = ntiles < (size_t)num_cus() ? ntiles : (size_t)num_cus();
    const double vox = (double)a.N * a.OD * a.OH * a.OW;
    LaunchScope ls(name, s, 2.0 * 27.0 * a.Ci * a.Co * vox,
                   4.0 * ((double)a.N * a.D * a.H * a.W * a.Ci + vox * a.Co * (a.res ? 2 : 1)));
    hipLaunchKernelGGL((conv3d_k3s1_f16s_ws<TD, TH, TW, BW, MB, NB>), dim3((unsigned)nblk), dim3(512), 0, s, a);
    return check_launch(name);
}

}  // namespace msnet

using namespace msnet;

// Split-fp16 packed size in floats (same byte count as the fp32 packing: 2 halves per weight).
extern "C" int msnet_pack_conv_weight_f16s(const float* w, void* packed, int Ci, int Co, int transposed,
                                           msnet_stream_t stream) {
    if (!w || !packed) return fail("msnet_pack_conv_weight_f16s: null pointer");
    if (Ci <= 0 || Ci % 32 != 0) return fail("msnet_pack_conv_weight_f16s: Ci=%d must be a positive multiple of 32", Ci);
    if (Co <= 0 || Co % 32 != 0) return fail("msnet_pack_conv_weight_f16s: Co=%d must be a positive multiple of 32", Co);
    const size_t total = (size_t)27 * Ci * Co * 2;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipStream_t s = (hipStream_t)stream;
    LaunchScope ls("pack_weight_f16s", s, 0, 6.0 * total);
    if (transposed) hipLaunchKernelGGL(pack_weight_f16s_kernel<true>, dim3(blocks), dim3(256), 0, s, w, (_Float16*)packed, Ci, Co);
    else            hipLaunchKernelGGL(pack_weight_f16s_kernel<false>, dim3(blocks), dim3(256), 0, s, w, (_Float16*)packed, Ci, Co);
    return check_launch("msnet_pack_conv_weight_f16s");
}

extern "C" int msnet_conv3d_k3_f16s_supported(int Ci, int Co, int stride) {
    return (stride == 1 && Ci > 0 && Ci % 32 == 0 && (Co == 32 || Co == 64)) ? 1 : 0;
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
    a.N = N; a.D = D; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co; a.relu = relu;
    a.OD = D; a.OH = H; a.OW = W;
    hipStream_t s = (hipStream_t)stream;
    //                                    TD TH TW  BW MB NB
    if (Co == 64) return launch_f16s<2, 8, 16, 16, 2, 2>("conv3d_s1_f16s", a, s);
    return launch_f16s<2, 8, 16, 16, 2, 1>("conv3d_s1_f16s", a, s);
}

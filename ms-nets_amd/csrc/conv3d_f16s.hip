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
//   [voxel][ hi c0..c31 (64 B) | lo c0..c31 (64 B) ]      (128-byte records, 16-byte slots XOR-swizzled by (voxel>>1)&7
//                                                        so the 16-lane groups of ds_read_b128 hit 16 distinct banks)
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
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // first-class vector (HIP's uint4 is a class)

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

// SWZ = false: 144-byte voxel records (16 B pad): with 1x32-voxel M-blocks every ds_read_b128 lane group hits 16
//               distinct bank slots and all fragment addresses are base + immediate (no VALU in the MFMA stream).
// SWZ = true : 128-byte records with the 16-byte slots XOR-swizzled by (voxel>>1)&7 -- same conflict-freeness in
//               13 KB less LDS (what lets the Co=64 weight double buffer fit), at ~6 VALU per fragment address.
template <int TD, int TH, int TW, int BW, int MB, int NB, bool SWZ>
__global__ __launch_bounds__(512, 2) void conv3d_k3s1_f16s_ws(ConvArgs a) {
    constexpr int CC = 32;
    constexpr int BH = 32 / BW;
    constexpr int ID = TD + 2, IH = TH + 2, IW = TW + 2;
    constexpr int RB = SWZ ? 128 : 144;                 // bytes per voxel record in LDS (64 hi + 64 lo [+ 16 pad])
    static_assert(BW == 32, "bank-conflict analysis assumes M-blocks of 32 consecutive voxels");
    constexpr int MW = TW / BW, MH = TH / BH;
    constexpr int V = CC / 4;
    constexpr int NPOS = ID * IH * IW;
    constexpr int GB = 3 * 2 * NB * 2 * 1024;           // bytes of one weight group
    constexpr int NLB = GB / 16 / 256;                  // 16-byte pieces per loader thread per group
    static_assert(TD * MH * MW == 4 * MB, "M-block count mismatch");
    static_assert(NLB == 3 || NLB == 6, "weight group = 3 or 6 16-byte pieces per loader thread");
    static_assert(NPOS * RB + 2 * GB <= 160 * 1024, "LDS budget");
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
    const u32x4* wg = reinterpret_cast<const u32x4*>(a.wpk);     // split-fp16 packed weights

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
        // three weight-group register sets as plain first-class vectors (a ring of HIP `uint4` class objects was kept in
        // scratch by hipcc, putting a memory round trip and a vmcnt wait between the L2 load and the LDS copy)
        struct BSet { u32x4 v0, v1, v2, v3, v4, v5; };
        BSet bw0, bw1, bw2;
        // The tile is staged one input depth-plane at a time (PL float4 per loader thread per plane) so that the
        // copy of the NEXT tile into LDS can start before the current tile is finished: group order is kd-major, so
        // plane 0 is dead after groups 0-2 and plane 1 after groups 3-5; only planes 2.. wait for the b1/b2 window.
        // Per-slot constants (position inside a plane, global byte offset relative to the plane's tile origin, LDS
        // offsets) are computed once; per item a slot costs one add + one buffer load (hardware range check returns 0
        // for the lanes whose offset is forced out of range = conv zero padding / partial last slot).
        constexpr int PSLOT = IH * IW * V;              // float4 per plane
        constexpr int PL = (PSLOT + 255) / 256;
        static_assert(ID == 4, "plane schedule below assumes TD == 2");
        f32x4 av[ID][PL];
        unsigned goff_[PL];                             // global byte offset of the slot from the plane tile origin
        int ihw_[PL];                                   // (ih << 8) | iw, or -1 for a slot past the plane's end
#pragma unroll
        for (int u = 0; u < PL; ++u) {
            const int slot = u * 256 + lt;
            const int pos = slot / V, c4 = slot % V;
            const int ih = pos / IW, iw = pos % IW;
            const bool ok = slot < PSLOT;
            goff_[u] = (unsigned)(((ih * a.W + iw) * a.Ci + c4 * 4) * 4);
            ihw_[u] = ok ? ((ih << 8) | iw) : -1;
        }
        // LDS offset of this thread's slot u in plane pl: voxel = pl*IH*IW + u*32 + (lt>>3), channel quad c4 = lt & 7.
        // The swizzle term (voxel>>1)&7 does not depend on u (u*32 is a multiple of 16), only on the plane.
        static_assert((IH * IW) % 2 == 0, "plane size must be even for the per-plane swizzle below");
        int lhi_[ID];                                   // offset of the hi half for u = 0
#pragma unroll
        for (int pl = 0; pl < ID; ++pl) {
            const int p0 = lt >> 3, c4 = lt & 7;
            const int sw = SWZ ? (((p0 >> 1) + pl * (IH * IW / 2)) & 7) : 0;
            lhi_[pl] = (pl * IH * IW + p0) * RB + (c4 & 1) * 8 + (((c4 >> 1) ^ sw) << 4);
        }
        const size_t sample_bytes = (size_t)a.D * a.H * a.W * a.Ci * 4;

        auto issue_a = [&](int it, int pl) {
            int n, od0, oh0, ow0, chunk;
            decode(it, n, od0, oh0, ow0, chunk);
            const int gd = od0 - 1 + pl;
            const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(a.x) + (size_t)n * (sample_bytes / 4), 0, (int)sample_bytes, 0x00020000);
            // byte offset of voxel (gd, oh0-1, ow0-1), channel chunk*32, inside the sample (may wrap below zero; the
            // in-range lanes add a positive goff_ that brings it back -- unsigned arithmetic)
            const unsigned base = (unsigned)((((long)gd * a.H + (oh0 - 1)) * a.W + (ow0 - 1)) * a.Ci + chunk * CC) * 4u;
            const bool plane_ok = (unsigned)gd < (unsigned)a.D;
            const bool interior = oh0 >= 1 && oh0 + TH + 1 <= a.H && ow0 >= 1 && ow0 + TW + 1 <= a.W;
#pragma unroll
            for (int u = 0; u < PL; ++u) {
                bool ok = ihw_[u] >= 0;
                if (!interior) {
                    const int gh = oh0 - 1 + (ihw_[u] >> 8), gw = ow0 - 1 + (ihw_[u] & 255);
                    ok = ok && (unsigned)gh < (unsigned)a.H && (unsigned)gw < (unsigned)a.W;
                }
                const unsigned voff = (ok && plane_ok) ? base + goff_[u] : 0xffffffffu;
                const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
                av[pl][u] = __builtin_bit_cast(f32x4, raw);
            }
        };
        auto write_a = [&](int pl) {
#pragma unroll
            for (int u = 0; u < PL; ++u) {
                if (ihw_[u] >= 0) {
                    half4 hi, lo;
                    split4(av[pl][u], hi, lo);
                    const int off = lhi_[pl] + u * 32 * RB;
                    *reinterpret_cast<half4*>(lds + off) = hi;
                    *reinterpret_cast<half4*>(lds + (SWZ ? (off ^ 64) : off + 64)) = lo;
                }
            }
        };
        // Weight groups form one endless stream k = it*9 + g (chunk = it % nchunks).  Group k lives in register set
        // k % PD from the moment it is requested (while group k-PD-1 is multiplied, i.e. ~PD group times = several L2
        // latencies earlier) until it is copied into LDS buffer k & 1 (while group k-1 is multiplied).
        const int ngroups_total = nitems * 9;
        auto b_src = [&](int k) {
            k = k < ngroups_total ? k : ngroups_total - 1;     // past the end: harmless re-read
            return wg + (size_t)(((k / 9) % nchunks) * 9 + (k % 9)) * (GB / 16) + lt;
        };
#define MSNET_ISSUE_B(K, SET)                                                                     \
    do {                                                                                          \
        const u32x4* src_ = b_src(K);                                                             \
        SET.v0 = src_[0]; SET.v1 = src_[256]; SET.v2 = src_[512];                                 \
        if constexpr (NLB > 3) { SET.v3 = src_[768]; SET.v4 = src_[1024]; SET.v5 = src_[1280]; }  \
    } while (0)
#define MSNET_WRITE_B(K, SET)                                                                     \
    do {                                                                                          \
        u32x4* dst_ = reinterpret_cast<u32x4*>(lds_b + ((K) & 1) * GB) + lt;                      \
        dst_[0] = SET.v0; dst_[256] = SET.v1; dst_[512] = SET.v2;                                 \
        if constexpr (NLB > 3) { dst_[768] = SET.v3; dst_[1024] = SET.v4; dst_[1280] = SET.v5; }  \
    } while (0)
#ifndef EXP_NO_GROUP_BARRIER
#define MSNET_GROUP(G, SET)                     \
    MSNET_WRITE_B(k0 + (G) + 1, SET);           \
    MSNET_ISSUE_B(k0 + (G) + 1 + 3, SET);       \
    MSNET_LDS_BARRIER();
#else
#define MSNET_GROUP(G, SET)                     \
    MSNET_WRITE_B(k0 + (G) + 1, SET);           \
    MSNET_ISSUE_B(k0 + (G) + 1 + 3, SET);
#endif

        issue_a(0, 0); issue_a(0, 1); issue_a(0, 2); issue_a(0, 3);
        MSNET_ISSUE_B(0, bw0);
        MSNET_ISSUE_B(1, bw1);
        MSNET_ISSUE_B(2, bw2);
        bool early = false;                             // planes 0,1 of this item already copied during the previous one
        for (int it = 0; it < nitems; ++it) {
            const int k0 = it * 9;                      // 9 % 3 == 0: group k0+g always uses set g % 3
            const bool more = it + 1 < nitems;
            MSNET_LDS_BARRIER();                        // b1: MFMA waves are done with the previous tile
#ifndef EXP_NO_A_STAGE
            if (!early) { write_a(0); write_a(1); }
            write_a(2); write_a(3);
#endif
            MSNET_WRITE_B(k0, bw0);
            MSNET_ISSUE_B(k0 + 3, bw0);
            MSNET_LDS_BARRIER();                        // b2: tile and group 0 are in LDS
            // group g+1 is copied to LDS (and group g+4 requested) while group g is multiplied; barrier g_g ends it.
            // The next tile's planes are requested one per group and planes 0 / 1 copied as soon as they are dead.
#ifndef EXP_NO_A_STAGE
            if (more) issue_a(it + 1, 0);
#endif
            MSNET_GROUP(0, bw1)
#ifndef EXP_NO_A_STAGE
            if (more) issue_a(it + 1, 1);
#endif
            MSNET_GROUP(1, bw2)
            MSNET_GROUP(2, bw0)                         // g_2 passed: kd = 0 groups done, plane 0 is dead
#ifndef EXP_NO_A_STAGE
            if (more) write_a(0);
            if (more) issue_a(it + 1, 2);
#endif
            MSNET_GROUP(3, bw1)
#ifndef EXP_NO_A_STAGE
            if (more) issue_a(it + 1, 3);
#endif
            MSNET_GROUP(4, bw2)
            MSNET_GROUP(5, bw0)                         // g_5 passed: kd = 1 groups done, plane 1 is dead
#ifndef EXP_NO_A_STAGE
            if (more) write_a(1);
#endif
            MSNET_GROUP(6, bw1)
            MSNET_GROUP(7, bw2)
            early = more;
        }
#undef MSNET_GROUP
#undef MSNET_WRITE_B
#undef MSNET_ISSUE_B
        return;
    }

    // ------------------------------ MFMA waves ------------------------------
    const int wm = wave;                                // WM = 4, WN = 1
    const int r = lane & 31, hh = lane >> 5;
    int vox0[MB];                                       // LDS voxel index of this lane's output voxel (tap 0,0,0)
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int mb = wm * MB + i;
        const int bw_ = mb % MW, bh = (mb / MW) % MH, bd = mb / (MW * MH);
        const int lh = bh * BH + r / BW, lw = bw_ * BW + r % BW;
        vox0[i] = (bd * IH + lh) * IW + lw;
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
        // 6 steps per group (t = kw tap, ks = 16-channel K-step); fragments of step s+1 are read while step s multiplies.
        // The tile is stable across the group barriers, so the A fragments of a group's first step are read BEFORE the
        // barrier that publishes its weights; only the B fragments wait for it.
        half8 ah[2][MB], al[2][MB], bh_[2][NB], bl[2][NB];
        auto frag_a = [&](int s, int slot, int goff) {    // goff: voxel offset of the group's (kd, kh) row
            const int t = s >> 1, ks = s & 1;
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                if (SWZ) {
                    const int vox = vox0[i] + goff + t;
                    const int off = vox * RB + (((ks * 2 + hh) ^ ((vox >> 1) & 7)) << 4);
                    ah[slot][i] = *reinterpret_cast<const half8*>(lds + off);
                    al[slot][i] = *reinterpret_cast<const half8*>(lds + (off ^ 64));
                } else {
                    const unsigned char* p = lds + (vox0[i] + goff) * RB + 16 * hh + t * RB + ks * 32;
                    ah[slot][i] = *reinterpret_cast<const half8*>(p);
                    al[slot][i] = *reinterpret_cast<const half8*>(p + 64);
                }
            }
        };
        auto frag_b = [&](int s, int slot, const unsigned char* bb) {
            const int t = s >> 1, ks = s & 1;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const unsigned char* p = bb + (((t * 2 + ks) * NB + j) * 2) * 1024;
                bh_[slot][j] = *reinterpret_cast<const half8*>(p);
                bl[slot][j] = *reinterpret_cast<const half8*>(p + 1024);
            }
        };
        frag_a(0, 0, 0);
#pragma unroll 1
        for (int g = 0; g < 9; ++g) {
            const unsigned char* bb = lds_b + ((gg0 + g) & 1) * GB + lane * 16;
            const int goff = ((g / 3) * IH + (g % 3)) * IW;              // (kd, kh) row of this group, in voxels
            const int goff_next = (((g + 1) / 3) * IH + ((g + 1) % 3)) * IW;
            frag_b(0, 0, bb);
#pragma unroll
            for (int s = 0; s < 6; ++s) {
#ifndef EXP_NO_FRAG
                if (s + 1 < 6) { frag_a(s + 1, (s + 1) & 1, goff); frag_b(s + 1, (s + 1) & 1, bb); }
                else if (g < 8) frag_a(0, 0, goff_next);
#endif
                __builtin_amdgcn_sched_barrier(0);
#ifndef EXP_NO_MFMA
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        acc0[i][j] = mfma16(ah[s & 1][i], bh_[s & 1][j], acc0[i][j]);
                        acc1[i][j] = mfma16(al[s & 1][i], bh_[s & 1][j], acc1[i][j]);
                        acc1[i][j] = mfma16(ah[s & 1][i], bl[s & 1][j], acc1[i][j]);
                    }
#else
#pragma unroll
                for (int i = 0; i < MB; ++i) asm volatile("" ::"v"(ah[s & 1][i]), "v"(al[s & 1][i]));
#pragma unroll
                for (int j = 0; j < NB; ++j) asm volatile("" ::"v"(bh_[s & 1][j]), "v"(bl[s & 1][j]));
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
#ifndef EXP_NO_GROUP_BARRIER
            if (g < 8) MSNET_LDS_BARRIER();             // g_g
#endif
        }
        if (chunk == nchunks - 1) { pending = true; pn = n; pod0 = od0; poh0 = oh0; pow0 = ow0; }
    }
    if (pending) epilogue(pn, pod0, poh0, pow0);
}

template <int TD, int TH, int TW, int BW, int MB, int NB, bool SWZ>
static int launch_f16s(const char* name, ConvArgs a, hipStream_t s) {
    a.ntd = cdiv(a.OD, TD); a.nth = cdiv(a.OH, TH); a.ntw = cdiv(a.OW, TW);
    a.ngroups = 1; a.nbtot = a.Co / 32;
    const size_t ntiles = (size_t)a.N * a.ntd * a.nth * a.ntw;
    if (ntiles == 0 || ntiles > 0x7fffffffu) return fail("%s: bad tile count %zu", name, ntiles);
    const size_t nblk = ntiles < (size_t)num_cus() ? ntiles : (size_t)num_cus();
    const double vox = (double)a.N * a.OD * a.OH * a.OW;
    LaunchScope ls(name, s, 2.0 * 27.0 * a.Ci * a.Co * vox,
                   4.0 * ((double)a.N * a.D * a.H * a.W * a.Ci + vox * a.Co * (a.res ? 2 : 1)));
    hipLaunchKernelGGL((conv3d_k3s1_f16s_ws<TD, TH, TW, BW, MB, NB, SWZ>), dim3((unsigned)nblk), dim3(512), 0, s, a);
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
    if (Co == 64) return launch_f16s<2, 4, 32, 32, 2, 2, true>("conv3d_s1_f16s", a, s);
    return launch_f16s<2, 4, 32, 32, 2, 1, false>("conv3d_s1_f16s", a, s);
}

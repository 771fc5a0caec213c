// Split-fp16 3x3x3 convolution on the fp16 MFMA (v_mfma_f32_32x32x16_f16, 16x the fp32-MFMA rate).
//
// Every fp32 operand is written as  x = hi + lo * 2^-11  with  hi = fp16(x),  lo = fp16((x - hi) * 2^11)
// (22 significand bits; storing lo pre-scaled keeps it a NORMAL fp16 whenever hi is), and a product is
//     a*w  ~=  ah*wh  +  2^-11 * (al*wh + ah*wl)                (the dropped al*wl term is 2^-22 relative)
// i.e. three fp16 MFMAs into two fp32 accumulators (acc0: ah*wh, acc1: al*wh + ah*wl), combined once in the
// epilogue.  fp16 x fp16 products are exact in fp32, accumulation is fp32, so the result differs from the exact
// fp32 conv by ~3*2^-23 per product -- measured end to end on the parity fixtures this is below the fp32
// reference's own rounding noise (DESIGN.md "Numerics"); plain fp16 / bf16 / tf32 inputs are NOT (1e-2..1e-1).
// Requirement: |activation| < 32752 (half the fp16 range of `hi`: conv_common.h); BN+ReLU activations of these nets are O(1..100).
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
#include <utility>

#include "conv_common.h"

namespace msnet {

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N-1>{})
template <class F, int... Ks>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Ks...>) {
    (f(std::integral_constant<int, Ks>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

#ifndef SLIDE_LOADER_WAVES
#define SLIDE_LOADER_WAVES 4
#endif
#ifndef DEC_LOADER_WAVES
#define DEC_LOADER_WAVES 4     // 8 was measured: the 168-VGPR cap of a 768-thread workgroup spills the MFMA waves (1.10 -> 2.07 ms)
#endif
#ifndef S2_LOADER_WAVES
#define S2_LOADER_WAVES 8
#endif
#ifdef EXP_S2_PADDED
#define S2_SWZ false            // experiment: the round-2 stride-2 layout (padded 80-byte records, two weight buffers)
#else
#define S2_SWZ true
#endif
#ifndef MSNET_A_AUX
#define MSNET_A_AUX 0           // cache-policy bits of the loaders' tile requests (2 = nt, measured: see DESIGN.md 4.1d)
#endif
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // first-class vector (HIP's uint4 is a class)

__device__ __forceinline__ f32x16 mfma16(half8 a, half8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

#ifdef EXP_STAMP
// Diagnostic build only: cycle stamps of block 0 (wave 0 = MFMA, wave 4 = loader) at every barrier of the first items.
__device__ unsigned long long g_stamps[12][128];     // [wave][stamp]
__device__ __forceinline__ void stamp(int role, int& idx, int lane) {
    if (blockIdx.x == 0 && lane == 0 && idx < 128 && role < 12) g_stamps[role][idx] = __builtin_amdgcn_s_memtime();
    ++idx;
}
#define STAMP(role, idx, lane) stamp(role, idx, lane)
#else
#define STAMP(role, idx, lane) do {} while (0)
#endif

constexpr float kLoScale = 2048.f;          // 2^11
constexpr float kLoInv = 1.f / 2048.f;

// hi = fp16(x), lo = fp16((x - hi) * 2^11) for four values in TEN vector instructions (hipcc's own code for the plain C++ form
// below takes 16: it converts hi back to fp32 and multiplies separately): two packed conversions, four mixed-precision fmas
// that read the fp16 half directly (x - hi is exact in fp32), four fmas that scale, round to fp16 and write one half each.
// Bit-identical to the C++ form (tools/split_test.hip checks 8M random / denormal / large values on the device).
// Only for values that go to LDS next: hipcc cannot see what an asm statement executes, so it would not pad the wait states an
// MFMA needs behind a VALU write of its operand (the direct kernel, which feeds split values straight into MFMAs, read stale
// registers with this form) -- split4_cxx below is for those.
__device__ __forceinline__ void split4_cxx(const f32x4 v, half4& hi, half4& lo) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const _Float16 h = (_Float16)v[k];
        hi[k] = h;
        lo[k] = (_Float16)((v[k] - (float)h) * kLoScale);
    }
}
__device__ __forceinline__ void split4(const f32x4 v, half4& hi, half4& lo) {
#ifdef EXP_SPLIT_CXX
    split4_cxx(v, hi, lo);
#else
    unsigned h01, h23, l01, l23;
    float t0, t1, t2, t3;
    const float k = kLoScale;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h01) : "v"(v[0]), "v"(v[1]));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h23) : "v"(v[2]), "v"(v[3]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(t0) : "v"(h01), "v"(v[0]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(t1) : "v"(h01), "v"(v[1]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(t2) : "v"(h23), "v"(v[2]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(t3) : "v"(h23), "v"(v[3]));
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(l01) : "v"(t0), "v"(k));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(l01) : "v"(t1), "v"(k));
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(l23) : "v"(t2), "v"(k));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(l23) : "v"(t3), "v"(k));
    struct U2 { unsigned a, b; };
    hi = __builtin_bit_cast(half4, U2{h01, h23});
    lo = __builtin_bit_cast(half4, U2{l01, l23});
#endif
}

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

// Work-item counter of the persistent kernels.  Item `it` of a workgroup is (unit = lb + (it / per_unit) * G, pos = it % per_unit)
// and a unit is a mixed-radix number (channel group, w tile, h tile, d tile or depth segment, sample).  Decoding that from `it`
// costs six integer divisions by run-time values per item -- ~250 instructions, in waves that share a SIMD with an MFMA wave
// (and once more per weight group for the weight stream's address).  Items are visited in order, so the digits are ADVANCED
// instead: the stride G is decomposed once, `next()` is a handful of scalar add / compare / select.
struct TileCtr {
    int pos, cg, tw, th, td, n;                         // td: depth tile (plain kernels) or depth segment (sliding window)
    int s_cg, s_tw, s_th, s_td, s_n;                    // digits of the stride G
    int ncg, ntw, nth, ntd, per_unit;
    __device__ __forceinline__ void init(unsigned lb, unsigned G, int ncg_, int ntw_, int nth_, int ntd_, int per_unit_) {
        ncg = ncg_; ntw = ntw_; nth = nth_; ntd = ntd_; per_unit = per_unit_;
        unsigned t = lb;
        cg = t % ncg; t /= ncg; tw = t % ntw; t /= ntw; th = t % nth; t /= nth; td = t % ntd; n = t / ntd;
        t = G;
        s_cg = t % ncg; t /= ncg; s_tw = t % ntw; t /= ntw; s_th = t % nth; t /= nth; s_td = t % ntd; s_n = t / ntd;
        pos = 0;
    }
    __device__ __forceinline__ void next() {
        if (++pos < per_unit) return;
        pos = 0;
        cg += s_cg;      int c = cg >= ncg; cg -= c ? ncg : 0;
        tw += s_tw + c;  c = tw >= ntw;     tw -= c ? ntw : 0;
        th += s_th + c;  c = th >= nth;     th -= c ? nth : 0;
        td += s_td + c;  c = td >= ntd;     td -= c ? ntd : 0;
        n += s_n + c;
    }
};

// SWZ = false: 144-byte voxel records (16 B pad): with 1x32-voxel M-blocks every ds_read_b128 lane group hits 16
//               distinct bank slots and all fragment addresses are base + immediate (no VALU in the MFMA stream).
// SWZ = true : 128-byte records with the 16-byte slots XOR-swizzled by (tile column >> 1) & 7 -- same conflict-freeness in
//               13 KB less LDS (what lets the Co=64 weight double buffer fit), at ~6 VALU per fragment address.
// KS = 16-channel K-steps per staged chunk: 2 (32-channel chunks) or 1 (the 8-channel first layer, zero-padded to 16).
// RESB = true: all 27 taps of the (single-chunk) weight tensor stay resident in LDS for the whole kernel -- used when
//               they fit beside the tile (the 8-channel layer: 54 KB): no weight streaming, 2 barriers per item instead of 10.
// STRIDE = 1 or 2 (stride 2: the input tile is (2T+1)^3, so it is staged 16 channels at a time, KS = 1).
// SLIDE (single-chunk stride-1 layers): a workgroup walks a column of tiles along d, so consecutive tiles share two of
// their four input planes.  The LDS plane slots rotate by two per step (logical plane p of step j lives in slot
// (p + 2j) & 3); only the two new planes are fetched, split and copied -- into the slots of the two planes that die
// first -- and the two-barrier staging window between tiles is empty except at the start of a column.
// LW = loader waves (4, or 8 for the stride-2 layers whose staging work per MFMA is 2.5x that of the stride-1 layers).
template <int TD, int TH, int TW, int BW, int MB, int NB, bool SWZ, int KS, bool RESB, int STRIDE, bool SLIDE = false, int LW = 4>
__global__ __launch_bounds__(256 + 64 * LW, (256 + 64 * LW) / 256) void conv3d_k3s1_f16s_ws(ConvArgs a) {
    constexpr int LT = 64 * LW;                          // loader threads
    static_assert(!SLIDE || (STRIDE == 1 && !RESB && !SWZ && TD == 2), "sliding window: stride 1, streamed weights, padded records");
    constexpr int CC = 16 * KS;
    constexpr int BH = 32 / BW;
    constexpr int ID = (TD - 1) * STRIDE + 3, IH = (TH - 1) * STRIDE + 3, IW = (TW - 1) * STRIDE + 3;
    constexpr int HB = 2 * CC;                          // bytes of the hi (or lo) half of a voxel record
    constexpr int RB = SWZ ? 2 * HB : 2 * HB + 16;      // bytes per voxel record in LDS (hi + lo [+ 16 pad])
    // S2SWZ (stride 2, 16-channel chunks): 64-byte records [hi 32 B | lo 32 B] whose four 16-byte slots are XORed with
    // (record column >> 2) & 3 -- lanes of a ds_read_b128 group whose records share a bank base (every fourth record) then read
    // different slots, for any tap offset.  26 KB less LDS than the padded 80-byte records, which is what makes room for the
    // third weight buffer (B3) on the stride-2 kernel.
    constexpr bool S2SWZ = SWZ && STRIDE == 2 && KS == 1;
    static_assert(!SWZ || KS == 2 || S2SWZ, "the swizzle is written for 128-byte records (and 64-byte records at stride 2)");
    // M-blocks of 32 consecutive voxels read conflict-free; BW = 16 (two 16-voxel rows) leaves one of the four
    // ds_read_b128 lane groups 2-way conflicted on 4 lanes with the 128-byte swizzle -- accepted for the 16-mod-32 widths.
    static_assert(BW == 32 || (BW == 16 && SWZ), "M-block shapes the LDS layouts were checked for");
    constexpr int MW = TW / BW, MH = TH / BH;
    constexpr int V = CC / 4;                           // float4 per voxel record half-row (incl. zero padding)
    constexpr int NPOS = ID * IH * IW;
    constexpr int GB = 3 * KS * NB * 2 * 1024;          // bytes of one weight group
    constexpr int PG = GB / 16;                         // 16-byte pieces per weight group
    constexpr int NLB = (PG + LT - 1) / LT;               // pieces per loader thread per group
    static_assert(TD * MH * MW == 4 * MB, "M-block count mismatch");
    static_assert(NLB == 2 || NLB == 3 || NLB == 6, "weight group = 2, 3 or 6 16-byte pieces per loader thread");
    // Weight-group buffers in LDS.  B3 (three buffers, where they fit: the Co = 32 stride-1 kernels): group g+2 is copied while
    // group g is multiplied, so group g+1 has been in LDS since barrier g_(g-1) and its first B fragments are read BEFORE
    // barrier g_g, like the A fragments -- with two buffers every group started with an exposed LDS round trip behind its barrier.
    constexpr bool B3 = !RESB && ((STRIDE == 1 && NB == 1) || S2SWZ) && NPOS * RB + 3 * GB <= 160 * 1024;
    // KHS (the Co = 32 kernels: 1x32-voxel M-blocks, a wave's two M-blocks are adjacent h rows): weight groups are (kd, kw)
    // COLUMNS of the 3x3x3 stencil instead of (kd, kh) rows.  M-block 0 at tap row kh+1 reads the LDS row M-block 1 reads at kh,
    // so a 16-channel step loads four A rows once (8 ds_read_b128) and uses them for 3 kh x 2 M-blocks: 28 fragment reads per
    // 36 MFMAs instead of 36.  The packed weight image is unchanged; the loaders pick each group's three taps out of it.
#ifdef EXP_NO_KHS
    constexpr bool KHS = false;
#else
    constexpr bool KHS = B3 && KS == 2 && MB == 2 && BW == 32 && !SWZ && (TH / (32 / BW)) % 2 == 0 && LW == 4;
#endif
    constexpr int NBUF = RESB ? 9 : (B3 ? 3 : 2);
    static_assert(NPOS * RB + NBUF * GB <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) unsigned char lds[NPOS * RB + NBUF * GB];
    unsigned char* const lds_b = lds + NPOS * RB;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const unsigned G = gridDim.x;
    const unsigned lb = xcd_remap(blockIdx.x, G);
    const int nchunks = a.Ci < CC ? 1 : a.Ci / CC;
    const int ncg = a.ngroups;                          // output-channel groups of 32*NB channels
    // work units dealt to the workgroups: tiles (x nchunks items each), or for SLIDE column segments (x seglen items each)
    const int per_unit = SLIDE ? a.seglen : nchunks;
    const unsigned T = SLIDE ? (unsigned)a.N * a.nseg * a.nth * a.ntw * ncg : (unsigned)a.N * a.ntd * a.nth * a.ntw * ncg;
    const int my_tiles = (T > lb) ? (int)((T - lb + G - 1) / G) : 0;
    const int nitems = my_tiles * per_unit;
    if (nitems == 0) return;
    const u32x4* wg = reinterpret_cast<const u32x4*>(a.wpk);     // split-fp16 packed weights

    // coordinates of the item a counter points at
    auto coords = [&](const TileCtr& c, int& n, int& od0, int& oh0, int& ow0, int& chunk, int& cg) {
        n = c.n; cg = c.cg; ow0 = c.tw * TW; oh0 = c.th * TH;
        chunk = SLIDE ? 0 : c.pos;
        od0 = SLIDE ? (c.td * a.seglen + c.pos) * TD : c.td * TD;
    };
    TileCtr ctr0;
    ctr0.init(lb, G, ncg, a.ntw, a.nth, SLIDE ? a.nseg : a.ntd, per_unit);

    if (wave >= 4) {
        // ------------------------------ loader waves ------------------------------
        // (s_setprio(2) here was measured: the loader gets no faster and the MFMA groups slow down by ~10 %.)
        const int lt = tid - 256;
        // three weight-group register sets as plain first-class vectors (a ring of HIP `uint4` class objects was kept in
        // scratch by hipcc, putting a memory round trip and a vmcnt wait between the L2 load and the LDS copy)
        struct BSet { u32x4 v0, v1, v2, v3, v4, v5; };
        // Weight-group register sets.  Three sets = three groups of look-ahead; the six-piece groups of the Co = 64 kernels
        // (24 registers a set) use two -- two groups, ~4.6 K cycles, is still several L2 latencies -- which is what keeps their
        // loader inside the 256 registers of a 512-thread workgroup.  With two sets the set of group k is k & 1 and a tile has
        // nine groups, so the item body exists once per item parity (PAR below).
        constexpr int NSETS = NLB > 3 ? 2 : 3;
        constexpr int BA = B3 ? 2 : 1;
        static_assert(!B3 || NSETS == 3, "three LDS buffers go with three register sets");
        BSet bw[3];
#define MSNET_SETI(J, PAR) (NSETS == 3 ? (J) % 3 : (((J) + (PAR)) & 1))
        // The tile is staged one input depth-plane at a time (PL float4 per loader thread per plane) so that the
        // copy of the NEXT tile into LDS can start before the current tile is finished: group order is kd-major, so
        // plane 0 is dead after groups 0-2 and plane 1 after groups 3-5; only planes 2.. wait for the b1/b2 window.
        // Per-slot constants (position inside a plane, global byte offset relative to the plane's tile origin, LDS
        // offsets) are computed once; per item a slot costs one add + one buffer load (hardware range check returns 0
        // for the lanes whose offset is forced out of range = conv zero padding / partial last slot).
        // VR = float4 actually staged per voxel: the 8-channel first layer (RESB variant) only moves its 2 real quads;
        // the padding channels of its LDS records are zeroed once below and never touched again.
        constexpr int VR = RESB ? 2 : V;
        constexpr int PSLOT = IH * IW * VR;             // float4 per plane
        constexpr int PL = (PSLOT + LT - 1) / LT;
        static_assert(TD == 2 && (ID == 4 || ID == 5), "plane schedule below assumes TD == 2 (input planes d*S + kd)");
        f32x4 av[ID][PL];
        unsigned goff_[PL];                             // global byte offset of the slot from the plane tile origin
        unsigned mask0 = 0;                             // bit u: slot u exists (not past the plane's end, not a padding channel quad)
#pragma unroll
        for (int u = 0; u < PL; ++u) {
            const int slot = u * LT + lt;
            const int pos = slot / VR, c4 = slot % VR;
            const int ih = pos / IW, iw = pos % IW;
            const bool ok = slot < PSLOT && c4 * 4 < a.Ci;      // channels beyond Ci are zero padding
            goff_[u] = (unsigned)(((ih * a.W + iw) * a.Ci + c4 * 4) * 4);
            mask0 |= (ok ? 1u : 0u) << u;
        }
        // LDS offset of this thread's slot u in plane pl: voxel = pl*IH*IW + u*(LT/VR) + lt/VR, channel quad c4 = lt % VR.
        // Swizzled records: the 16-byte slot is XORed with (iw >> 1) & 7, iw = the voxel's COLUMN in the tile.  (Keying on the
        // linear voxel index instead made the two 16-voxel rows of a 2x16 M-block -- 18 voxels apart -- collide on two of the
        // 16 slots in every ds_read_b128 lane group: 31 % of the LDS cycles of the 2x8x16 kernel were conflict cycles.)
        static_assert(!SWZ || S2SWZ || (IW % 2 == 0), "an even row pitch keeps record parity = column parity");
        const int lhi0 = (lt / VR) * RB + ((lt % VR) & 1) * 8 + (((lt % VR) >> 1) << 4);   // hi half of slot u = 0 in plane slot 0 (padded records)
        // Stride 2: a lane's voxels are two columns apart, and with 16-byte-aligned records any padded layout then puts 16
        // lanes on 8 distinct bank slots (2-way conflict on every A read: 27 % of the kernel's LDS cycles).  The columns of a
        // tile row are therefore stored de-interleaved -- even columns first, then the odd ones -- so that a tap reads
        // consecutive records again (tap kw: parity kw & 1, start kw >> 1).
        constexpr bool CPERM = STRIDE == 2;
        constexpr int CHALF = (IW + 1) / 2;
        int lsw_[(SWZ || CPERM) ? PL : 1];              // in-plane LDS offset of slot u (swizzle / column permutation included)
        if constexpr (SWZ || CPERM) {
#pragma unroll
            for (int u = 0; u < PL; ++u) {
                const int slot = u * LT + lt, pos = slot / VR, c4 = slot % VR;
                const int ih = pos / IW, iw = pos % IW;
                const int col = CPERM ? (iw & 1) * CHALF + (iw >> 1) : iw;
                const int key = S2SWZ ? ((col >> 2) & 3) : SWZ ? ((iw >> 1) & 7) : 0;
                lsw_[u] = (ih * IW + col) * RB + (c4 & 1) * 8 + (((c4 >> 1) ^ key) << 4);
            }
        }
        const size_t sample_bytes = (size_t)a.D * a.H * a.W * a.Ci * 4;

        struct Coord { int n, od0, oh0, ow0, chunk; };
        auto coord_of = [&](const TileCtr& t) {
            Coord c;
            int cg_;
            coords(t, c.n, c.od0, c.oh0, c.ow0, c.chunk, cg_);
            return c;
        };
        TileCtr cur = ctr0, nxt = ctr0;                 // the current item and the one after it
        nxt.next();
        // Request slots [u0, u1) of input plane pl of the tile at c into the register set `dst`.  NO load sits inside a
        // branch: with loads on both sides of an if / else (edge vs interior tile, continuation vs column start, `if (more)`)
        // hipcc unified the destination registers at the join with v_mov copies of loads still in flight -- i.e. an
        // s_waitcnt vmcnt(0) in the loader's groups 0-2 that drained the next tile's HBM requests while the MFMA waves stood at
        // the group barrier (1100-2400 cycles per barrier in the per-wave stamps).  The validity of a slot is a bit of `mask`
        // (a plain register: the edge-tile branch only computes it), `live` = false turns the whole request into
        // out-of-range offsets (no memory traffic, zeros returned), and the plane index may be a run-time value.
        static_assert(PL <= 32, "slot validity mask");
        auto issue_a = [&](f32x4 (&dst)[PL], const Coord& c, int pl, bool live, int u0, int u1) {
            const int gd = c.od0 * STRIDE - 1 + pl;
            const int ih0 = c.oh0 * STRIDE - 1, iw0 = c.ow0 * STRIDE - 1;      // input origin of the tile
            const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(a.x) + (size_t)c.n * (sample_bytes / 4), 0, (int)sample_bytes, 0x00020000);
            // byte offset of voxel (gd, oh0-1, ow0-1), channel chunk*CC, inside the sample (may wrap below zero; the
            // in-range lanes add a positive goff_ that brings it back -- unsigned arithmetic)
            const unsigned base =
                (unsigned)((((long)gd * a.H + ih0) * a.W + iw0) * a.Ci + c.chunk * CC) * 4u;
            static_assert(PL * LT >= PSLOT, "slots cover the plane");
            const bool plane_ok = live && (unsigned)gd < (unsigned)a.D;
            const bool interior = ih0 >= 0 && ih0 + IH <= a.H && iw0 >= 0 && iw0 + IW <= a.W;
            unsigned mask = mask0;
            if (!interior) {                            // uniform branch, VALU only
                mask = 0;
#pragma unroll
                for (int u = 0; u < PL; ++u) {              // (slot position recomputed here: edge tiles only, no registers held)
                    const int pos = (u * LT + lt) / VR;
                    const int gh = ih0 + pos / IW, gw = iw0 + pos % IW;
                    const bool ok = (unsigned)gh < (unsigned)a.H && (unsigned)gw < (unsigned)a.W;
                    mask |= (ok ? 1u : 0u) << u;
                }
                mask &= mask0;
            }
            mask = plane_ok ? mask : 0u;
#ifdef EXP_NO_PLANE0
            if (STRIDE == 2 && pl == 0) mask = 0u;      // diagnostic (wrong numerics): what a d-sliding window would save in requests
#endif
#ifdef EXP_NO_A_LOAD
            if (STRIDE == 2) mask = 0u;                 // diagnostic (wrong numerics): requests go out dead, no memory traffic
#endif
#pragma unroll
            for (int u = 0; u < PL; ++u) {
                if (u < u0 || u >= u1) continue;
                const unsigned voff = ((mask >> u) & 1u) ? base + goff_[u] : 0xffffffffu;
                const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, MSNET_A_AUX);
                dst[u] = __builtin_bit_cast(f32x4, raw);
            }
        };
        // split + copy slots [u0, u1) of the register set `src` into LDS plane slot `pslot` (a run-time value in the sliding kernel).
        // pre = true: the set already holds hi|lo fp16 quads (presplit below), only the two LDS stores are left.
        auto write_a = [&](const f32x4 (&src)[PL], int pslot, int u0, int u1, auto prec) {
            constexpr bool PRE = decltype(prec)::value;
            struct H2 { half4 a, b; };
#pragma unroll
            for (int u = 0; u < PL; ++u) {
                if (u < u0 || u >= u1) continue;
#ifdef EXP_NO_A_WRITE
                if (STRIDE == 2) continue;              // diagnostic (wrong numerics): no split, no LDS copy
#endif
#ifdef EXP_PRESPLIT
                if constexpr (SLIDE) {                  // experiment: the input holds split records, a slot is 16 bytes of one verbatim
                    if (u * LT + lt < PSLOT)
                        *reinterpret_cast<f32x4*>(lds + pslot * (IH * IW * RB) + (lt / VR) * RB + (lt % VR) * 16 + u * (LT / VR) * RB) = src[u];
                    continue;
                }
#endif
                if (u * LT + lt < PSLOT) {
                    half4 hi, lo;
#ifdef EXP_NO_SPLIT
                    {   // diagnostic: pure copy (wrong numerics) -- what the loader costs without the split VALU work
                        const H2 t = __builtin_bit_cast(H2, src[u]);
                        hi = t.a; lo = t.b;
                    }
#else
                    if constexpr (PRE) { const H2 t = __builtin_bit_cast(H2, src[u]); hi = t.a; lo = t.b; }
                    else split4(src[u], hi, lo);
#endif
                    const int off = pslot * (IH * IW * RB) + ((SWZ || CPERM) ? lsw_[u] : lhi0 + u * (LT / VR) * RB);
                    *reinterpret_cast<half4*>(lds + off) = hi;
                    *reinterpret_cast<half4*>(lds + (SWZ ? (off ^ HB) : off + HB)) = lo;
                }
            }
        };
        // Stride 2: the planes that can only be copied in the b1/b2 window (they are read until the last group) are split in
        // registers during the last groups, under the MFMAs; the window -- in which the MFMA waves wait -- then holds only their
        // LDS stores (32->64: -2.4 %).  Measured on the stride-1 Co = 64 kernels too: +1 % (more spills), so not used there.
        constexpr bool PRESPLIT = STRIDE == 2 && ID == 5;
        auto presplit = [&](f32x4 (&v)[PL]) {
            struct H2 { half4 a, b; };
#pragma unroll
            for (int u = 0; u < PL; ++u) {
                half4 hi, lo;
                split4(v[u], hi, lo);
                v[u] = __builtin_bit_cast(f32x4, H2{hi, lo});
            }
        };
        constexpr std::integral_constant<bool, false> RAW{};
        [[maybe_unused]] constexpr std::integral_constant<bool, true> SPLIT{};
        // Weight groups form one endless stream k = it*9 + g (chunk = it % nchunks).  Group k lives in register set
        // k % PD from the moment it is requested (while group k-PD-1 is multiplied, i.e. ~PD group times = several L2
        // latencies earlier) until it is copied into LDS buffer k & 1 (while group k-1 is multiplied).
        // j = group index relative to the CURRENT item's first group (0..8: this item, 9..17: the next one; past the last
        // item the counter runs on and the read is a harmless one of some valid group)
        auto b_src = [&](int j) {
            const TileCtr& t = j < 9 ? cur : nxt;
            return wg + (size_t)((t.cg * nchunks + (SLIDE ? 0 : t.pos)) * 9 + (j < 9 ? j : j - 9)) * PG;
        };
        // piece index of this thread's u-th piece (clamped for the partial last piece of a 384-piece group)
        int bi_[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) bi_[u] = (PG % LT == 0 || u * LT + lt < PG) ? u * LT + lt : PG - 1;
#ifdef EXP_HALF_B
        constexpr bool EXP_HALF_B_ON = STRIDE == 2;     // diagnostic (wrong numerics): half the stride-2 weight stream
#else
        constexpr bool EXP_HALF_B_ON = false;
#endif
#ifdef EXP_BGLOB
#define MSNET_ISSUE_B(K, SET) do { (void)(SET); } while (0)
#define MSNET_WRITE_B(K, SET) do { (void)(SET); } while (0)
#else
// J: group index RELATIVE to the current item's first group (a compile-time constant at every call site)
#define MSNET_ISSUE_B(J, SET)                                                                                      \
    do {                                                                                                           \
        if constexpr (KHS) {    /* group j = (kd, kw): piece u of a thread is tap kh = u (256 pieces a tap) */     \
            constexpr int j_ = (J) < 9 ? (J) : (J) - 9;                                                            \
            const u32x4* base_ = b_src((J) < 9 ? 0 : 9) + (size_t)(((j_ / 3) * 9 + j_ % 3) * 256) + lt;            \
            SET.v0 = base_[0]; SET.v1 = base_[3 * 256]; SET.v2 = base_[6 * 256];                                   \
            break;                                                                                                 \
        }                                                                                                          \
        const u32x4* src_ = b_src(J);                                                                              \
        SET.v0 = src_[bi_[0]]; if (!EXP_HALF_B_ON) SET.v1 = src_[bi_[1]];                                          \
        if constexpr (NLB > 2) SET.v2 = src_[bi_[2]];                                                              \
        if constexpr (NLB > 3) { SET.v3 = src_[3 * LT + lt]; SET.v4 = src_[4 * LT + lt]; SET.v5 = src_[5 * LT + lt]; }    \
    } while (0)
#define MSNET_WRITE_B(K, SET)                                                                                      \
    do {                                                                                                           \
        u32x4* dst_ = reinterpret_cast<u32x4*>(lds_b + (B3 ? ((K) - k0) % 3 : ((K) & 1)) * GB);                    \
        dst_[bi_[0]] = SET.v0; if (!EXP_HALF_B_ON) dst_[bi_[1]] = SET.v1;                                          \
        if constexpr (NLB > 2) dst_[bi_[2]] = SET.v2;                                                              \
        if constexpr (NLB > 3) { dst_[3 * LT + lt] = SET.v3; dst_[4 * LT + lt] = SET.v4; dst_[5 * LT + lt] = SET.v5; }    \
    } while (0)
#endif
#ifndef EXP_NO_GROUP_BARRIER
// slot of group G: copy group G + BA (BA = 2 with three LDS buffers, else 1) and request the group NSETS later into the freed set
#ifdef EXP_B_LAST
#define MSNET_GROUP_FIRST(PAR)
#define MSNET_GROUP(G, PAR)                                                         \
    MSNET_WRITE_B(k0 + (G) + BA, bw[MSNET_SETI((G) + BA, PAR)]);                    \
    MSNET_ISSUE_B((G) + BA + NSETS, bw[MSNET_SETI((G) + BA, PAR)]);                 \
    STAMP(wave, sidx, lane);                                                        \
    MSNET_LDS_BARRIER();                                                            \
    STAMP(wave, sidx, lane);
#else
// The weight copy + request of a slot come FIRST in it (right behind the barrier that opens it), the tile requests and plane copies
// behind them: vmcnt counts in order, so a weight copy NSETS slots on then waits for tile requests up to the slot BEFORE its own
// request, not including that slot's (HBM) requests.
#define MSNET_GROUP_B(G, PAR)                                                       \
    MSNET_WRITE_B(k0 + (G) + BA, bw[MSNET_SETI((G) + BA, PAR)]);                    \
    MSNET_ISSUE_B((G) + BA + NSETS, bw[MSNET_SETI((G) + BA, PAR)]);
#define MSNET_GROUP_FIRST(PAR) MSNET_GROUP_B(0, PAR)
#define MSNET_GROUP(G, PAR)                                                         \
    STAMP(wave, sidx, lane);                                                        \
    MSNET_LDS_BARRIER();                                                            \
    STAMP(wave, sidx, lane);                                                        \
    if constexpr ((G) < 7) { MSNET_GROUP_B((G) + 1, PAR) }
#endif
#else
#define MSNET_GROUP_FIRST(PAR)
#define MSNET_GROUP(G, PAR)                                                         \
    MSNET_WRITE_B(k0 + (G) + BA, bw[MSNET_SETI((G) + BA, PAR)]);                    \
    MSNET_ISSUE_B((G) + BA + NSETS, bw[MSNET_SETI((G) + BA, PAR)]);
#endif
// b1/b2 window: two buffers -- group 0 of the item; three -- nothing (groups 0, 1 were copied during the previous item)
#define MSNET_WINDOW_B(PAR)                                                         \
    if constexpr (!B3) {                                                            \
        MSNET_WRITE_B(k0, bw[MSNET_SETI(0, PAR)]);                                  \
        MSNET_ISSUE_B(NSETS, bw[MSNET_SETI(0, PAR)]);                               \
    }
// behind g_7 (three buffers only): the next item's group 1 goes into the buffer group 7 has just released
#define MSNET_TAIL_B(PAR)                                                           \
    if constexpr (B3) {                                                             \
        MSNET_WRITE_B(k0 + 10, bw[MSNET_SETI(10, PAR)]);                            \
        MSNET_ISSUE_B(10 + NSETS, bw[MSNET_SETI(10, PAR)]);                         \
    }

        if constexpr (RESB) {
            for (int p = lt * 16; p < NPOS * RB; p += LT * 16)      // zero the records once (padding channels stay zero)
                *reinterpret_cast<u32x4*>(lds + p) = u32x4{0u, 0u, 0u, 0u};
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            // weights: one pass, all 9 groups, before the first tile is published
            for (int k = 0; k < 9; ++k) {               // (RESB is only used with a single channel group and chunk)
                const u32x4* src = wg + (size_t)k * PG;
                u32x4* dst = reinterpret_cast<u32x4*>(lds_b + k * GB);
                for (int p = lt; p < PG; p += LT) dst[p] = src[p];
            }
            {
                const Coord c0 = coord_of(cur);
#pragma unroll
                for (int pl = 0; pl < ID; ++pl) issue_a(av[pl], c0, pl, true, 0, PL);
            }
            for (int it = 0; it < nitems; ++it) {
                MSNET_LDS_BARRIER();                    // b1
#pragma unroll
                for (int pl = 0; pl < ID; ++pl) write_a(av[pl], pl, 0, PL, RAW);
                MSNET_LDS_BARRIER();                    // b2
                const Coord c = coord_of(nxt);
#pragma unroll
                for (int pl = 0; pl < ID; ++pl) issue_a(av[pl], c, pl, it + 1 < nitems, 0, PL);
                nxt.next();
            }
            return;
        }
        {
            const Coord c0 = coord_of(cur);
#pragma unroll
            for (int pl = 0; pl < ID; ++pl) issue_a(av[pl], c0, pl, true, 0, PL);
        }
        {
            const int k0 = 0;
            MSNET_ISSUE_B(0, bw[0]);
            MSNET_ISSUE_B(1, bw[1]);
            if constexpr (NSETS == 3) MSNET_ISSUE_B(2, bw[2]);
            if constexpr (B3) {                         // groups 0 and 1 of the first item (nobody reads the buffers before b2)
                MSNET_WRITE_B(0, bw[0]); MSNET_ISSUE_B(3, bw[0]);
                MSNET_WRITE_B(1, bw[1]); MSNET_ISSUE_B(4, bw[1]);
            }
        }
        if constexpr (SLIDE) {
            static_assert(!SLIDE || NSETS == 3, "the sliding kernel's groups are three pieces per thread");
            // Register sets by ROLE, not by plane: av[0], av[1] hold the next item's first two missing planes (its logical planes
            // 2, 3 inside a column, 0, 1 at a column start) -- either way they go into the LDS slots of the current item's
            // logical planes 0 and 1 (slots 2*rot, 2*rot + 1), which die after groups 2 / 5; av[2], av[3] hold planes 2, 3 of a
            // column start and are copied in that item's own b1/b2 window.
            constexpr int H0 = (PL + 2) / 3, H1 = (2 * PL + 2) / 3, HH = (PL + 1) / 2;
            bool early = false;                         // planes 0,1 of this column-start item were copied during the previous item
            int rot = 0;                                // plane-slot rotation of the current item
            [[maybe_unused]] int sidx = 0;
            for (int it = 0; it < nitems; ++it) {
                const int k0 = it * 9;
                const bool more = it + 1 < nitems;
                const bool cs = cur.pos == 0;           // the current item starts a column: its planes 2,3 (0,1) are not resident
                if (!cs) rot ^= 1;
                const bool ncont = more && nxt.pos != 0;
                const Coord nx = coord_of(nxt);
                const int p0 = ncont ? 2 : 0;           // first missing plane of the next item
                STAMP(wave, sidx, lane);
                MSNET_LDS_BARRIER();                    // b1: MFMA waves are done with the previous tile
                STAMP(wave, sidx, lane);
                if (cs) {                               // (LDS copies only inside the branches)
                    if (!early) { write_a(av[0], (2 * rot) & 3, 0, PL, RAW); write_a(av[1], (2 * rot + 1) & 3, 0, PL, RAW); }
                    write_a(av[2], (2 * rot + 2) & 3, 0, PL, RAW); write_a(av[3], (2 * rot + 3) & 3, 0, PL, RAW);
                }
                MSNET_WINDOW_B(0)
                STAMP(wave, sidx, lane);
                MSNET_LDS_BARRIER();                    // b2: tile and group 0 are in LDS
                STAMP(wave, sidx, lane);
                MSNET_GROUP_FIRST(0)
                issue_a(av[0], nx, p0, more, 0, PL); issue_a(av[1], nx, p0 + 1, more, 0, HH);
                MSNET_GROUP(0, 0)
                issue_a(av[1], nx, p0 + 1, more, HH, PL); issue_a(av[2], nx, 2, more && !ncont, 0, PL);
                MSNET_GROUP(1, 0)
                issue_a(av[3], nx, 3, more && !ncont, 0, PL);
                MSNET_GROUP(2, 0)                     // g_2 passed: this item's logical plane 0 (slot 2*rot) is dead
                write_a(av[0], 2 * rot, 0, H0, RAW);
                MSNET_GROUP(3, 0)
                write_a(av[0], 2 * rot, H0, H1, RAW);
                MSNET_GROUP(4, 0)
                write_a(av[0], 2 * rot, H1, PL, RAW);
                MSNET_GROUP(5, 0)                     // g_5 passed: logical plane 1 (slot 2*rot + 1) is dead
                write_a(av[1], 2 * rot + 1, 0, HH, RAW);
                MSNET_GROUP(6, 0)
                write_a(av[1], 2 * rot + 1, HH, PL, RAW);
                MSNET_GROUP(7, 0)
                MSNET_TAIL_B(0)
                early = more && !ncont;
                cur = nxt; nxt.next();
            }
            return;
        }
        bool early = false;                             // planes 0,1 of this item already copied during the previous one
        [[maybe_unused]] int sidx = 0;
#ifndef EXP_NO_A_STAGE
        if constexpr (PRESPLIT) { presplit(av[2]); presplit(av[3]); presplit(av[4]); }   // first item: its window expects hi|lo quads
#endif
        // The loader shares each SIMD with an MFMA wave and runs ~3x slower than alone, so its per-item work (28 loads,
        // 28 split+copy, 27 weight pieces) is spread evenly over the nine group slots instead of bunched at the barriers.
        constexpr int H0 = (PL + 2) / 3, H1 = (2 * PL + 2) / 3, HH = (PL + 1) / 2;
        auto item = [&](auto parc, const int it) {
            [[maybe_unused]] constexpr int PAR = decltype(parc)::value;  // it & 1 (two sets); unused with three (9 % 3 == 0: group k0+g uses set g % 3)
            const int k0 = it * 9;
            const bool more = it + 1 < nitems;
            STAMP(wave, sidx, lane);
            MSNET_LDS_BARRIER();                        // b1: MFMA waves are done with the previous tile
            STAMP(wave, sidx, lane);
#ifndef EXP_NO_A_STAGE
            if (!early) { write_a(av[0], 0, 0, PL, RAW); write_a(av[1], 1, 0, PL, RAW); }
            if constexpr (PRESPLIT) { write_a(av[2], 2, 0, PL, SPLIT); write_a(av[3], 3, 0, PL, SPLIT); write_a(av[4], 4, 0, PL, SPLIT); }
            else { write_a(av[2], 2, 0, PL, RAW); write_a(av[3], 3, 0, PL, RAW); }
#endif
            MSNET_WINDOW_B(PAR)
            STAMP(wave, sidx, lane);
            MSNET_LDS_BARRIER();                        // b2: tile and group 0 are in LDS
            STAMP(wave, sidx, lane);
            MSNET_GROUP_FIRST(PAR)
            // group g+1 is copied to LDS (and group g+4 requested) while group g is multiplied; barrier g_g ends it.
            // The next tile is requested during groups 0-2; its planes 0 / 1 are copied as soon as they are dead.  (Past the
            // last item the requests are dead -- `more` = false -- and the copies put zeros into planes nobody reads again.)
            const Coord nx = coord_of(nxt);
            // Stride 2 (five planes, 104 KB per item and CU): the next tile's requests go out evenly over all eight slots, SPS per
            // thread and slot.  What limits this kernel is the rate at which a CU can take in lines that miss its L1 -- ~12 B/clk,
            // i.e. ~13 KB per group: with everything in groups 0-2 (or a plane per group in 0-4) single buffer loads took
            // 300-500 cycles to ISSUE, the loader waves reached the group barriers late and the MFMA waves sat there; a
            // probe (-DEXP_LAT_PROBE) shows the data back ~400 cycles after the last request of a slot has been accepted.
            constexpr bool SPREAD = ID > 4;
            constexpr int SPS = SPREAD ? (ID * PL + 7) / 8 : 0;
            [[maybe_unused]] auto issue_seq = [&](auto slotc) {     // requests [slot*SPS, slot*SPS + SPS) of the plane-major sequence
                constexpr int k0 = decltype(slotc)::value * SPS;
                static_for<SPS>([&](auto kc) {
                    constexpr int k = k0 + decltype(kc)::value;
                    if constexpr (k < ID * PL) issue_a(av[k / PL], nx, k / PL, more, k % PL, k % PL + 1);
                });
            };
#define MSNET_SEQ(S) issue_seq(std::integral_constant<int, S>{})
#ifndef EXP_NO_A_STAGE
            if constexpr (SPREAD) MSNET_SEQ(0);
            else { issue_a(av[0], nx, 0, more, 0, PL); issue_a(av[1], nx, 1, more, 0, HH); }
#endif
#ifdef EXP_LAT_PROBE
            // diagnostic: how long until the requests just issued (and everything older) have returned
            STAMP(wave, sidx, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            STAMP(wave, sidx, lane);
#endif
            MSNET_GROUP(0, PAR)
#ifndef EXP_NO_A_STAGE
            if constexpr (SPREAD) MSNET_SEQ(1);
            else { issue_a(av[1], nx, 1, more, HH, PL); issue_a(av[2], nx, 2, more, 0, PL); }
#endif
            MSNET_GROUP(1, PAR)
#ifndef EXP_NO_A_STAGE
            if constexpr (SPREAD) MSNET_SEQ(2);
            else issue_a(av[3], nx, 3, more, 0, PL);
#endif
            MSNET_GROUP(2, PAR)                         // g_2 passed: kd = 0 groups done, plane 0 is dead
#ifndef EXP_NO_A_STAGE
            if constexpr (SPREAD) MSNET_SEQ(3);
            write_a(av[0], 0, 0, H0, RAW);
#endif
            MSNET_GROUP(3, PAR)
#ifndef EXP_NO_A_STAGE
            if constexpr (SPREAD) MSNET_SEQ(4);
            write_a(av[0], 0, H0, H1, RAW);
#endif
            MSNET_GROUP(4, PAR)
#ifndef EXP_NO_A_STAGE
            if constexpr (SPREAD) MSNET_SEQ(5);
            write_a(av[0], 0, H1, PL, RAW);
#endif
            MSNET_GROUP(5, PAR)                         // g_5 passed: kd = 1 groups done, plane 1 is dead
#ifndef EXP_NO_A_STAGE
            if constexpr (SPREAD) MSNET_SEQ(6);
            write_a(av[1], 1, 0, HH, RAW);
            if constexpr (PRESPLIT) presplit(av[2]);
#endif
            MSNET_GROUP(6, PAR)
#ifndef EXP_NO_A_STAGE
            if constexpr (SPREAD) MSNET_SEQ(7);
            write_a(av[1], 1, HH, PL, RAW);
            if constexpr (PRESPLIT) presplit(av[3]);
#endif
            MSNET_GROUP(7, PAR)
#undef MSNET_SEQ
            MSNET_TAIL_B(PAR)
#ifndef EXP_NO_A_STAGE
            if constexpr (PRESPLIT) presplit(av[4]);    // (its last request went out in slot 7: split behind g_7, before b1)
#endif
            early = more;
            cur = nxt; nxt.next();
        };
        if constexpr (NSETS == 3) {
            for (int it = 0; it < nitems; ++it) item(std::integral_constant<int, 0>{}, it);
        } else {
            for (int it = 0; it < nitems; it += 2) {
                item(std::integral_constant<int, 0>{}, it);
                if (it + 1 < nitems) item(std::integral_constant<int, 1>{}, it + 1);
            }
        }
#undef MSNET_GROUP
#undef MSNET_GROUP_FIRST
#ifdef MSNET_GROUP_B
#undef MSNET_GROUP_B
#endif
#undef MSNET_WINDOW_B
#undef MSNET_TAIL_B
#undef MSNET_WRITE_B
#undef MSNET_ISSUE_B
#undef MSNET_SETI
        return;
    }

    // ------------------------------ MFMA waves ------------------------------
#ifdef EXP_MFMA_PRIO
    __builtin_amdgcn_s_setprio(EXP_MFMA_PRIO);          // experiment: static priority of the MFMA waves over the loader waves
#endif
    const int wm = wave;                                // WM = 4, WN = 1
    const int r = lane & 31, hh = lane >> 5;
    int vox0[MB];                                       // LDS voxel index of this lane's output voxel (tap 0,0,0)
    [[maybe_unused]] int lwv[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int mb = wm * MB + i;
        const int bw_ = mb % MW, bh = (mb / MW) % MH, bd = mb / (MW * MH);
        const int lh = bh * BH + r / BW, lw = bw_ * BW + r % BW;
        // SLIDE: the plane comes from grp_off.  Stride 2 with de-interleaved columns: output column lw reads record lw of
        // the even half (kw = 0, 2) or of the odd half (kw = 1), see tap_col below.
        vox0[i] = ((SLIDE ? 0 : bd * STRIDE * IH) + lh * STRIDE) * IW + lw;      // (stride 2: de-interleaved columns, record lw)
        lwv[i] = lw;                                    // record column of the lane's voxel at kw = 0 (swizzle key)
    }
    int rot = 0;                                        // SLIDE: plane-slot rotation of the current item
    // voxel offset of group g's (kd, kh) row for M-block i
    auto grp_off = [&](int g, int i) {
        if (SLIDE) {
            const int bd = (wm * MB + i) / (MW * MH);
            return ((((bd + g / 3 + 2 * rot) & 3) * IH) + g % 3) * IW;
        }
        return ((g / 3) * IH + (g % 3)) * IW;
    };
    const int stride_w = a.Co, stride_h = a.OW * a.Co;

    f32x16 acc0[MB][NB], acc1[MB][NB];
    int pn = 0, pod0 = 0, poh0 = 0, pow0 = 0, pcg = 0;  // coordinates of the item whose epilogue is pending
    bool pending = false;

    // Epilogue of a finished tile over buffer descriptors (an element outside the tensor gets offset 0xffffffff: load 0 / store
    // dropped, no branch).  vmcnt counts stores as well as loads on this part, so nothing here may wait for "all loads": the
    // earlier form joined an optional residual load with the stores of every 32x32 block, and the s_waitcnt vmcnt(0) at that
    // join made each block of 16 stores wait for the ACKNOWLEDGEMENT of the previous block's stores (the per-wave stamps showed
    // 5.0-5.7 K cycles for the 64 KB of a Co = 64 tile).  Without a residual there is no load and no wait at all; with one, the
    // residual of block b+1 is requested before block b is stored, so the counted wait for it leaves b's stores in flight.
    const size_t osample = (size_t)a.OD * a.OH * a.OW * a.Co * 4;
    auto epilogue = [&](int n, int od0, int oh0, int ow0, int cg) {
        const auto rs_y = make_rsrc(a.y + (size_t)n * (osample / 4), osample);
        constexpr int NBLK = MB * NB;
        auto geom = [&](int b, unsigned& off, bool& rowok, int& hlim, int& wlim, float& sc, float& sh) {
            const int i = b / NB, j = b % NB;
            const int mb = wm * MB + i;
            const int bw_ = mb % MW, bh = (mb / MW) % MH, bd = mb / (MW * MH);
            const int od = od0 + bd, ohb = oh0 + bh * BH, owb = ow0 + bw_ * BW + 4 * hh;
            const int co = (cg * NB + j) * 32 + r;
            sc = a.scale ? a.scale[co] : 1.f;
            sh = a.shift ? a.shift[co] : 0.f;
            rowok = od < a.OD;
            hlim = a.OH - ohb; wlim = a.OW - owb;
            off = (unsigned)((((size_t)od * a.OH + ohb) * a.OW + owb) * a.Co + co) * 4u;
        };
        auto block_acc = [&](int b) {
            f32x16 v;
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = acc0[b / NB][b % NB][e] + acc1[b / NB][b % NB][e] * kLoInv;
            return v;
        };
        if (!a.res) {
            f32x16 zero;
#pragma unroll
            for (int e = 0; e < 16; ++e) zero[e] = 0.f;
#pragma unroll
            for (int b = 0; b < NBLK; ++b) {
                unsigned off; bool rowok; int hlim, wlim; float sc, sh;
                geom(b, off, rowok, hlim, wlim, sc, sh);
                epilogue_store<BW>(block_acc(b), zero, sc, sh, rs_y, off, stride_h * 4, stride_w * 4, a.relu,
                                   [&](int lh, int lw) { return rowok && lh < hlim && lw < wlim; }, a.oflag);
            }
        } else {
            const auto rs_res = make_rsrc(a.res + (size_t)n * (osample / 4), osample);
            f32x16 rv[2];
            {
                unsigned off; bool rowok; int hlim, wlim; float sc, sh;
                geom(0, off, rowok, hlim, wlim, sc, sh);
                residual_prefetch<BW>(rv[0], rs_res, off, stride_h * 4, stride_w * 4, [&](int lh, int lw) { return rowok && lh < hlim && lw < wlim; });
            }
#pragma unroll
            for (int b = 0; b < NBLK; ++b) {
                if (b + 1 < NBLK) {
                    unsigned off; bool rowok; int hlim, wlim; float sc, sh;
                    geom(b + 1, off, rowok, hlim, wlim, sc, sh);
                    residual_prefetch<BW>(rv[(b + 1) & 1], rs_res, off, stride_h * 4, stride_w * 4,
                                          [&](int lh, int lw) { return rowok && lh < hlim && lw < wlim; });
                }
                unsigned off; bool rowok; int hlim, wlim; float sc, sh;
                geom(b, off, rowok, hlim, wlim, sc, sh);
                epilogue_store<BW>(block_acc(b), rv[b & 1], sc, sh, rs_y, off, stride_h * 4, stride_w * 4, a.relu,
                                   [&](int lh, int lw) { return rowok && lh < hlim && lw < wlim; }, a.oflag);
            }
        }
    };

    // DRAIN (sliding-window kernel, layers without a residual): the finished tile is not stored in a burst at the
    // hand-over (32 KB per CU against a store path of ~16 B/clk: ~2000 cycles during which the MFMA waves do nothing
    // else, and after the sliding window that burst IS the hand-over) but parked in `pend` and stored one element
    // per K-step under the next tile's first groups.
    constexpr bool DRAIN = SLIDE && LW == 4;             // (more loader waves leave no registers for the parked tile)
    constexpr int PIECES = MB * NB * 16;
    static_assert(!DRAIN || PIECES <= 9 * 3 * KS, "a tile's groups must cover the previous tile's pieces");
    f32x16 pend[DRAIN ? MB : 1][DRAIN ? NB : 1];
    unsigned pbase[MB][NB];
    int plh[MB], plw[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) { plh[i] = 0; plw[i] = 0; }
    bool pend_live = false;
    float psc[NB], psh[NB], pamax = 0.f;                 // parked tile: per-channel scale / shift, running max magnitude
#pragma unroll
    for (int j = 0; j < NB; ++j) { psc[j] = 1.f; psh[j] = 0.f; }
    __amdgpu_buffer_rsrc_t pend_rs = make_rsrc(a.y, 0);
    auto park = [&](int n, int od0, int oh0, int ow0, int cg) {
        pend_rs = make_rsrc(a.y + (size_t)n * (osample / 4), osample);
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const int mb = wm * MB + i;
            const int bw_ = mb % MW, bh = (mb / MW) % MH, bd = mb / (MW * MH);
            const int od = od0 + bd, ohb = oh0 + bh * BH, owb = ow0 + bw_ * BW + 4 * hh;
            plh[i] = a.OH - ohb; plw[i] = a.OW - owb;
            // one compare per drained store: rows of this M-block beyond the tensor (or a whole M-block beyond its depth) get
            // a column limit of zero (BW == 32: an M-block is one row, lh is always 0)
            if (od >= a.OD || plh[i] <= 0) plw[i] = 0;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int co = (cg * NB + j) * 32 + r;
                pbase[i][j] = od < a.OD ? (unsigned)((((size_t)od * a.OH + ohb) * a.OW + owb) * a.Co + co) * 4u : 0xffffffffu;
                psc[j] = a.scale ? a.scale[co] : 1.f;
                psh[j] = a.shift ? a.shift[co] : 0.f;
                // only the hi/lo combine happens here (the MFMA pipe idles while the tile is parked); scale, shift, ReLU and
                // the range check ride with the drained stores, one element per K-step between the next tile's MFMAs
                f32x16 t;
#pragma unroll
                for (int e = 0; e < 16; ++e) t[e] = acc0[i][j][e] + acc1[i][j][e] * kLoInv;
                pend[DRAIN ? i : 0][DRAIN ? j : 0] = t;
            }
        }
        pend_live = true;
    };
    auto drain_piece = [&](auto qc) {
        constexpr int q = decltype(qc)::value;
        if constexpr (DRAIN && q < PIECES) {
            constexpr int e = q % 16, j = (q / 16) % NB, i = q / (16 * NB);
            constexpr int c = (e & 3) + 8 * (e >> 2), lh = c / BW, lw = c % BW;
#ifndef EXP_NO_SGB
            {   // branch-free: with nothing parked (plw == 0) the offset is out of range and the store is dropped
                static_assert(BW == 32 || !DRAIN, "drained stores assume one-row M-blocks");
                const bool ok = lw < plw[i];
#else
            if (pend_live) {
                const bool ok = pbase[i][j] != 0xffffffffu && lh < plh[i] && lw < plw[i];
#endif
                const unsigned o = ok ? pbase[i][j] + (unsigned)(lh * stride_h + lw * stride_w) * 4u : 0xffffffffu;
                float val = pend[i][j][e] * psc[j] + psh[j];      // (bit_cast applied to the vector element itself reads element 0)
                if (a.relu) val = fmaxf(val, 0.f);
                pamax = fmaxf(pamax, ok ? fabsf(val) : 0.f);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), pend_rs, o, 0, 0);
            }
        }
    };

#ifdef EXP_BGLOB
    // Experiment: the MFMA waves stream the B operand (weights) straight from L2 / L1 into a register ring BR steps deep
    // instead of reading it from the LDS image the loaders maintain.
    constexpr int NSB = 3 * KS;
    constexpr int BR = (NB == 1 && KS == 2) ? 6 : 3;
    constexpr int PFB = BR - 1;
    static_assert(NSB % BR == 0, "B ring must tile the group");
    half8 bgh[BR][NB], bgl[BR][NB];
    auto frag_bg = [&](int s_, int slot, const u32x4* base) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const u32x4* p_ = base + ((s_ * NB + j) * 2) * 64;
            bgh[slot][j] = __builtin_bit_cast(half8, p_[0]);
            bgl[slot][j] = __builtin_bit_cast(half8, p_[64]);
        }
    };
    auto wbase_of = [&](int it_) {                      // (experiment only: decodes with divisions)
        const unsigned t_ = lb + (unsigned)((it_ < nitems ? it_ : nitems - 1) / per_unit) * G;
        const int ch_ = SLIDE ? 0 : (it_ < nitems ? it_ : nitems - 1) % nchunks;
        return wg + (size_t)(((int)(t_ % ncg) * nchunks + ch_) * 9) * PG + lane;
    };
    {
        const u32x4* w0 = wbase_of(0);
#pragma unroll
        for (int q = 0; q < PFB; ++q) frag_bg(q, q, w0);
    }
#endif
    [[maybe_unused]] int sidx = 0;
    TileCtr ctr = ctr0;
#ifdef EXP_STAGGER
    // Experiment (measured +-1 %): identical persistent workgroups run in lockstep, so all 256 CUs reach their
    // epilogues together and the 64 KB store bursts of a Co = 64 tile queue on HBM (5.2 K cycles per tile in the per-wave
    // stamps).  Delaying the workgroups of phase blockIdx & 3 by phase * stagger spreads the bursts; the loaders wait at b1.
    if (a.stagger) {
        const int nsl = (int)(blockIdx.x & 3u) * a.stagger;
        for (int q = 0; q < nsl; ++q) __builtin_amdgcn_s_sleep(16);
    }
#endif
    for (int it = 0; it < nitems; ++it) {
        int n, od0, oh0, ow0, chunk, cg;
        coords(ctr, n, od0, oh0, ow0, chunk, cg);
        if (SLIDE && ctr.pos != 0) rot ^= 1;             // next tile of the same column
        ctr.next();
        STAMP(wave, sidx, lane);
        MSNET_LDS_BARRIER();                            // b1
        STAMP(wave, sidx, lane);
        if (pending) {
            if (DRAIN && !a.res) park(pn, pod0, poh0, pow0, pcg);
            else epilogue(pn, pod0, poh0, pow0, pcg);
            pending = false;
        }
        STAMP(wave, sidx, lane);
        MSNET_LDS_BARRIER();                            // b2
        STAMP(wave, sidx, lane);
        if (chunk == 0) {
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) { acc0[i][j][e] = 0.f; acc1[i][j][e] = 0.f; }
        }
        if constexpr (KHS) {
            // ---- (kd, kw) column groups, A rows shared between the wave's two M-blocks (see KHS above) ----
            // step q = ((g*2 + ks)*3 + kh): super-step S = g*2 + ks holds rows k = 0..3 (input rows bh0 + k at column offset kw,
            // 16 channels) in row set S & 1; M-block i multiplies row kh + i with the weights of tap (kd, kh, kw).
            static_assert(NB == 1 && KS == 2 && MB == 2, "KHS shapes");
            int pofs[3];                                // voxel offset of the lane's output-depth plane for kd = 0..2
#pragma unroll
            for (int kd = 0; kd < 3; ++kd)
                pofs[kd] = SLIDE ? ((((wm * MB) / (MW * MH) + kd + 2 * rot) & 3) * IH) * IW : kd * IH * IW;
            const unsigned char* const arow0 = lds + vox0[0] * RB + 16 * hh;
            half8 rh[2][4], rl[2][4], qh[3], ql[3];
            auto ld_row = [&](auto setc, auto kc, auto gc, auto ksc) {
                constexpr int set = decltype(setc)::value, k = decltype(kc)::value, g = decltype(gc)::value, ks = decltype(ksc)::value;
                const unsigned char* p_ = arow0 + (pofs[g / 3] + k * IW + g % 3) * RB + ks * 32;
                rh[set][k] = *reinterpret_cast<const half8*>(p_);
                rl[set][k] = *reinterpret_cast<const half8*>(p_ + HB);
            };
            auto ld_b = [&](auto qc) {                  // B fragments of step q into ring slot q % 3
                constexpr int q = decltype(qc)::value, g = q / 6, ks = (q / 3) % 2, kh = q % 3;
                const unsigned char* p_ = lds_b + (g % 3) * GB + lane * 16 + ((kh * KS + ks) * NB) * 2 * 1024;
                qh[q % 3] = *reinterpret_cast<const half8*>(p_);
                ql[q % 3] = *reinterpret_cast<const half8*>(p_ + 1024);
            };
            using I0 = std::integral_constant<int, 0>;
            static_for<4>([&](auto kc) { ld_row(I0{}, kc, I0{}, I0{}); });
            ld_b(std::integral_constant<int, 0>{});
            ld_b(std::integral_constant<int, 1>{});
            static_for<9>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                static_for<6>([&](auto sc_) {
                    constexpr int s_ = decltype(sc_)::value, ks = s_ / 3, kh = s_ % 3;
                    constexpr int S = g * 2 + ks, q = S * 3 + kh;
                    // prefetch: rows of super-step S+1 (two rows at kh = 0, one each at kh = 1, 2), B fragments of step q+2
                    constexpr int S1 = S + 1;
                    [[maybe_unused]] constexpr int nrows = kh == 0 ? 2 : 1;
                    if constexpr (S1 < 18) {
                        using SETC = std::integral_constant<int, S1 & 1>;
                        using G1 = std::integral_constant<int, S1 / 2>;
                        using K1 = std::integral_constant<int, S1 % 2>;
                        if constexpr (kh == 0) { ld_row(SETC{}, std::integral_constant<int, 0>{}, G1{}, K1{}); ld_row(SETC{}, std::integral_constant<int, 1>{}, G1{}, K1{}); }
                        else ld_row(SETC{}, std::integral_constant<int, kh + 1>{}, G1{}, K1{});
                    }
                    if constexpr (q + 2 < 54) ld_b(std::integral_constant<int, q + 2>{});
#ifndef EXP_NO_MFMA
                    acc0[0][0] = mfma16(rh[S & 1][kh], qh[q % 3], acc0[0][0]);
                    acc1[0][0] = mfma16(rl[S & 1][kh], qh[q % 3], acc1[0][0]);
                    acc1[0][0] = mfma16(rh[S & 1][kh], ql[q % 3], acc1[0][0]);
                    acc0[1][0] = mfma16(rh[S & 1][kh + 1], qh[q % 3], acc0[1][0]);
                    acc1[1][0] = mfma16(rl[S & 1][kh + 1], qh[q % 3], acc1[1][0]);
                    acc1[1][0] = mfma16(rh[S & 1][kh + 1], ql[q % 3], acc1[1][0]);
#endif
                    if constexpr (DRAIN) drain_piece(std::integral_constant<int, q>{});
                    {   // interleave as in the row-group loop: one LDS read and two VALU behind each MFMA
                        constexpr int NRD_ = (S1 < 18 ? 2 * nrows : 0) + (q + 2 < 54 ? 2 : 0);
#pragma unroll
                        for (int m = 0; m < 6; ++m) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            if (m < NRD_) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                            if (m == 4) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
#ifndef EXP_NO_GROUP_BARRIER
                if constexpr (g < 8) {
                    // g_g: in flight are the A rows and B fragments of group g+1's first steps -- live tile planes and the weight
                    // buffer published at g_(g-1), neither of which a loader writes before g_(g+1): no drain (see do_group)
                    STAMP(wave, sidx, lane);
#ifdef EXP_FULL_GROUP_BARRIER
                    MSNET_LDS_BARRIER();
#else
                    MSNET_READER_BARRIER();
#endif
                    STAMP(wave, sidx, lane);
                }
#endif
            });
            if constexpr (DRAIN) {
                if (pend_live) flag_overflow(a.oflag, pamax);
                pamax = 0.f;
                pend_live = false;
#pragma unroll
                for (int i = 0; i < MB; ++i) plw[i] = 0;
            }
        } else {
        const int gg0 = it * 9;
        // 3*KS steps per group (t = kw tap, ks = 16-channel K-step); fragments of step s+1 are read while step s multiplies.
        // The tile is stable across the group barriers, so the A fragments of a group's first step are read BEFORE the
        // barrier that publishes its weights; only the B fragments wait for it.
        constexpr int NS = 3 * KS;
        // Fragment ring of R slots: R = 3 (two steps of look-ahead) where registers allow, else 2.  Step s of any group
        // uses slot s % R (NS % R == 0), so the A fragments of the next group's first R-1 steps can be read before the
        // barrier that publishes its weights.
        constexpr int R = (NB == 1 || KS == 1) ? 3 : 2;
        constexpr int PF = R - 1;
        static_assert(NS % R == 0 && PF <= NS, "fragment ring must tile the group");
        half8 ah[R][MB], al[R][MB], bh_[R][NB], bl[R][NB];
        auto frag_a = [&](int s, int slot, const int (&goffs)[MB]) {    // goffs[i]: voxel offset of the group's (kd, kh) row
            const int t = s / KS, ks = s % KS;
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                const int goff = goffs[i];
                if constexpr (S2SWZ) {
                    constexpr int CH = (IW + 1) / 2;
                    const int tcol = (t & 1) * CH + (t >> 1);                            // record offset of tap kw = t
                    const int off = (vox0[i] + goff + tcol) * RB + ((hh ^ (((lwv[i] + tcol) >> 2) & 3)) << 4);
                    ah[slot][i] = *reinterpret_cast<const half8*>(lds + off);
                    al[slot][i] = *reinterpret_cast<const half8*>(lds + (off ^ HB));
                } else if (SWZ) {
                    const int vox = vox0[i] + goff + t;
                    const int off = vox * RB + (((ks * 2 + hh) ^ (((lwv[i] + t) >> 1) & 7)) << 4);
                    ah[slot][i] = *reinterpret_cast<const half8*>(lds + off);
                    al[slot][i] = *reinterpret_cast<const half8*>(lds + (off ^ 64));
                } else {
                    constexpr int CH = (IW + 1) / 2;
                    const int tcol = (STRIDE == 2) ? (t & 1) * CH + (t >> 1) : t;      // record offset of tap kw = t
                    const unsigned char* p = lds + (vox0[i] + goff) * RB + 16 * hh + tcol * RB + ks * 32;
                    ah[slot][i] = *reinterpret_cast<const half8*>(p);
                    al[slot][i] = *reinterpret_cast<const half8*>(p + HB);
                }
            }
        };
        auto frag_b = [&](int s, int slot, const unsigned char* bb) {
            const int t = s / KS, ks = s % KS;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const unsigned char* p = bb + (((t * KS + ks) * NB + j) * 2) * 1024;
                bh_[slot][j] = *reinterpret_cast<const half8*>(p);
                bl[slot][j] = *reinterpret_cast<const half8*>(p + 1024);
            }
        };
        int goff[MB], goff_next[MB];
#pragma unroll
        for (int i = 0; i < MB; ++i) goff_next[i] = grp_off(0, i);
#pragma unroll
        for (int q = 0; q < PF; ++q) frag_a(q, q, goff_next);
        // one weight group; `drain(s)` runs behind the MFMAs of step s (s as an integral_constant)
#ifdef EXP_BGLOB
        const u32x4* const wcur = wbase_of(it);
        const u32x4* const wnxt = wbase_of(it + 1);
#endif
        // FIRST / LAST (compile-time: is this group 0 / group 8 of the item): the rolled loop over groups 1..7 then has NO
        // run-time condition around its fragment prefetches.  With `if (g < 8)` / `if (g == 0)` inside one rolled body the
        // compiler's waitcnt pass had to assume the path WITHOUT the next group's prefetch burst, so the last step of every group
        // waited with lgkmcnt(4) .. lgkmcnt(0) -- i.e. for the 12 reads just issued for the NEXT group -- and every group paid an
        // LDS round trip (~450 of ~1050 cycles per 18-MFMA group of the stride-2 kernel in the per-wave stamps).
        auto do_group = [&](int g, auto firstc, auto lastc, auto drain) {
            constexpr bool FIRST = decltype(firstc)::value, LAST = decltype(lastc)::value;
            const int g3 = g - 3 * ((g * 11) >> 5);     // g % 3 (g < 9)
            const unsigned char* bb = lds_b + (RESB ? g : (B3 ? g3 : ((gg0 + g) & 1))) * GB + lane * 16;
#ifdef EXP_B_EARLY
            constexpr bool BEARLY = true;               // diagnostic (wrong numerics): every kernel reads the next group's first B
            [[maybe_unused]] const unsigned char* bb_next =   // fragments before the barrier -- what a third buffer would buy in time
                lds_b + (B3 ? (g3 == 2 ? 0 : g3 + 1) : ((gg0 + g + 1) & 1)) * GB + lane * 16;
#else
            constexpr bool BEARLY = B3;
            [[maybe_unused]] const unsigned char* bb_next = lds_b + (g3 == 2 ? 0 : g3 + 1) * GB + lane * 16;   // B3: group g+1's buffer
#endif
#pragma unroll
            for (int i = 0; i < MB; ++i) { goff[i] = goff_next[i]; goff_next[i] = grp_off(g + 1, i); }   // (kd, kh) rows, in voxels
#ifdef EXP_BGLOB
            const u32x4* const bg = wcur + (size_t)g * PG;
            const u32x4* const bgn = g < 8 ? bg + PG : wnxt;
            (void)bb;
#else
            if (!BEARLY || FIRST) {                     // (B3: the previous group read these before its barrier)
#pragma unroll
                for (int q = 0; q < PF; ++q) frag_b(q, q, bb);
            }
#endif
            static_for<NS>([&](auto sc_) {
                constexpr int s = decltype(sc_)::value;
#ifndef EXP_NO_FRAG
#ifdef EXP_BGLOB
                if (s + PF < NS) frag_a(s + PF, (s + PF) % R, goff);
                else if (!LAST) frag_a(s + PF - NS, (s + PF) % R, goff_next);
                if (s + PFB < NS) frag_bg(s + PFB, (s + PFB) % BR, bg);
                else frag_bg(s + PFB - NS, (s + PFB) % BR, bgn);
#else
                if (s + PF < NS) { frag_a(s + PF, (s + PF) % R, goff); frag_b(s + PF, (s + PF) % R, bb); }
                else if constexpr (!LAST) {
                    frag_a(s + PF - NS, (s + PF) % R, goff_next);
                    if constexpr (BEARLY) frag_b(s + PF - NS, (s + PF) % R, bb_next);
                }
#endif
#endif
#ifdef EXP_NO_SGB
                __builtin_amdgcn_sched_barrier(0);
#endif
#ifndef EXP_NO_MFMA
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
#ifdef EXP_BGLOB
                        acc0[i][j] = mfma16(ah[s % R][i], bgh[s % BR][j], acc0[i][j]);
                        acc1[i][j] = mfma16(al[s % R][i], bgh[s % BR][j], acc1[i][j]);
                        acc1[i][j] = mfma16(ah[s % R][i], bgl[s % BR][j], acc1[i][j]);
#else
                        acc0[i][j] = mfma16(ah[s % R][i], bh_[s % R][j], acc0[i][j]);
                        acc1[i][j] = mfma16(al[s % R][i], bh_[s % R][j], acc1[i][j]);
                        acc1[i][j] = mfma16(ah[s % R][i], bl[s % R][j], acc1[i][j]);
#endif
                    }
#else
#pragma unroll
                for (int i = 0; i < MB; ++i) asm volatile("" ::"v"(ah[s % R][i]), "v"(al[s % R][i]));
#pragma unroll
                for (int j = 0; j < NB; ++j) asm volatile("" ::"v"(bh_[s % R][j]), "v"(bl[s % R][j]));
#endif
                drain(sc_);
#ifndef EXP_NO_SGB
                {   // Interleave: the wave is in-order, so everything placed after a step's last MFMA delays the next step's
                    // first one.  One LDS read and two VALU behind each MFMA instead (an MFMA leaves ~24 issue cycles free).
                    constexpr int NM_ = 3 * MB * NB, NRD_ = 2 * MB + 2 * NB;
#pragma unroll
                    for (int m = 0; m < NM_; ++m) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (m < NRD_) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                        if (m == NM_ - 2) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
                    }
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
            });
#ifndef EXP_NO_GROUP_BARRIER
            if constexpr (!RESB && !LAST) {
                // g_g.  The reads in flight here are fragment prefetches for group g+1: A fragments of tile planes that are still
                // live (the loaders overwrite a plane only after the barrier that ends its LAST group, and group g+1 never reads a
                // plane that dies at g_g), and -- three buffers -- B fragments of buffer g+1, which is next written two barriers
                // later.  The buffer the loaders refill after g_g (group g's) was consumed by this group's MFMAs.  So no drain.
                STAMP(wave, sidx, lane);
#ifdef EXP_FULL_GROUP_BARRIER
                MSNET_LDS_BARRIER();
#else
                MSNET_READER_BARRIER();
#endif
                STAMP(wave, sidx, lane);
            }
#endif
        };
        if constexpr (DRAIN) {
            static_for<9>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                do_group(g, std::integral_constant<bool, g == 0>{}, std::integral_constant<bool, g == 8>{},
                         [&](auto sc_) { drain_piece(std::integral_constant<int, g * NS + decltype(sc_)::value>{}); });
            });
            if (pend_live) flag_overflow(a.oflag, pamax);
            pamax = 0.f;
            pend_live = false;
#pragma unroll
            for (int i = 0; i < MB; ++i) plw[i] = 0;
        } else {
            using T_ = std::integral_constant<bool, true>;
            using F_ = std::integral_constant<bool, false>;
            do_group(0, T_{}, F_{}, [](auto) {});
#pragma unroll 1                                 // (expanding all nine groups was measured: Co=64 spills, stride 2 +-0)
            for (int g = 1; g < 8; ++g) do_group(g, F_{}, F_{}, [](auto) {});
            do_group(8, F_{}, T_{}, [](auto) {});
        }
        }
        if (chunk == nchunks - 1) { pending = true; pn = n; pod0 = od0; poh0 = oh0; pow0 = ow0; pcg = cg; }
    }
    if (pending) epilogue(pn, pod0, poh0, pow0, pcg);
}

// ---------------------------------------------------------------------------------------------
// Transposed conv (k3, s2, p1, op1) on the split-fp16 MFMA.  Same decomposition as deconv3d_k3s2_mfma (8 output-parity
// classes sharing one LDS tile of INPUT voxels, 27 (class, tap) pairs = the dense definition's MACs) and the same
// wave-specialised persistent scheme as the forward conv above.  Differences:
//   * the tile holds ALL input channels (CI = 16*KS, records of 4*CI bytes + 16 pad), so classes can be finished one
//     after another with a single accumulator pair; a tile is staged once and used by all 27 weight groups;
//   * a weight group is one (class, tap): KS K-steps x NB x (hi, lo) KiB pairs, double-buffered in LDS, streamed by the
//     loader waves three groups ahead; one barrier per group;
//   * after the last tap of a class the MFMA waves run that class's strided epilogue (+ residual, ReLU).
// Work item = (input tile 2x4x32, output-channel group of 32*NB).
// ---------------------------------------------------------------------------------------------
struct DTap { int pd, ph, pw, dd, dh, dw, kd, kh, kw, last; };
__host__ __device__ constexpr DTap dtap(int k) {
    // class order 7,6,5,3,4,2,1,0 (8,4,4,4,2,2,2,1 taps); taps of a class in (dd, dh, dw) order
    constexpr int order[8] = {7, 6, 5, 3, 4, 2, 1, 0};
    int base = 0;
    for (int c = 0; c < 8; ++c) {
        const int cls = order[c];
        const int pd = cls >> 2, ph = (cls >> 1) & 1, pw = cls & 1;
        const int nt = (pd + 1) * (ph + 1) * (pw + 1);
        if (k < base + nt) {
            const int tp = k - base;
            const int dw = tp % (pw + 1), dh = (tp / (pw + 1)) % (ph + 1), dd = tp / ((pw + 1) * (ph + 1));
            return DTap{pd, ph, pw, dd, dh, dw, pd ? (dd ? 0 : 2) : 1, ph ? (dh ? 0 : 2) : 1, pw ? (dw ? 0 : 2) : 1,
                        tp == nt - 1};
        }
        base += nt;
    }
    return DTap{0, 0, 0, 0, 0, 0, 1, 1, 1, 1};
}

// first group (in dtap order) of the class at position c of the class order
__host__ __device__ constexpr int class_first_group(int c) {
    int k = 0, pos = 0;
    while (pos < c) {
        if (dtap(k).last) ++pos;
        ++k;
    }
    return k;
}

// dtap(k) as one word per group for the kernel's run-time loop: dd | dh<<1 | dw<<2 | pd<<3 | ph<<4 | pw<<5 | first<<6 | last<<7
struct DeconvTapTable { int e[27]; };
constexpr DeconvTapTable make_deconv_taps() {
    DeconvTapTable t{};
    for (int k = 0; k < 27; ++k) {
        const DTap d = dtap(k);
        const bool first = (k == 0) || dtap(k > 0 ? k - 1 : 0).last;
        t.e[k] = d.dd | d.dh << 1 | d.dw << 2 | d.pd << 3 | d.ph << 4 | d.pw << 5 | (first ? 64 : 0) | (d.last ? 128 : 0);
    }
    return t;
}
__constant__ DeconvTapTable kDeconvTaps = make_deconv_taps();

// packed deconv weights (16-byte units): idx = ((((cg*27 + k)*KS + ks)*NB + nbl)*2 + hl)*64 + lane, k = group in dtap order,
// element j of lane (r, h): W[ci = ks*16 + h*8 + j][co = (cg*NB + nbl)*32 + r][tap = (kd*3+kh)*3+kw]   (ConvTranspose3d layout)
__global__ void pack_deconv_weight_f16s_kernel(const float* __restrict__ w, _Float16* __restrict__ out, int Ci, int Co,
                                               int KS, int NB) {
    const size_t total = (size_t)27 * Ci * Co * 2;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        size_t i = o;
        const int j = i & 7; i >>= 3;
        const int lane = i & 63; i >>= 6;
        const int hl = i & 1; i >>= 1;
        const int nbl = i % NB; i /= NB;
        const int ks = i % KS; i /= KS;
        const int k = i % 27;
        const int cg = (int)(i / 27);
        const DTap t = dtap(k);
        const int co = (cg * NB + nbl) * 32 + (lane & 31);
        const int ci = ks * 16 + (lane >> 5) * 8 + j;
        const float v = w[((size_t)ci * Co + co) * 27 + (t.kd * 3 + t.kh) * 3 + t.kw];
        const _Float16 h = (_Float16)v;
        out[o] = hl ? (_Float16)((v - (float)h) * kLoScale) : h;
    }
}

#ifndef DEC_SPREAD_MIN
#define DEC_SPREAD_MIN 2
#endif
#ifdef DEXP_NO_GBAR
#define MSNET_DBAR() do {} while (0)
#else
#define MSNET_DBAR() MSNET_LDS_BARRIER()
#endif
template <int KS, int NB, int LW = 4>
__global__ __launch_bounds__(256 + 64 * LW, (256 + 64 * LW) / 256) void deconv3d_k3s2_f16s_ws(ConvArgs a) {
    constexpr int LT = 64 * LW;                          // loader threads
    constexpr int TD = 2, TH = 4, TW = 32, MB = 2;
    constexpr int CI = 16 * KS;
    constexpr int ID = TD + 1, IH = TH + 1, IW = TW + 1;
    constexpr int HB = 2 * CI;                          // bytes of the hi (or lo) half of a voxel record
    constexpr int RB = 2 * HB + 16;                     // odd number of 16-byte slots => conflict-free 1x32 M-blocks
    constexpr int V = CI / 4;
    constexpr int NPOS = ID * IH * IW;
    constexpr int NSLOT = NPOS * V;
    constexpr int NL = (NSLOT + LT - 1) / LT;             // fp32 float4 per loader thread per tile
    constexpr int GB = KS * NB * 2 * 1024;              // bytes of one weight group (one tap)
    constexpr int PG = GB / 16;
    constexpr int NLB = PG / LT;                       // 16-byte pieces per loader thread per group
    static_assert(PG % LT == 0 && (NLB == 1 || NLB == 2 || NLB == 4), "weight group pieces per loader thread");
    static_assert(NPOS * RB + 2 * GB <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) unsigned char lds[NPOS * RB + 2 * GB];
    unsigned char* const lds_b = lds + NPOS * RB;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const unsigned G = gridDim.x;
    const unsigned lb = xcd_remap(blockIdx.x, G);
    const int ncg = a.ngroups;                          // output-channel groups of 32*NB
    const unsigned T = (unsigned)a.N * a.ntd * a.nth * a.ntw * ncg;
    const int nitems = (T > lb) ? (int)((T - lb + G - 1) / G) : 0;
    if (nitems == 0) return;
    const u32x4* wg = reinterpret_cast<const u32x4*>(a.wpk);

    auto decode = [&](int it, int& n, int& d0, int& h0, int& w0, int& cg) {
        unsigned t = lb + (unsigned)it * G;
        cg = t % ncg; t /= ncg;
        w0 = (t % a.ntw) * TW; t /= a.ntw;
        h0 = (t % a.nth) * TH; t /= a.nth;
        d0 = (t % a.ntd) * TD;
        n = t / a.ntd;
    };

    if (wave >= 4) {
        // ------------------------------ loader waves ------------------------------
        const int lt = tid - 256;
        struct BSet { u32x4 v0, v1, v2, v3; };
        BSet bw0, bw1, bw2;
        f32x4 av[NL];
        const size_t sample_bytes = (size_t)a.D * a.H * a.W * a.Ci * 4;
        // The next tile is requested a few loads per weight group (NL = 30 per thread: as one burst behind b2 the loader spent
        // several groups just issuing them, and its weight copies -- which the MFMA waves wait for at every group barrier -- queued
        // up behind).  `live` = false (past the last item) turns the requests into out-of-range offsets.
        int nx_d0 = 0, nx_h0 = 0, nx_w0 = 0;
        unsigned nx_base = 0;
        bool nx_live = false;
        __amdgpu_buffer_rsrc_t nx_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, 0, 0x00020000);
        auto prep_a = [&](int it, bool live) {
            int n, cg;
            decode(live ? it : 0, n, nx_d0, nx_h0, nx_w0, cg);
            nx_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x) + (size_t)n * (sample_bytes / 4), 0, (int)sample_bytes, 0x00020000);
            nx_base = (unsigned)((((long)nx_d0 * a.H + nx_h0) * a.W + nx_w0) * a.Ci) * 4u;
            nx_live = live;
        };
        auto issue_part = [&](int u0, int u1) {
            int ltv = lt;
            asm volatile("" : "+v"(ltv));               // keep the per-slot index math inside the loop (registers)
#pragma unroll
            for (int u = 0; u < NL; ++u) {
                if (u < u0 || u >= u1) continue;
                const int slot = u * LT + ltv;
                const int pos = slot / V, c4 = slot % V;
                const int iw = pos % IW, ih = (pos / IW) % IH, id = pos / (IW * IH);
                const bool ok = nx_live && slot < NSLOT && nx_d0 + id < a.D && nx_h0 + ih < a.H && nx_w0 + iw < a.W;
                const unsigned voff = ok ? nx_base + (unsigned)((((id * a.H + ih) * a.W + iw) * a.Ci + c4 * 4) * 4) : 0xffffffffu;
                av[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(nx_rsrc, voff, 0, 0));
            }
        };
        auto write_a = [&]() {
#pragma unroll
            for (int u = 0; u < NL; ++u) {
                const int slot = u * LT + lt;
                if (slot < NSLOT) {
                    half4 hi, lo;
                    split4(av[u], hi, lo);
                    unsigned char* rec = lds + (slot / V) * RB + (slot % V) * 8;
                    *reinterpret_cast<half4*>(rec) = hi;
                    *reinterpret_cast<half4*>(rec + HB) = lo;
                }
            }
        };
        // weight groups: endless stream k = it*27 + g; group k uses register set k % 3 (27 % 3 == 0) and LDS buffer k & 1
        const int ngroups_total = nitems * 27;
        int b_item = 0;                                 // item whose group is cg_cur
        int cg_cur = 0, cg_next = 0;                    // output-channel group of the current / next item
        auto cg_of = [&](int it) {
            int n, d0, h0, w0, cg;
            decode(it < nitems ? it : nitems - 1, n, d0, h0, w0, cg);
            return cg;
        };
        auto b_src = [&](int k) {                       // k - k0 is a compile-time constant at every call site
            const int k0_ = (k / 27) * 27;
            (void)k0_;
            k = k < ngroups_total ? k : ngroups_total - 1;
            const int gi = k % 27;
            const int cg = (k / 27 == b_item) ? cg_cur : cg_next;
            return wg + (size_t)(cg * 27 + gi) * PG + lt;
        };
#define MSNET_ISSUE_B(K, SET)                                                                     \
    do {                                                                                          \
        const u32x4* src_ = b_src(K);                                                             \
        SET.v0 = src_[0];                                                                         \
        if constexpr (NLB > 1) SET.v1 = src_[LT];                                                \
        if constexpr (NLB > 2) { SET.v2 = src_[2 * LT]; SET.v3 = src_[3 * LT]; }                        \
    } while (0)
#define MSNET_WRITE_B(K, SET)                                                                     \
    do {                                                                                          \
        u32x4* dst_ = reinterpret_cast<u32x4*>(lds_b + ((K) & 1) * GB) + lt;                      \
        dst_[0] = SET.v0;                                                                         \
        if constexpr (NLB > 1) dst_[LT] = SET.v1;                                                \
        if constexpr (NLB > 2) { dst_[2 * LT] = SET.v2; dst_[3 * LT] = SET.v3; }                        \
    } while (0)
#if defined(DEXP_NO_B)
#define MSNET_DGROUP(G, SET) MSNET_DBAR();
#else
// group G: copy group G+1's weights, request group G+4's, then this group's share of the next tile (APG loads), barrier
#define MSNET_DGROUP(G, SET)                                                        \
    MSNET_WRITE_B(k0 + (G) + 1, SET);                                               \
    MSNET_ISSUE_B(k0 + (G) + 1 + 3, SET);                                           \
    if constexpr ((G) * APG < NL) issue_part((G) * APG, (G) * APG + APG);           \
    MSNET_DBAR();
#endif
        constexpr int APG = (NL + 19) / 20;             // loads per group: the tile is complete after at most 20 of the 26 groups
        prep_a(0, true);
        issue_part(0, NL);
        cg_cur = cg_of(0); cg_next = cg_of(1);
        MSNET_ISSUE_B(0, bw0);
        MSNET_ISSUE_B(1, bw1);
        MSNET_ISSUE_B(2, bw2);
        for (int it = 0; it < nitems; ++it) {
            const int k0 = it * 27;
            if (it > 0) { b_item = it; cg_cur = cg_next; cg_next = cg_of(it + 1); }
            MSNET_LDS_BARRIER();                        // b1: MFMA waves are done with the previous tile
#ifndef DEXP_NO_A
            write_a();
#endif
            MSNET_WRITE_B(k0, bw0);
            MSNET_ISSUE_B(k0 + 3, bw0);
            MSNET_LDS_BARRIER();                        // b2: tile and group 0 are in LDS
            prep_a(it + 1, it + 1 < nitems);
            MSNET_DGROUP(0, bw1)  MSNET_DGROUP(1, bw2)  MSNET_DGROUP(2, bw0)  MSNET_DGROUP(3, bw1)  MSNET_DGROUP(4, bw2)
            MSNET_DGROUP(5, bw0)  MSNET_DGROUP(6, bw1)  MSNET_DGROUP(7, bw2)  MSNET_DGROUP(8, bw0)  MSNET_DGROUP(9, bw1)
            MSNET_DGROUP(10, bw2) MSNET_DGROUP(11, bw0) MSNET_DGROUP(12, bw1) MSNET_DGROUP(13, bw2) MSNET_DGROUP(14, bw0)
            MSNET_DGROUP(15, bw1) MSNET_DGROUP(16, bw2) MSNET_DGROUP(17, bw0) MSNET_DGROUP(18, bw1) MSNET_DGROUP(19, bw2)
            MSNET_DGROUP(20, bw0) MSNET_DGROUP(21, bw1) MSNET_DGROUP(22, bw2) MSNET_DGROUP(23, bw0) MSNET_DGROUP(24, bw1)
            MSNET_DGROUP(25, bw2)
        }
#undef MSNET_DGROUP
#undef MSNET_WRITE_B
#undef MSNET_ISSUE_B
        return;
    }

    // ------------------------------ MFMA waves ------------------------------
    const int wm = wave;
    const int r = lane & 31, hh = lane >> 5;
    [[maybe_unused]] int sidx_d = 0;
    int abase[MB];                                      // byte offset of this lane's input voxel record (+ lane-half slot)
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int mb = wm * MB + i;                     // M-block = (bd, bh) row of 32 input voxels
        const int bh = mb % TH, bd = mb / TH;
        abase[i] = ((bd * IH + bh) * IW + r) * RB + 16 * hh;
    }
    const int stride_w = 2 * a.Co * 4;                  // bytes between the output voxels of consecutive input voxels

    // The eight classes are expanded at compile time (their group loops stay rolled).  For the classes with two or more
    // taps the residual is requested piece by piece during their first two groups -- four dword loads per K-step instead
    // of a burst of 32 that blocks the wave for ~2300 cycles while the CU's memory pipe drains; the one-tap class keeps
    // the burst (DEC_SPREAD_MIN = 2 / 4 / 8 measured: 1.08 / 1.10 / 1.12 ms on deconvbn4).  Requesting a whole class ahead was built too: it needs
    // a second residual register set (spills) and, with 32 stores + 32 loads younger than the loads being waited for, runs
    // into the 6-bit vmcnt, i.e. ends up waiting for store acknowledgements -- slower than this.
    constexpr int PIECES = MB * NB * 16;
    f32x16 acc0[MB][NB], acc1[MB][NB], rres[1][MB][NB];
    unsigned obase[1][MB][NB];                          // byte offset of the lane's first output element
    int wlim[1];                                        // a.W - iwb (column validity)
    half8 ah[2][MB], al[2][MB], bh_[2][NB], bl[2][NB];
    const size_t osample = (size_t)a.OD * a.OH * a.OW * a.Co * 4;

    // output offsets of class (pd, ph, pw) of the tile at (d0, h0, w0), channel group nb0, into set `st`
    auto set_bases = [&](int st, int d0, int h0, int w0, int nb0, int pd, int ph, int pw) {
        wlim[st] = a.W - (w0 + 4 * hh);
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int mb = wm * MB + i;
                const int ihb = h0 + mb % TH, iwb = w0 + 4 * hh, id = d0 + mb / TH;
                obase[st][i][j] = (unsigned)(((((size_t)2 * id + pd) * a.OH + 2 * ihb + ph) * a.OW + 2 * iwb + pw) * a.Co +
                                             (nb0 + j) * 32 + r) * 4u;
                if (id >= a.D || ihb >= a.H) obase[st][i][j] = 0xffffffffu;     // whole M-block outside the input
            }
    };
    // request pieces [q0, q0 + cnt) of set `st` (piece = (i, j, e); element e is voxel column (e&3) + 8*(e>>2))
    auto request = [&](auto stc, auto q0c, auto cntc, __amdgpu_buffer_rsrc_t rs) {
        constexpr int st = decltype(stc)::value, q0 = decltype(q0c)::value, cnt = decltype(cntc)::value;
#pragma unroll
        for (int q = q0; q < q0 + cnt; ++q) {
            const int e = q % 16, j = (q / 16) % NB, i = q / (16 * NB);
            const int c = (e & 3) + 8 * (e >> 2);
            const unsigned ob = obase[st][i][j];
            const unsigned o = (ob != 0xffffffffu && c < wlim[st]) ? ob + (unsigned)(c * stride_w) : 0xffffffffu;
            rres[st][i][j][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, o, 0, 0));
        }
    };
    auto rs_res_of = [&](int n) { return make_rsrc(a.res ? a.res + (size_t)n * (osample / 4) : nullptr, a.res ? osample : 0); };

    for (int it = 0; it < nitems; ++it) {
        int n, d0, h0, w0, cg;
        decode(it, n, d0, h0, w0, cg);
        const int nb0 = cg * NB;
        float sc[NB], sh[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            sc[j] = a.scale ? a.scale[(nb0 + j) * 32 + r] : 1.f;
            sh[j] = a.shift ? a.shift[(nb0 + j) * 32 + r] : 0.f;
        }
        MSNET_LDS_BARRIER();                            // b1
        MSNET_LDS_BARRIER();                            // b2
#pragma unroll
        for (int j = 0; j < NB; ++j) asm volatile("" : "+v"(sc[j]), "+v"(sh[j]));     // land them before the residual stream starts
        const int gg0 = it * 27;
        const auto rs_y = make_rsrc(a.y + (size_t)n * (osample / 4), osample);
        const auto rs_res = rs_res_of(n);

        // one weight group: KS K-steps of MB x NB x 3 MFMAs; `after(ks)` runs behind the MFMAs of step ks
        auto group = [&](int k, int toff, auto after) {
            if (wave == 0) STAMP(0, sidx_d, lane);
            const unsigned char* bb = lds_b + ((gg0 + k) & 1) * GB + lane * 16;
            auto frag = [&](int ks, int slot) {
#pragma unroll
                for (int i = 0; i < MB; ++i) {
                    const unsigned char* p = lds + abase[i] + toff + ks * 32;
                    ah[slot][i] = *reinterpret_cast<const half8*>(p);
                    al[slot][i] = *reinterpret_cast<const half8*>(p + HB);
                }
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const unsigned char* p = bb + ((ks * NB + j) * 2) * 1024;
                    bh_[slot][j] = *reinterpret_cast<const half8*>(p);
                    bl[slot][j] = *reinterpret_cast<const half8*>(p + 1024);
                }
            };
            frag(0, 0);
            static_for<KS>([&](auto ksc) {
                constexpr int ks = decltype(ksc)::value;
                if (ks + 1 < KS) frag(ks + 1, (ks + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
#ifdef DEXP_NO_MFMA
                        acc0[i][j][0] += ah[ks & 1][i][0] * bh_[ks & 1][j][0];
                        acc1[i][j][0] += al[ks & 1][i][0] * bl[ks & 1][j][0];
#else
                        acc0[i][j] = mfma16(ah[ks & 1][i], bh_[ks & 1][j], acc0[i][j]);
                        acc1[i][j] = mfma16(al[ks & 1][i], bh_[ks & 1][j], acc1[i][j]);
                        acc1[i][j] = mfma16(ah[ks & 1][i], bl[ks & 1][j], acc1[i][j]);
#endif
                    }
                after(ksc);
                __builtin_amdgcn_sched_barrier(0);
            });
            if (k < 26) MSNET_DBAR();                   // g_k: this group's weights are consumed, the next are published
        };

        static_for<8>([&](auto cc) {
            constexpr int C = decltype(cc)::value;
            constexpr int K0 = class_first_group(C);
            constexpr DTap tc = dtap(K0);
            constexpr int NT = (tc.pd + 1) * (tc.ph + 1) * (tc.pw + 1);        // groups (taps) of this class
            constexpr bool SPREAD = NT >= DEC_SPREAD_MIN;                       // long classes: residual requested over two groups
            constexpr int P = SPREAD ? 2 : 0;
            constexpr int L = PIECES / (KS * 2);                                // requests per K-step when spread
            static_assert(PIECES % (KS * 2) == 0, "request schedule");
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) { acc0[i][j][e] = 0.f; acc1[i][j][e] = 0.f; }
            set_bases(0, d0, h0, w0, nb0, tc.pd, tc.ph, tc.pw);
            if (!SPREAD)
                request(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, PIECES>{}, rs_res);
            static_for<P>([&](auto pc) {
                constexpr int p = decltype(pc)::value;
                constexpr DTap t = dtap(K0 + p);
                group(K0 + p, ((t.dd * IH + t.dh) * IW + t.dw) * RB, [&](auto ksc) {
                    constexpr int ks = decltype(ksc)::value;
                    request(std::integral_constant<int, 0>{}, std::integral_constant<int, (p * KS + ks) * L>{},
                            std::integral_constant<int, L>{}, rs_res);
                });
            });
            for (int k = K0 + P; k < K0 + NT; ++k) {
                const int te = __builtin_amdgcn_readfirstlane(kDeconvTaps.e[k]);
                group(k, (((te & 1) * IH + ((te >> 1) & 1)) * IW + ((te >> 2) & 1)) * RB, [](auto) {});
            }
            // epilogue of class (pd, ph, pw): output voxels (2*id+pd, 2*ih+ph, 2*iw+pw)
            if (wave == 0) STAMP(0, sidx_d, lane);
#ifdef DEXP_NO_EPI
            if (acc0[0][0][0] == 123.456f)
#endif
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    f32x16 v;
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = acc0[i][j][e] + acc1[i][j][e] * kLoInv;
                    epilogue_store<32>(v, rres[0][i][j], sc[j], sh[j], rs_y, obase[0][i][j], 0, stride_w, a.relu,
                                       [&](int, int lw) { return obase[0][i][j] != 0xffffffffu && lw < wlim[0]; }, a.oflag);
                }
        });
    }
}

// ---------------------------------------------------------------------------------------------
// First layer (8 input channels, stride 1) on the split-fp16 MFMA.  With 8 channels a 16-wide K-step holds TWO taps: lane
// half hh of the A operand reads the voxel shifted by tap 2s+hh (8 channels = one 16-byte fragment), so the 27 taps
// take 14 K-steps instead of 27 half-empty ones.  LDS records are 32 bytes (hi | lo of the 8 channels, the two halves
// swapped on odd 8-voxel groups so 16 consecutive voxels cover all 64 banks), a 2x4x32 tile plus ALL weights is 54 KB:
// two workgroups per CU, no wave specialisation -- every wave loads, splits, multiplies and stores, and the other
// workgroup's MFMAs cover this one's staging and its 32 KB of output stores (this layer is HBM-store heavy).
//   packed weights (16-byte units): idx = ((s*NB + nb)*2 + hl)*64 + lane, element j of lane (r, hh):
//       W[co = nb*32 + r][ci = j][tap = 2s + hh]  (zero for tap 27);  hl = 0 hi, 1 lo.
// ---------------------------------------------------------------------------------------------
__global__ void pack_weight_c8_f16s_kernel(const float* __restrict__ w, _Float16* __restrict__ out, int Co) {
    const int NB = Co / 32;
    const size_t total = (size_t)14 * NB * 2 * 64 * 8;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        size_t i = o;
        const int j = i & 7; i >>= 3;
        const int lane = i & 63; i >>= 6;
        const int hl = i & 1; i >>= 1;
        const int nb = i % NB;
        const int sstep = (int)(i / NB);
        const int tap = 2 * sstep + (lane >> 5), co = nb * 32 + (lane & 31);
        const float v = tap < 27 ? w[((size_t)co * 8 + j) * 27 + tap] : 0.f;
        const _Float16 h = (_Float16)v;
        out[o] = hl ? (_Float16)((v - (float)h) * kLoScale) : h;
    }
}

// NCS = true: the input is the module's NCDHW volume [N][8][D][H][W] itself (the layout cbmv_generator.py:307-308 hands over):
// a slot's four channels come from four planes (buffer_load_dword per plane, 64 consecutive voxels of a tile row per wave
// instruction), so the separate NCDHW -> NDHWC pass over the 401 MB volume (0.14 ms, 802 MB of traffic) does not exist on
// this path.  The fp16-range check of the module input, which that pass carried, is made here on the staged values (bit 1 of
// the overflow word).  Slots: thread t holds voxels t, t+256, ... of the tile, both channel quads (NL = 2 * ceil(NPOS/256)).
// INCHK: the input IS the module input (NCS, or a channels-last volume handed to forward_ndhwc): check its fp16 range here.
template <int NB, bool NCS, bool INCHK = NCS>
__global__ __launch_bounds__(256, NB == 1 ? 2 : 1) void conv3d_c8_f16s_kernel(ConvArgs a) {    // (Co = 64: 128 accumulator registers, one workgroup per CU)
    constexpr int TD = 2, TH = 4, TW = 32, ID = TD + 2, IH = TH + 2, IW = TW + 2, NPOS = ID * IH * IW;
    constexpr int NSLOT = NCS ? ((NPOS + 255) / 256) * 512 : NPOS * 2;
    constexpr int NL = (NSLOT + 255) / 256;                             // float4 (channel quads) per thread per tile
    constexpr int WB = 14 * NB * 2 * 1024;
    __shared__ __attribute__((aligned(16))) unsigned char lds_a[NPOS * 32];
    __shared__ __attribute__((aligned(16))) unsigned char lds_b[WB];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const unsigned G = gridDim.x, lb = xcd_remap(blockIdx.x, G);
    const unsigned T = (unsigned)a.N * a.ntd * a.nth * a.ntw;
    const int nitems = (T > lb) ? (int)((T - lb + G - 1) / G) : 0;
    if (nitems == 0) return;
    {
        const u32x4* src = reinterpret_cast<const u32x4*>(a.wpk);
        u32x4* dst = reinterpret_cast<u32x4*>(lds_b);
        for (int k = tid; k < WB / 16; k += 256) dst[k] = src[k];
    }
    // loader role: slot = u*256 + tid -> (pos = slot >> 1, quad = slot & 1).  Per-slot constants (position in the tile, byte
    // offset from the tile origin) are computed once; an interior tile costs one add + one buffer load per slot.
    const size_t isample = (size_t)a.D * a.H * a.W * 8 * 4;
    const size_t iplane = (size_t)a.D * a.H * a.W * 4;  // NCS: bytes of one channel plane
    f32x4 av[NL];
    unsigned rel_[NL];                                   // byte offset of the slot from the tile's input origin (d0-1, h0-1, w0-1)
    int dhw_[NL];                                        // (id << 16) | (ih << 8) | iw, or -1 past the tile's end
    auto slot_pos = [&](int u) { return NCS ? (u >> 1) * 256 + tid : (u * 256 + tid) >> 1; };
    auto slot_q = [&](int u) { return NCS ? (u & 1) : ((u * 256 + tid) & 1); };
#pragma unroll
    for (int u = 0; u < NL; ++u) {
        const int pos = slot_pos(u), q = slot_q(u);
        const int iw = pos % IW, ih = (pos / IW) % IH, id = pos / (IW * IH);
        rel_[u] = NCS ? (unsigned)(((id * a.H + ih) * a.W + iw) * 4) : (unsigned)((((id * a.H + ih) * a.W + iw) * 8 + q * 4) * 4);
        dhw_[u] = pos < NPOS ? ((id << 16) | (ih << 8) | iw) : -1;
    }
    unsigned in_amax = 0u;                               // NCS: running max of the staged module input's magnitude BITS (NaN-aware)
    TileCtr ctr, nxt;                                    // current item / the one being fetched
    // (Tile order: w fastest, d slowest.  FETCH_SIZE reports 0.76-1.0 GB per launch for the 0.40 GB input: the two input planes
    // d-neighbours share come back over the fabric a thousand tiles later (Infinity Cache, not necessarily HBM).  Measured
    // alternatives, all slower: d as the fastest or second tile digit (0.95 GB fetched, +2-3 %); a sliding window along d as in
    // the 32->32 kernel (two new planes per tile, bit-identical results, +6 %) -- both scatter the 1.6 GB of stores, which adjacent
    // workgroups otherwise write as contiguous rows; the requests spread between the K-steps instead of one burst (+-0).
    // Ablations on the layer bench (0.89 ms): requests sent dead 0.59 ms, one store in sixteen 0.72 ms.)
    ctr.init(lb, G, 1, a.ntw, a.nth, a.ntd, 1);
    nxt = ctr;
    auto issue_a = [&](const TileCtr& c) {
        const int d0 = c.td * TD, h0 = c.th * TH, w0 = c.tw * TW;
        const unsigned base = (unsigned)(((((long)(d0 - 1) * a.H + (h0 - 1)) * a.W + (w0 - 1)) * (NCS ? 1 : 8)) * 4);   // may wrap; in-range slots bring it back
        const bool interior = d0 >= 1 && d0 - 1 + ID <= a.D && h0 >= 1 && h0 - 1 + IH <= a.H && w0 >= 1 && w0 - 1 + IW <= a.W;
        auto slot_ok = [&](int u) {
            bool ok = dhw_[u] >= 0;
            if (!interior) {
                const int gd = d0 - 1 + (dhw_[u] >> 16), gh = h0 - 1 + ((dhw_[u] >> 8) & 255), gw = w0 - 1 + (dhw_[u] & 255);
                ok = ok && (unsigned)gd < (unsigned)a.D && (unsigned)gh < (unsigned)a.H && (unsigned)gw < (unsigned)a.W;
            }
            return ok;
        };
        if constexpr (NCS) {
            // one descriptor per channel plane (the plane offset must not ride in soffset next to an out-of-range voffset)
            const float* xs = a.x + (size_t)c.n * (isample / 4);
#pragma unroll
            for (int u = 0; u < NL; ++u) {
                const unsigned off = slot_ok(u) ? base + rel_[u] : 0xffffffffu;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const auto rs = make_rsrc(xs + (size_t)((u & 1) * 4 + k) * (iplane / 4), iplane);
                    av[u][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
                }
            }
        } else {
            const auto rsrc = make_rsrc(a.x + (size_t)c.n * (isample / 4), isample);
#pragma unroll
            for (int u = 0; u < NL; ++u)
                av[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, slot_ok(u) ? base + rel_[u] : 0xffffffffu, 0, 0));
        }
    };
    auto write_a = [&]() {
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int pos = slot_pos(u), q = slot_q(u);
            if (pos < NPOS) {
                half4 hi, lo;
                if constexpr (INCHK) in_amax = max(max(in_amax, max(magnitude_bits(av[u][0]), magnitude_bits(av[u][1]))), max(magnitude_bits(av[u][2]), magnitude_bits(av[u][3])));
                split4(av[u], hi, lo);
                const int sw = ((pos >> 3) & 1) * 16;
                *reinterpret_cast<half4*>(lds_a + pos * 32 + sw + q * 8) = hi;
                *reinterpret_cast<half4*>(lds_a + pos * 32 + (sw ^ 16) + q * 8) = lo;
            }
        }
    };
    // MFMA role: wave owns M-blocks (bd = wave >> 1, bh = (wave & 1)*2 + i), i = 0..1
    int vox0[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) vox0[i] = ((wave >> 1) * IH + (wave & 1) * 2 + i) * IW + r;
    const size_t osample = (size_t)a.OD * a.OH * a.OW * a.Co * 4;

    issue_a(nxt);
    for (int it = 0; it < nitems; ++it) {
        const int n = ctr.n, d0 = ctr.td * TD, h0 = ctr.th * TH, w0 = ctr.tw * TW;
        ctr.next();
        nxt.next();
        __syncthreads();                                // previous tile fully consumed (and the weights are in LDS)
        write_a();
        __syncthreads();
        if (it + 1 < nitems) issue_a(nxt);              // in flight during the MFMAs and the epilogue
        f32x16 acc0[2][NB], acc1[2][NB];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) { acc0[i][j][e] = 0.f; acc1[i][j][e] = 0.f; }
        static_for<14>([&](auto sc_) {
            constexpr int sstep = decltype(sc_)::value;
            constexpr int t0 = 2 * sstep, t1 = 2 * sstep + 1 < 27 ? 2 * sstep + 1 : 26;
            constexpr int off0 = ((t0 / 9) * IH + (t0 / 3) % 3) * IW + t0 % 3;
            constexpr int off1 = ((t1 / 9) * IH + (t1 / 3) % 3) * IW + t1 % 3;
            half8 ah[2], al[2], bh_[NB], bl[NB];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int pos = vox0[i] + (hh ? off1 : off0);
                const int sw = ((pos >> 3) & 1) * 16;
                ah[i] = *reinterpret_cast<const half8*>(lds_a + pos * 32 + sw);
                al[i] = *reinterpret_cast<const half8*>(lds_a + pos * 32 + (sw ^ 16));
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                bh_[j] = *reinterpret_cast<const half8*>(lds_b + ((sstep * NB + j) * 2) * 1024 + lane * 16);
                bl[j] = *reinterpret_cast<const half8*>(lds_b + ((sstep * NB + j) * 2 + 1) * 1024 + lane * 16);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    acc0[i][j] = mfma16(ah[i], bh_[j], acc0[i][j]);
                    acc1[i][j] = mfma16(al[i], bh_[j], acc1[i][j]);
                    acc1[i][j] = mfma16(ah[i], bl[j], acc1[i][j]);
                }
        });
        // epilogue: lane = output channel, register e = voxel (e&3) + 8*(e>>2) + 4*hh of the 32-voxel row
        const auto rs_y = make_rsrc(a.y + (size_t)n * (osample / 4), osample);
        // (no load and no s_waitcnt between the stores when there is no residual -- the first layer's case; vmcnt counts stores too,
        // so a wait at an "optional residual" join would hold every block until the previous block's stores are acknowledged)
        auto block = [&](int b, unsigned& off, bool& rowok, int& wlim, float& sc, float& sh, f32x16& v) {
            const int i = b / NB, j = b % NB;
            const int od = d0 + (wave >> 1), oh = h0 + (wave & 1) * 2 + i, owb = w0 + 4 * hh;
            const int co = j * 32 + r;
            rowok = od < a.OD && oh < a.OH;
            wlim = a.OW - owb;
            sc = a.scale ? a.scale[co] : 1.f;
            sh = a.shift ? a.shift[co] : 0.f;
            off = (unsigned)((((size_t)od * a.OH + oh) * a.OW + owb) * a.Co + co) * 4u;
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = acc0[i][j][e] + acc1[i][j][e] * kLoInv;
        };
        if (!a.res) {
            f32x16 zero;
#pragma unroll
            for (int e = 0; e < 16; ++e) zero[e] = 0.f;
#pragma unroll
            for (int b = 0; b < 2 * NB; ++b) {
                unsigned off; bool rowok; int wlim; float sc, sh; f32x16 v;
                block(b, off, rowok, wlim, sc, sh, v);
#ifdef EXP_PRESPLIT
                epilogue_store_split<32>(v, zero, sc, sh, rs_y, sr_lane_offset(off, r), 0, a.Co * 4, a.relu, [&](int, int lw) { return rowok && lw < wlim; }, r, a.oflag);
#else
                epilogue_store<32>(v, zero, sc, sh, rs_y, off, 0, a.Co * 4, a.relu, [&](int, int lw) { return rowok && lw < wlim; }, a.oflag);
#endif
            }
        } else {
            const auto rs_res = make_rsrc(a.res + (size_t)n * (osample / 4), osample);
#pragma unroll
            for (int b = 0; b < 2 * NB; ++b) {
                unsigned off; bool rowok; int wlim; float sc, sh; f32x16 v, rv;
                block(b, off, rowok, wlim, sc, sh, v);
                residual_prefetch<32>(rv, rs_res, off, 0, a.Co * 4, [&](int, int lw) { return rowok && lw < wlim; });
                epilogue_store<32>(v, rv, sc, sh, rs_y, off, 0, a.Co * 4, a.relu, [&](int, int lw) { return rowok && lw < wlim; }, a.oflag);
            }
        }
    }
    if constexpr (INCHK) {
        // magnitude bits: out-of-range values, inf and NaN all compare >= the limit's bits
        if (a.oflag && in_amax >= kSplitMaxBits) atomicOr(a.oflag, 2u);
    }
}

template <int NB, bool NCS = false, bool INCHK = NCS>
static int launch_c8_f16s(const char* name, ConvArgs a, hipStream_t s) {
    a.ntd = cdiv(a.OD, 2); a.nth = cdiv(a.OH, 4); a.ntw = cdiv(a.OW, 32);
    const size_t ntiles = (size_t)a.N * a.ntd * a.nth * a.ntw;
    if (ntiles == 0 || ntiles > 0x7fffffffu) return fail("%s: bad tile count %zu", name, ntiles);
    if ((size_t)a.OD * a.OH * a.OW * a.Co * 4 > 0xfffffff0u || (size_t)a.D * a.H * a.W * 32 > 0xfffffff0u)
        return fail("%s: a sample exceeds the 4 GB buffer-descriptor range", name);
    const size_t cap = 2 * (size_t)num_cus();
    const size_t nblk = ntiles < cap ? ntiles : cap;
    const double vox = (double)a.N * a.OD * a.OH * a.OW;
    LaunchScope ls(name, s, 2.0 * 27.0 * a.Ci * a.Co * vox,
                   4.0 * ((double)a.N * a.D * a.H * a.W * a.Ci + vox * a.Co * (a.res ? 2 : 1)));
    hipLaunchKernelGGL((conv3d_c8_f16s_kernel<NB, NCS, INCHK>), dim3((unsigned)nblk), dim3(256), 0, s, a);
    return check_launch(name);
}

template <int KS, int NB>
static int launch_deconv_f16s(const char* name, ConvArgs a, hipStream_t s) {
    a.ntd = cdiv(a.D, 2); a.nth = cdiv(a.H, 4); a.ntw = cdiv(a.W, 32);
    a.ngroups = a.Co / (32 * NB);
    a.nbtot = a.Co / 32;
    const size_t nitems = (size_t)a.N * a.ntd * a.nth * a.ntw * a.ngroups;
    if (nitems == 0 || nitems > 0x7fffffffu) return fail("%s: bad item count %zu", name, nitems);
    if ((size_t)a.OD * a.OH * a.OW * a.Co * 4 > 0xfffffff0u || (size_t)a.D * a.H * a.W * a.Ci * 4 > 0x7ffffff0u)
        return fail("%s: a sample exceeds the buffer-descriptor range of this kernel (use the fp32 path)", name);
    const size_t nblk = nitems < (size_t)num_cus() ? nitems : (size_t)num_cus();
    const double ivox = (double)a.N * a.D * a.H * a.W;
    LaunchScope ls(name, s, 2.0 * 27.0 * a.Ci * a.Co * ivox, 4.0 * (ivox * a.Ci + 8.0 * ivox * a.Co * (a.res ? 2 : 1)));
    hipLaunchKernelGGL((deconv3d_k3s2_f16s_ws<KS, NB, DEC_LOADER_WAVES>), dim3((unsigned)nblk), dim3(256 + 64 * DEC_LOADER_WAVES), 0, s, a);
    return check_launch(name);
}

// ---------------------------------------------------------------------------------------------
// Direct kernel for SMALL layers (6x17x30 ... 12x34x60 grids): the tiled persistent kernels above give such a layer a
// few dozen work items, each walking all its weight groups alone (75 us for 128->128 on 3060 voxels, 144 us for the
// fp32 transposed conv).  Here a workgroup is ONE 32-voxel x 32-channel output block; its four waves split the taps,
// read operand A straight from global memory (fp32, split in registers; out-of-range taps are buffer loads at offset
// 0xffffffff = zeros), read operand B from the same packed weight images the tiled kernels use, and add their partial
// sums through LDS.  M-blocks are 32 consecutive voxels of the flattened output, so there are no edge tiles, and the
// epilogue needs no coordinates at all (element offset = voxel*Co + channel; the buffer bound drops the tail).
// TRANSPOSED: out[o] += in[i] w[k] for o = 2i - 1 + k, i.e. tap k reads i = (o + 1 - k)/2 where that is an integer; taps
// no lane of the block can use are skipped (for a block inside one output row that is 3/4 of them).
// ---------------------------------------------------------------------------------------------
template <bool TR, int KK>
__global__ __launch_bounds__(256) void conv3d_direct_f16s_kernel(ConvArgs a, int stride, int KS, int NBG) {
    __shared__ float red[3][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int nblk = a.Co / 32;
    // M-blocks never straddle two samples (blocks per sample = ceil(voxels / 32)), so a sample's result does not depend on
    // its position in the batch: the same taps go to the same waves and the partial sums meet in the same order.
    const long svox = (long)a.OD * a.OH * a.OW;           // output voxels per sample
    const long bps = (svox + 31) / 32;
    const long mblk_all = blockIdx.x / nblk;
    const int nb = blockIdx.x % nblk;
    const int n = (int)(mblk_all / bps);
    const long mblk = mblk_all % bps;
    long v = mblk * 32 + r;
    const bool vok = v < svox;
    if (!vok) v = svox - 1;
    const int ow = (int)(v % a.OW), oh = (int)((v / a.OW) % a.OH), od = (int)(v / ((long)a.OW * a.OH));
    const size_t ibytes = (size_t)a.D * a.H * a.W * a.Ci * 4;                   // descriptors cover ONE sample
    const auto rs_x = make_rsrc(a.x + (size_t)n * (ibytes / 4), ibytes);
    const u32x4* wq = reinterpret_cast<const u32x4*>(a.wpk);
    const int nchunks = KK / KS;

    f32x16 acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
    int nvalid = 0;
    for (int k = 0; k < 27; ++k) {
        int kd, kh, kw;
        if (TR) {
            const int te = __builtin_amdgcn_readfirstlane(kDeconvTaps.e[k]);
            // (pd, dd) -> kd: class parity pd = 0 uses kd = 1; pd = 1 uses kd = 0 (dd = 1) or 2 (dd = 0)
            const int dd = te & 1, dh = (te >> 1) & 1, dw = (te >> 2) & 1, pd = (te >> 3) & 1, ph = (te >> 4) & 1, pw = (te >> 5) & 1;
            kd = pd ? (dd ? 0 : 2) : 1; kh = ph ? (dh ? 0 : 2) : 1; kw = pw ? (dw ? 0 : 2) : 1;
        } else {
            kd = k / 9; kh = (k / 3) % 3; kw = k % 3;
        }
        int id, ih, iw;
        bool ok = vok;
        if (TR) {
            const int td = od + 1 - kd, th = oh + 1 - kh, tw = ow + 1 - kw;
            ok = ok && !((td | th | tw) & 1) && td >= 0 && th >= 0 && tw >= 0;
            id = td >> 1; ih = th >> 1; iw = tw >> 1;
        } else {
            id = od * stride - 1 + kd; ih = oh * stride - 1 + kh; iw = ow * stride - 1 + kw;
        }
        ok = ok && (unsigned)id < (unsigned)a.D && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
        if (__builtin_amdgcn_ballot_w64(ok) == 0) continue;
        if ((nvalid++ & 3) != wave) continue;           // the block's usable taps are dealt round-robin to its four waves
        const unsigned voff = ok ? (unsigned)((((size_t)id * a.H + ih) * a.W + iw) * a.Ci) * 4u + 32u * hh : 0xffffffffu;
        f32x4 x0[KK], x1[KK];
        u32x4 wb[KK][2];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const unsigned o = ok ? voff + 64u * kk : voff;
            x0[kk] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, o, 0, 0));
            x1[kk] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? o + 16u : o, 0, 0));
            size_t u;
            if (TR) u = ((((size_t)nb * 27 + k) * KK + kk) * 2) * 64 + lane;
            else    u = (((((((size_t)(nb / NBG) * nchunks + kk / KS) * 9 + k / 3) * 3 + k % 3) * KS + kk % KS) * NBG + nb % NBG) * 2) * 64 + lane;
            wb[kk][0] = wq[u];
            wb[kk][1] = wq[u + 64];
        }
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            half4 h0, l0, h1, l1;
            split4_cxx(x0[kk], h0, l0);              // operands of the MFMAs right below: compiler-scheduled form (hazards)
            split4_cxx(x1[kk], h1, l1);
            const half8 ah = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
            const half8 al = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
            const half8 bh = __builtin_bit_cast(half8, wb[kk][0]), bl = __builtin_bit_cast(half8, wb[kk][1]);
            acc0 = mfma16(ah, bh, acc0);
            acc1 = mfma16(al, bh, acc1);
            acc1 = mfma16(ah, bl, acc1);
        }
    }
    f32x16 sum;
#pragma unroll
    for (int e = 0; e < 16; ++e) sum[e] = acc0[e] + acc1[e] * kLoInv;
    if (wave > 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) red[wave - 1][e][lane] = sum[e];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int e = 0; e < 16; ++e) sum[e] += red[0][e][lane] + red[1][e][lane] + red[2][e][lane];
    // epilogue: lane = channel nb*32 + r, register e = voxel mblk*32 + (e&3) + 8*(e>>2) + 4*hh of the flattened output
    const size_t obytes = (size_t)svox * a.Co * 4;         // one sample; the buffer bound drops the tail of its last block
    const auto rs_y = make_rsrc(a.y + (size_t)n * (obytes / 4), obytes);
    const auto rs_res = make_rsrc(a.res ? a.res + (size_t)n * (obytes / 4) : nullptr, a.res ? obytes : 0);
    const int co = nb * 32 + r;
    const float sc = a.scale ? a.scale[co] : 1.f;
    const float sh = a.shift ? a.shift[co] : 0.f;
    const unsigned off = (unsigned)(((size_t)mblk * 32 + 4 * hh) * a.Co + co) * 4u;
    f32x16 rv;
    residual_prefetch<32>(rv, rs_res, off, 0, a.Co * 4, [](int, int) { return true; });
    epilogue_store<32>(sum, rv, sc, sh, rs_y, off, 0, a.Co * 4, a.relu, [](int, int) { return true; }, a.oflag);
}

// true if the direct kernel ran (small layer), false if the caller should use a tiled kernel, negative never
template <bool TR>
static int launch_direct_f16s(const char* name, ConvArgs a, int stride, int KS, int NBG, hipStream_t s) {
    const size_t svox = (size_t)a.OD * a.OH * a.OW;
    const size_t total = (size_t)a.N * svox;
    const size_t nblocks = (size_t)a.N * ((svox + 31) / 32) * (a.Co / 32);
    if (nblocks > 0x7fffffffu) return fail("%s: too many blocks", name);
    const int KK = a.Ci / 16;
    const double vox = TR ? (double)a.N * a.D * a.H * a.W : (double)total;
    LaunchScope ls(name, s, 2.0 * 27.0 * a.Ci * a.Co * vox,
                   4.0 * ((double)a.N * a.D * a.H * a.W * a.Ci + (double)total * a.Co * (a.res ? 2 : 1)));
    const dim3 g((unsigned)nblocks), b(256);
    if (KK == 2)      hipLaunchKernelGGL((conv3d_direct_f16s_kernel<TR, 2>), g, b, 0, s, a, stride, KS, NBG);
    else if (KK == 4) hipLaunchKernelGGL((conv3d_direct_f16s_kernel<TR, 4>), g, b, 0, s, a, stride, KS, NBG);
    else              hipLaunchKernelGGL((conv3d_direct_f16s_kernel<TR, 8>), g, b, 0, s, a, stride, KS, NBG);
    return check_launch(name);
}

// shapes the direct kernel can run at all
static bool direct_shape_ok(const ConvArgs& a) {
    const int KK = a.Ci / 16;
    if (a.Ci % 16 || !(KK == 2 || KK == 4 || KK == 8) || a.Co % 32) return false;
    return (size_t)a.D * a.H * a.W * a.Ci * 4 <= 0xfffffff0u && (size_t)a.OD * a.OH * a.OW * a.Co * 4 <= 0xfffffff0u;     // per sample
}
// a layer is "small" when the tiled kernel would have work for fewer than a quarter of the CUs (measured: at 108 tiles the
// tiled kernel still wins, 49 vs 61 us; at 30-54 tiles the direct one does, 37 vs 75 and 23 vs 40 us)
static bool direct_eligible(const ConvArgs& a, size_t tiled_items) {
    if (!direct_shape_ok(a)) return false;
    if (const char* e = getenv("MSNET_DIRECT")) {       // test hook: "0" never, "1" whenever the shape allows
        if (e[0] == '0') return false;
        if (e[0] == '1') return true;
    }
    return tiled_items < (size_t)num_cus() / 4;
}

template <int TD, int TH, int TW, int BW, int MB, int NB, bool SWZ, int KS, bool RESB, int STRIDE = 1, int LW = 4>
static int launch_f16s(const char* name, ConvArgs a, hipStream_t s) {
    a.ntd = cdiv(a.OD, TD); a.nth = cdiv(a.OH, TH); a.ntw = cdiv(a.OW, TW);
    a.ngroups = a.Co / (32 * NB); a.nbtot = a.Co / 32;
    const size_t ntiles = (size_t)a.N * a.ntd * a.nth * a.ntw * a.ngroups;
    if (ntiles == 0 || ntiles > 0x7fffffffu) return fail("%s: bad tile count %zu", name, ntiles);
    if ((size_t)a.D * a.H * a.W * a.Ci * 4 > 0x7ffffff0u || (size_t)a.OD * a.OH * a.OW * a.Co * 4 > 0xfffffff0u)
        return fail("%s: a sample exceeds the range of the kernel's buffer descriptors (2 GB in, 4 GB out; use the fp32 path)", name);
    const size_t nblk = ntiles < (size_t)num_cus() ? ntiles : (size_t)num_cus();
    const double vox = (double)a.N * a.OD * a.OH * a.OW;
#ifdef EXP_STAGGER
    if (const char* e = getenv("MSNET_EXP_STAGGER")) a.stagger = atoi(e);
#endif
    LaunchScope ls(name, s, 2.0 * 27.0 * a.Ci * a.Co * vox,
                   4.0 * ((double)a.N * a.D * a.H * a.W * a.Ci + vox * a.Co * (a.res ? 2 : 1)));
    hipLaunchKernelGGL((conv3d_k3s1_f16s_ws<TD, TH, TW, BW, MB, NB, SWZ, KS, RESB, STRIDE, false, LW>), dim3((unsigned)nblk), dim3(256 + 64 * LW), 0, s, a);
    return check_launch(name);
}

// Sliding-window launch for single-chunk stride-1 layers.  A tile column (all d at one (h, w) tile) is cut into `nseg`
// segments that are dealt to the persistent workgroups; within a segment every tile after the first stages two planes
// instead of four (measured: 0.85 of a tile's time), so longer segments are cheaper per tile but balance worse.  Returns -1
// when plain tiles are estimated to be no slower (the caller then launches the ordinary kernel).
template <int TH, int TW, int MB, int NB>
static int launch_f16s_slide(const char* name, ConvArgs a, hipStream_t s) {
    constexpr int TD = 2;
    a.ntd = cdiv(a.OD, TD); a.nth = cdiv(a.OH, TH); a.ntw = cdiv(a.OW, TW);
    a.ngroups = a.Co / (32 * NB); a.nbtot = a.Co / 32;
    const size_t cols = (size_t)a.N * a.nth * a.ntw * a.ngroups;
    if (cols == 0 || cols * a.ntd > 0x7fffffffu) return fail("%s: bad tile count", name);
    if ((size_t)a.OD * a.OH * a.OW * a.Co * 4 > 0xfffffff0u || (size_t)a.D * a.H * a.W * a.Ci * 4 > 0x7ffffff0u)
        return -1;                                      // 32-bit offsets inside a sample (drained stores, loader descriptor)
    const double G = (double)num_cus();
    const double plain = ceil((double)cols * a.ntd / G);    // plain tiles, one unit of time each
    double best = plain;
    int best_seg = 0;
    for (int seg = 1; seg <= a.ntd; ++seg) {            // cheapest segmentation; it must beat plain tiles by 3 % to be used
        if (a.ntd % seg) continue;
        const int len = a.ntd / seg;
        if (len < 2) break;
        const double cost = ceil((double)cols * seg / G) * (1.0 + 0.85 * (len - 1));    // (0.85: measured on 48x136x240 and 96x272x480)
        if (cost < best && cost < 0.97 * plain) { best = cost; best_seg = seg; }
    }
    if (const char* e = getenv("MSNET_FORCE_SLIDE_SEG")) {     // test hook: force the sliding kernel with this many segments
        const int seg = atoi(e);
        if (seg >= 1 && a.ntd % seg == 0 && a.ntd / seg >= 2) best_seg = seg;
        else if (seg == 0) best_seg = 0;
    }
    if (!best_seg) return -1;
    a.nseg = best_seg; a.seglen = a.ntd / best_seg;
    const size_t units = cols * a.nseg;
    const size_t nblk = units < (size_t)num_cus() ? units : (size_t)num_cus();
    const double vox = (double)a.N * a.OD * a.OH * a.OW;
    LaunchScope ls(name, s, 2.0 * 27.0 * a.Ci * a.Co * vox,
                   4.0 * ((double)a.N * a.D * a.H * a.W * a.Ci + vox * a.Co * (a.res ? 2 : 1)));
    hipLaunchKernelGGL((conv3d_k3s1_f16s_ws<TD, TH, TW, 32, MB, NB, false, 2, false, 1, true, SLIDE_LOADER_WAVES>), dim3((unsigned)nblk), dim3(256 + 64 * SLIDE_LOADER_WAVES), 0, s, a);
    return check_launch(name);
}


// ---------------------------------------------------------------------------------------------
// Winograd F(2,3) along DEPTH for the 32 -> 32 stride-1 layers (conv3dbn_2; dres0/dres1 and the hourglass 32->32 layers of the
// PSMNet aggregator).  Two output planes o0, o1 of a tile need the four input planes p0..p3:
//     q0 = p0 - p2, q1 = p1 + p2, q2 = p2 - p1, q3 = p1 - p3                      (loader waves, fp32, before the hi/lo split)
//     m_k = conv2d_3x3(q_k, g_k),  g0 = w[kd=0], g1 = (w0+w1+w2)/2, g2 = (w0-w1+w2)/2, g3 = w[kd=2]   (host, fp64, then split)
//     o0 = m0 + m1 + m2,  o1 = m1 - m2 - m3                                          (epilogue)
// i.e. 36 (plane, tap) products per output pair instead of 54: two thirds of the MFMAs of the direct form.  Under the conv
// kernels the package sits at its power limit and the MFMA stream is ~70 % of a launch's energy, which is what
// this buys back.  F(2,3) is well conditioned: with 22-bit split operands a layer is as close to the fp64 conv as the direct
// split-fp16 form (3.5e-7 vs 3.3e-7 relative, CPU emulation; tests/test_gpu_aggregators.py::test_conv3d_layer_winograd_depth).
//
// Shape: tile 2 x 4 x 32 output voxels, MFMA wave w owns output row w of BOTH planes through four accumulator pairs (m0..m3:
// 128 registers); the four q planes live in LDS as 128-byte swizzled records, interleaved by row ([ih][k][iw]: every fragment
// address is then one of twelve per-lane bases plus a compile-time immediate below 64 KB -- 6 x 34 voxels x 4 planes, 104 KB) and die one
// after another (q_k after the group that holds tap 9k+8), so the loaders write the next tile's q_k into slot k under the
// remaining groups and only q3 waits for the b1/b2 window.  A workgroup walks tile columns along d as the sliding-window kernel
// does: the raw planes p2, p3 of one step are p0, p1 of the next and stay in the loaders' REGISTERS, only two planes are
// fetched per tile.  Weights: 36 taps in six groups of six (24 KB), double-buffered in LDS.
// ---------------------------------------------------------------------------------------------
__global__ void pack_weight_wd_f16s_kernel(const float* __restrict__ g36, _Float16* __restrict__ out) {
    // g36: f32 [Co = 32][Ci = 32][36] (tap T = k*9 + kh*3 + kw) -> idx = ((((T*2 + ks)*2 + hl)*64 + lane)*8 + j)
    const size_t total = (size_t)36 * 2 * 2 * 64 * 8;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        size_t i = o;
        const int j = i & 7; i >>= 3;
        const int lane = i & 63; i >>= 6;
        const int hl = i & 1; i >>= 1;
        const int ks = i & 1; i >>= 1;
        const int T = (int)i;
        const int co = lane & 31, ci = ks * 16 + (lane >> 5) * 8 + j;
        const float v = g36[((size_t)co * 32 + ci) * 36 + T];
        const _Float16 h = (_Float16)v;
        out[o] = hl ? (_Float16)((v - (float)h) * kLoScale) : h;
    }
}

__global__ __launch_bounds__(512, 2) void conv3d_wd_f16s_kernel(ConvArgs a) {
    constexpr int TD = 2, TH = 4, TW = 32, IH = TH + 2, IW = TW + 2, NPV = IH * IW;
    constexpr int RB = 128, PLANE = NPV * RB;                   // 26,112 bytes per q plane
    constexpr int GB = 6 * 2 * 2 * 1024, PG = GB / 16;          // one weight group: six taps, 24 KB, 1536 pieces
    constexpr int LT = 256, PSLOT = NPV * 8, PL = (PSLOT + LT - 1) / LT;
    static_assert(PG == 6 * LT, "six weight pieces per loader thread and group");
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * PLANE + 2 * GB];
    unsigned char* const lds_b = lds + 4 * PLANE;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const unsigned G = gridDim.x;
    const unsigned lb = xcd_remap(blockIdx.x, G);
    const unsigned T = (unsigned)a.N * a.nseg * a.nth * a.ntw;          // column segments
    const int my_units = (T > lb) ? (int)((T - lb + G - 1) / G) : 0;
    const int nitems = my_units * a.seglen;
    if (nitems == 0) return;
    [[maybe_unused]] const u32x4* wg = reinterpret_cast<const u32x4*>(a.wpk);     // (-DEXP_WD_GLOBAL_B only)
    TileCtr ctr0;
    ctr0.init(lb, G, 1, a.ntw, a.nth, a.nseg, a.seglen);
    struct Coord { int n, od0, oh0, ow0; };
    auto coord_of = [&](const TileCtr& c) { return Coord{c.n, (c.td * a.seglen + c.pos) * TD, c.th * TH, c.tw * TW}; };

    if (wave >= 4) {
        // ------------------------------ loader waves ------------------------------
        const int lt = tid - 256;
        unsigned goff_[PL];                              // byte offset of slot u from the tile's input origin (plane-relative)
        int loff_[PL];                                   // LDS offset of slot u inside a plane slot (hi half; lo at ^ 64)
        unsigned mask0 = 0;
#pragma unroll
        for (int u = 0; u < PL; ++u) {
            const int sl = u * LT + lt, pos = sl >> 3, c4 = sl & 7;
            const int ih = pos / IW, iw = pos % IW;
            goff_[u] = (unsigned)(((ih * a.W + iw) * 32 + c4 * 4) * 4);
            loff_[u] = (ih * 4 * IW + iw) * RB + (c4 & 1) * 8 + (((c4 >> 1) ^ ((iw >> 1) & 7)) << 4);   // (+ k * IW * RB: plane k)
            mask0 |= (sl < PSLOT ? 1u : 0u) << u;
        }
        const size_t sample_bytes = (size_t)a.D * a.H * a.W * 32 * 4;
        // Raw planes of the tile whose q planes are being built: p0, p1 in one register set, p2, p3 in the other.  Inside a column
        // the next tile's p0, p1 ARE this tile's p2, p3, so the two sets swap roles from tile to tile (the tile body exists once per
        // parity: no register copies) and only two planes are fetched per tile.
        f32x4 S[2][2][PL];
        // request input plane (od0 - 1 + pl) of the tile at c; out-of-range slots are out-of-range offsets (zeros come back)
        auto issue = [&](f32x4 (&dst)[PL], const Coord& c, int pl, bool live) {
            const int gd = c.od0 - 1 + pl, ih0 = c.oh0 - 1, iw0 = c.ow0 - 1;
            const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x) + (size_t)c.n * (sample_bytes / 4), 0,
                                                                (int)sample_bytes, 0x00020000);
            const unsigned base = (unsigned)((((long)gd * a.H + ih0) * a.W + iw0) * 32) * 4u;
            const bool interior = ih0 >= 0 && ih0 + IH <= a.H && iw0 >= 0 && iw0 + IW <= a.W;
            unsigned mask = mask0;
            if (!interior) {
                mask = 0;
                int ltv = lt;
                asm volatile("" : "+v"(ltv));           // (edge tiles only: keeps the 2 x PL row / column values out of the tile loop's registers)
#pragma unroll
                for (int u = 0; u < PL; ++u) {
                    const int pos = (u * LT + ltv) >> 3;
                    const int gh = ih0 + pos / IW, gw = iw0 + pos % IW;
                    mask |= (((unsigned)gh < (unsigned)a.H && (unsigned)gw < (unsigned)a.W) ? 1u : 0u) << u;
                }
                mask &= mask0;
            }
            mask = (live && (unsigned)gd < (unsigned)a.D) ? mask : 0u;
#pragma unroll
            for (int u = 0; u < PL; ++u) {
                const unsigned voff = ((mask >> u) & 1u) ? base + goff_[u] : 0xffffffffu;
                dst[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
            }
        };
        // q_k of the tile whose raw planes are (PA = p0, p1; PB = p2, p3) -> LDS plane k (split + two 8-byte stores per slot)
        auto write_q = [&](auto kc, const f32x4 (&PA)[2][PL], const f32x4 (&PB)[2][PL]) {
            constexpr int k = decltype(kc)::value;
#pragma unroll
            for (int u = 0; u < PL; ++u) {
                if (u * LT + lt < PSLOT) {
                    const f32x4 q = k == 0 ? PA[0][u] - PB[0][u] : k == 1 ? PA[1][u] + PB[0][u] : k == 2 ? PB[0][u] - PA[1][u] : PA[1][u] - PB[1][u];
                    half4 hi, lo;
                    split4(q, hi, lo);
                    // (opaque copies: hipcc otherwise hoists all 2 x 4 x PL store addresses out of the tile loop and spills the
                    // raw planes to make room -- their reloads wait with vmcnt(0) for the next tile's HBM requests)
                    int off = loff_[u];
                    asm volatile("" : "+v"(off));
                    int off_lo = off ^ 64;              // k * IW * RB is a multiple of 128: (off + imm) ^ 64 == (off ^ 64) + imm
                    *reinterpret_cast<half4*>(lds + off + k * (IW * RB)) = hi;
                    *reinterpret_cast<half4*>(lds + off_lo + k * (IW * RB)) = lo;
                }
            }
        };
        struct BSet { u32x4 v0, v1, v2, v3, v4, v5; };
        // one set: group g+1 is copied to LDS in slot g and group g+2 requested right behind it (weights are L2-resident).  A second set
        // (two slots of flight) is no faster: 1.577 vs 1.574-1.589 ms in the network (and with 64-bit-address loads it spilled: 1.90 ms).
        BSet bw[1];
#ifdef EXP_WD_GLOBAL_B
#define WD_ISSUE_B(GRP, SET)                                                                                        \
    do { const u32x4* src_ = wg + (size_t)((GRP) % 6) * PG + lt;                                                    \
         SET.v0 = src_[0]; SET.v1 = src_[LT]; SET.v2 = src_[2 * LT]; SET.v3 = src_[3 * LT]; SET.v4 = src_[4 * LT]; SET.v5 = src_[5 * LT]; } while (0)
#else
        // weight pieces through a buffer descriptor: one per-thread byte offset (lt * 16) + a compile-time scalar offset per piece,
        // instead of 64-bit addresses in VGPRs (36 of them, which hipcc hoists out of the tile loop and -- once anything else
        // needs the registers -- spills; a spill reload waits vmcnt(0), i.e. for every tile request in flight)
        const auto rs_w = make_rsrc(a.wpk, (size_t)6 * GB);
        const unsigned lt16 = (unsigned)lt * 16u;
#define WD_LOAD_B(GRP, K) __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, lt16, (((GRP) % 6) * PG + (K) * LT) * 16, 0))
#define WD_ISSUE_B(GRP, SET)                                                                                        \
    do { SET.v0 = WD_LOAD_B(GRP, 0); SET.v1 = WD_LOAD_B(GRP, 1); SET.v2 = WD_LOAD_B(GRP, 2);                          \
         SET.v3 = WD_LOAD_B(GRP, 3); SET.v4 = WD_LOAD_B(GRP, 4); SET.v5 = WD_LOAD_B(GRP, 5); } while (0)
#endif
#define WD_WRITE_B(GRP, SET)                                                                                        \
    do { u32x4* dst_ = reinterpret_cast<u32x4*>(lds_b + ((GRP) & 1) * GB) + lt;                                     \
         dst_[0] = SET.v0; dst_[LT] = SET.v1; dst_[2 * LT] = SET.v2; dst_[3 * LT] = SET.v3; dst_[4 * LT] = SET.v4; dst_[5 * LT] = SET.v5; } while (0)
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
        TileCtr cur = ctr0, nxt = ctr0, nxt2 = ctr0;    // this tile, the next one, the one after
        nxt.next();
        nxt2.next(); nxt2.next();
        [[maybe_unused]] int sidx = 0;
        {   // first tile: all four raw planes (its q planes are written in its window, like every column start)
            const Coord c0 = coord_of(cur);
            issue(S[0][0], c0, 0, true); issue(S[0][1], c0, 1, true); issue(S[1][0], c0, 2, true); issue(S[1][1], c0, 3, true);
            WD_ISSUE_B(0, bw[0]);
            WD_WRITE_B(0, bw[0]);                       // (nobody reads the weight buffers before the first b2)
            WD_ISSUE_B(1, bw[0]);
        }
        bool early = false;                             // q0..q2 of the current tile were written under the previous tile's groups
        auto item = [&](auto parc, const int it) {
            constexpr int P = decltype(parc)::value;    // S[P] = this tile's p0, p1;  S[1 - P] = its p2, p3
            const bool more = it + 1 < nitems;
            const bool ncont = more && nxt.pos != 0;    // the next tile continues this column: its p0, p1 are this tile's p2, p3
            const Coord nx = coord_of(nxt);
            STAMP(wave, sidx, lane);
            MSNET_LDS_BARRIER();                        // b1: the MFMA waves are done with the previous tile (plane 3, both weight buffers)
            STAMP(wave, sidx, lane);
#ifdef EXP_WD_P2_LATE
            if (!early) { write_q(I0{}, S[P], S[1 - P]); write_q(I1{}, S[P], S[1 - P]); write_q(I2{}, S[P], S[1 - P]); }   // column start
            // the next tile's p2 goes into this tile's p0 registers (dead once q0 is written): requested here already, two and a
            // half groups before its first use; p3 follows behind b2 into the p1 registers, which q3 below still reads
            issue(S[P][0], nx, 2, more);
#else
            // column start: its q0..q2 are built here, and the next tile's p2 is requested into this tile's p0 registers (dead once
            // q0 is written).  For a continuing tile that request went out six slots ago (behind g3 of the previous item).
            if (!early) {
                write_q(I0{}, S[P], S[1 - P]); write_q(I1{}, S[P], S[1 - P]); write_q(I2{}, S[P], S[1 - P]);
                issue(S[P][0], nx, 2, more);
            }
#endif
            write_q(I3{}, S[P], S[1 - P]);             // (weight group 0 was copied under the previous tile's last group)
            STAMP(wave, sidx, lane);
            MSNET_LDS_BARRIER();                        // b2: tile and weight group 0 are in LDS
            STAMP(wave, sidx, lane);
            // next tile: its p2, p3 always go into this tile's p0 / p1 registers (dead since the window); a column start also
            // fetches its own p0, p1 into this tile's p2 / p3 registers and builds all its q planes in its window
#ifdef EXP_WD_P2_LATE
            issue(S[P][1], nx, 3, more);
            if (!ncont) { issue(S[1 - P][0], nx, 0, more); issue(S[1 - P][1], nx, 1, more); }
            WD_WRITE_B(1, bw[0]); WD_ISSUE_B(2, bw[0]);
#else
            // (the weight request first: vmcnt counts in order, so the copy of group 2 one slot on must not have to wait for the
            // plane requests -- HBM -- that would otherwise sit in front of it)
            WD_WRITE_B(1, bw[0]); WD_ISSUE_B(2, bw[0]);
            issue(S[P][1], nx, 3, more);
            if (!ncont) issue(S[1 - P][1], nx, 1, more);            // (a column start's p0 follows behind g3, see there)
#endif
            STAMP(wave, sidx, lane);
            MSNET_LDS_BARRIER();                        // g0
            STAMP(wave, sidx, lane);
            WD_WRITE_B(2, bw[0]); WD_ISSUE_B(3, bw[0]);
            STAMP(wave, sidx, lane);
            MSNET_LDS_BARRIER();                        // g1: taps 0..11 done, q0 is dead
            STAMP(wave, sidx, lane);
            if (ncont) write_q(I0{}, S[1 - P], S[P]);
            WD_WRITE_B(3, bw[0]); WD_ISSUE_B(4, bw[0]);
            STAMP(wave, sidx, lane);
            MSNET_LDS_BARRIER();                        // g2: taps ..17 done, q1 is dead
            STAMP(wave, sidx, lane);
            if (ncont) write_q(I1{}, S[1 - P], S[P]);
            WD_WRITE_B(4, bw[0]); WD_ISSUE_B(5, bw[0]);
            STAMP(wave, sidx, lane);
            MSNET_LDS_BARRIER();                        // g3
            STAMP(wave, sidx, lane);
            WD_WRITE_B(5, bw[0]); WD_ISSUE_B(6, bw[0]);      // (group 6 = the next tile's group 0)
#ifndef EXP_WD_P2_LATE
            // The registers of p2 (= the next tile's p0) are dead since q0' was written behind g1: the p2 of the tile AFTER next goes into
            // them, six slots before its first use (q0'' behind the next g1) instead of two and a half -- the
            // q writes no longer wait for HBM.  At a column start (no q0' here) the same request fetches the next tile's p0.
            {
                const Coord nx2 = coord_of(ncont ? nxt2 : nxt);
                issue(S[1 - P][0], nx2, ncont ? 2 : 0, ncont ? it + 2 < nitems : more);
            }
#endif
            STAMP(wave, sidx, lane);
            MSNET_LDS_BARRIER();                        // g4: taps ..29 done, q2 is dead
            STAMP(wave, sidx, lane);
            if (ncont) write_q(I2{}, S[1 - P], S[P]);
            WD_WRITE_B(6, bw[0]); WD_ISSUE_B(7, bw[0]);      // the next tile's group 0 into buffer 0 (free since g4), its group 1 requested
            early = ncont;
            cur = nxt; nxt.next(); nxt2.next();
        };
        for (int it = 0; it < nitems; it += 2) {
            item(I0{}, it);
            if (it + 1 < nitems) item(I1{}, it + 1);
        }
#undef WD_ISSUE_B
#undef WD_WRITE_B
        return;
    }

    // ------------------------------ MFMA waves ------------------------------
    const int r = lane & 31, hh = lane >> 5;
    const int row = wave;                               // output row of the tile (both planes)
    f32x16 acc0[4], acc1[4];
    const int stride_w = a.Co;
    const size_t osample = (size_t)a.OD * a.OH * a.OW * a.Co * 4;
    int pn = 0, pod0 = 0, poh0 = 0, pow0 = 0;
    bool pending = false;
    // DRAIN (layers without a residual): the finished tile is parked in `pend` (o0, o1 already combined) and stored one element
    // per two K-steps under the next tile's first 64 steps instead of in a 32 KB burst at the hand-over (conv3d_k3s1_f16s_ws, SLIDE)
    f32x16 pend[2];
    unsigned pbase[2] = {0xffffffffu, 0xffffffffu};
    int plw[2] = {0, 0};
    float psc = 1.f, psh = 0.f, pamax = 0.f;
    bool pend_live = false;
    __amdgpu_buffer_rsrc_t pend_rs = make_rsrc(a.y, 0);
    auto park = [&](int n, int od0, int oh0, int ow0) {
        pend_rs = make_rsrc(a.y + (size_t)n * (osample / 4), osample);
        const int oh = oh0 + row, owb = ow0 + 4 * hh;
        psc = a.scale ? a.scale[r] : 1.f;
        psh = a.shift ? a.shift[r] : 0.f;
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
            const int od = od0 + bb;
            const bool rowok = od < a.OD && oh < a.OH;
            plw[bb] = rowok ? a.OW - owb : 0;
            pbase[bb] = (unsigned)((((size_t)od * a.OH + oh) * a.OW + owb) * a.Co + r) * 4u;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float m0 = acc0[0][e] + acc1[0][e] * kLoInv, m1 = acc0[1][e] + acc1[1][e] * kLoInv;
                const float m2 = acc0[2][e] + acc1[2][e] * kLoInv, m3 = acc0[3][e] + acc1[3][e] * kLoInv;
                pend[bb][e] = bb == 0 ? (m0 + m1) + m2 : (m1 - m2) - m3;
            }
        }
        pend_live = true;
    };
    auto drain_piece = [&](auto qc) {
        constexpr int q = decltype(qc)::value;
        if constexpr (q < 32) {
            constexpr int e = q % 16, bb = q / 16, c = (e & 3) + 8 * (e >> 2);
            const bool ok = c < plw[bb];                // nothing parked: plw == 0, the store is dropped
            const unsigned o = ok ? pbase[bb] + (unsigned)(c * stride_w) * 4u : 0xffffffffu;
            float val = pend[bb][e] * psc + psh;
            if (a.relu) val = fmaxf(val, 0.f);
            pamax = fmaxf(pamax, ok ? fabsf(val) : 0.f);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), pend_rs, o, 0, 0);
        }
    };
    auto epilogue = [&](int n, int od0, int oh0, int ow0) {
        const auto rs_y = make_rsrc(a.y + (size_t)n * (osample / 4), osample);
        const auto rs_res = make_rsrc(a.res ? a.res + (size_t)n * (osample / 4) : nullptr, a.res ? osample : 0);
        const int oh = oh0 + row, owb = ow0 + 4 * hh;
        const float sc = a.scale ? a.scale[r] : 1.f, sh = a.shift ? a.shift[r] : 0.f;
        const int wlim = a.OW - owb;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int od = od0 + b;
            const bool rowok = od < a.OD && oh < a.OH;
            const unsigned off = (unsigned)((((size_t)od * a.OH + oh) * a.OW + owb) * a.Co + r) * 4u;
            f32x16 v, rv;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float m0 = acc0[0][e] + acc1[0][e] * kLoInv, m1 = acc0[1][e] + acc1[1][e] * kLoInv;
                const float m2 = acc0[2][e] + acc1[2][e] * kLoInv, m3 = acc0[3][e] + acc1[3][e] * kLoInv;
                v[e] = b == 0 ? (m0 + m1) + m2 : (m1 - m2) - m3;
                rv[e] = 0.f;
            }
            if (a.res) residual_prefetch<32>(rv, rs_res, off, 0, stride_w * 4, [&](int, int lw) { return rowok && lw < wlim; });
            epilogue_store<32>(v, rv, sc, sh, rs_y, off, 0, stride_w * 4, a.relu, [&](int, int lw) { return rowok && lw < wlim; }, a.oflag);
        }
    };
    // fragment addresses: A = record (k, row + kh, r + kw) = per-lane base [kw][ks] + the immediate ((kh * 4 + k) * IW) * RB;
    // the lo half lives at base ^ 64 (the immediate is a multiple of 128).  B = weight buffer (g & 1), tap t: one base + immediate.
    const unsigned char* abase_hi[3][2];
    const unsigned char* abase_lo[3][2];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int col = r + kw;
            abase_hi[kw][ks] = lds + (row * 4 * IW + col) * RB + (((ks * 2 + hh) ^ ((col >> 1) & 7)) << 4);
            abase_lo[kw][ks] = lds + (((row * 4 * IW + col) * RB + (((ks * 2 + hh) ^ ((col >> 1) & 7)) << 4)) ^ 64);
        }
    const unsigned char* const bbase = lds_b + lane * 16;
    TileCtr ctr = ctr0;
    [[maybe_unused]] int sidx = 0;
    for (int it = 0; it < nitems; ++it) {
        const Coord c = coord_of(ctr);
        ctr.next();
        STAMP(wave, sidx, lane);
        MSNET_LDS_BARRIER();                            // b1
        STAMP(wave, sidx, lane);
        if (pending) {
            if (!a.res) park(pn, pod0, poh0, pow0);
            else epilogue(pn, pod0, poh0, pow0);
            pending = false;
        }
        STAMP(wave, sidx, lane);
        MSNET_LDS_BARRIER();                            // b2
        STAMP(wave, sidx, lane);
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc0[k][e] = 0.f; acc1[k][e] = 0.f; }
        // 72 K-steps (36 taps x two 16-channel halves), fragments two steps ahead in a ring of three
        half8 ah[3], al[3], bh_[3], bl[3];
        auto frag_a = [&](auto sc_) {
            constexpr int s = decltype(sc_)::value, T = s / 2, ks = s % 2, k = T / 9, kh = (T % 9) / 3, kw = T % 3;
            constexpr int IMM = ((kh * 4 + k) * IW) * RB;
            ah[s % 3] = *reinterpret_cast<const half8*>(abase_hi[kw][ks] + IMM);
#if defined(EXP_WD_SKIP_AL)          // timing experiment only (wrong results): how much the LDS fragment reads cost
            al[s % 3] = ah[s % 3];
#else
            al[s % 3] = *reinterpret_cast<const half8*>(abase_lo[kw][ks] + IMM);
#endif
        };
        auto frag_b = [&](auto sc_) {
            constexpr int s = decltype(sc_)::value, T = s / 2, ks = s % 2, g = T / 6, t = T % 6;
            constexpr int IMM = (g & 1) * GB + ((t * 2 + ks) * 2) * 1024;
            bh_[s % 3] = *reinterpret_cast<const half8*>(bbase + IMM);
#if defined(EXP_WD_SKIP_BL)
            bl[s % 3] = bh_[s % 3];
#else
            bl[s % 3] = *reinterpret_cast<const half8*>(bbase + IMM + 1024);
#endif
        };
        frag_a(std::integral_constant<int, 0>{}); frag_b(std::integral_constant<int, 0>{});
        frag_a(std::integral_constant<int, 1>{}); frag_b(std::integral_constant<int, 1>{});
        static_for<72>([&](auto sc_) {
            constexpr int s = decltype(sc_)::value, T = s / 2, k = T / 9, g = T / 6;
            constexpr int s2 = s + 2, g2 = (s2 / 2) / 6;
            // prefetch step s+2: A always (the tile is stable across the group barriers and a q plane outlives its last tap),
            // B only inside the same group -- the next group's buffer is published by the barrier that ends this one
            if constexpr (s2 < 72) {
                frag_a(std::integral_constant<int, s2>{});
                if constexpr (g2 == g) frag_b(std::integral_constant<int, s2>{});
            }
            acc0[k] = mfma16(ah[s % 3], bh_[s % 3], acc0[k]);
            acc1[k] = mfma16(al[s % 3], bh_[s % 3], acc1[k]);
            acc1[k] = mfma16(ah[s % 3], bl[s % 3], acc1[k]);
            if constexpr (s % 2 == 0) drain_piece(std::integral_constant<int, s / 2>{});      // 32 stores over the first 64 steps
            {   // interleave: one MFMA, then LDS reads of the prefetch (4: two behind the first MFMA, else one each), two VALU
                constexpr int NRD_ = s2 < 72 ? (g2 == g ? 4 : 2) : 0;
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if constexpr (NRD_ == 4) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                else if constexpr (NRD_ == 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if constexpr (NRD_ >= 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if constexpr (NRD_ == 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                if constexpr (s % 2 == 0 && s < 64) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);     // the drained store
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (s % 12 == 11 && s < 71) {     // end of a six-tap group
                STAMP(wave, sidx, lane);
                MSNET_READER_BARRIER();                 // g_g (reads in flight: A fragments of live q planes only)
                STAMP(wave, sidx, lane);
                frag_b(std::integral_constant<int, s + 1>{});
                frag_b(std::integral_constant<int, s + 2>{});
            }
        });
        if (pend_live) { flag_overflow(a.oflag, pamax); pamax = 0.f; pend_live = false; plw[0] = 0; plw[1] = 0; }
        pending = true; pn = c.n; pod0 = c.od0; poh0 = c.oh0; pow0 = c.ow0;
    }
    if (pending) epilogue(pn, pod0, poh0, pow0);
}

// Sliding-column launch of the Winograd-depth kernel (same segmentation rule as launch_f16s_slide); -1: shape not eligible.
static int launch_wd_f16s(const char* name, ConvArgs a, hipStream_t s) {
    constexpr int TD = 2, TH = 4, TW = 32;
    a.ntd = cdiv(a.OD, TD); a.nth = cdiv(a.OH, TH); a.ntw = cdiv(a.OW, TW);
    a.ngroups = 1; a.nbtot = 1;
    const size_t cols = (size_t)a.N * a.nth * a.ntw;
    if (cols == 0 || cols * a.ntd > 0x7fffffffu) return fail("%s: bad tile count", name);
    if ((size_t)a.OD * a.OH * a.OW * a.Co * 4 > 0xfffffff0u || (size_t)a.D * a.H * a.W * a.Ci * 4 > 0x7ffffff0u) return -1;
    const double G = (double)num_cus();
    double best = 1e300;
    int best_seg = 1;
    for (int seg = 1; seg <= a.ntd; ++seg) {            // balance over the CUs vs column starts (four planes instead of two)
        if (a.ntd % seg) continue;
        const int len = a.ntd / seg;
        const double cost = ceil((double)cols * seg / G) * (1.0 + 0.85 * (len - 1));
        if (cost < best) { best = cost; best_seg = seg; }
    }
    a.nseg = best_seg; a.seglen = a.ntd / best_seg;
    const size_t units = cols * a.nseg;
    const size_t nblk = units < (size_t)num_cus() ? units : (size_t)num_cus();
    const double vox = (double)a.N * a.OD * a.OH * a.OW;
    LaunchScope ls(name, s, 2.0 * 27.0 * a.Ci * a.Co * vox, 4.0 * ((double)a.N * a.D * a.H * a.W * a.Ci + vox * a.Co * (a.res ? 2 : 1)));
    hipLaunchKernelGGL(conv3d_wd_f16s_kernel, dim3((unsigned)nblk), dim3(512), 0, s, a);
    return check_launch(name);
}

}  // namespace msnet

using namespace msnet;

#ifdef EXP_STAMP
extern "C" int msnet_debug_read_stamps(unsigned long long* host) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 12 * 128) == hipSuccess ? 0 : 1;
}
#endif

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
        hipStream_t s8 = (hipStream_t)stream;
        LaunchScope ls("pack_weight_f16s", s8, 0, 6.0 * 28 * 8 * Co);
        hipLaunchKernelGGL(pack_weight_c8_f16s_kernel, dim3(64), dim3(256), 0, s8, w, (_Float16*)packed, Co);
        return check_launch("msnet_pack_conv_weight_f16s");
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
            return launch_direct_f16s<false>(stride == 2 ? "conv3d_s2_f16s" : (Co == 32 ? "conv3d_s1_f16s_co32" : "conv3d_s1_f16s_co64"),
                                             a, stride, stride == 2 ? 1 : 2, Co == 32 ? 1 : 2, s);
    }
    if (stride == 2)   // 2x2x32 output tile <- 5x5x65 input voxels x 16 channels (104 KB of 64-byte swizzled records); 4 M-blocks, one per MFMA wave
        return launch_f16s<2, 2, 32, 32, 1, 2, S2_SWZ, 1, false, 2, S2_LOADER_WAVES>("conv3d_s2_f16s", a, s);
    //                                    TD TH TW  BW MB NB
    if (Ci == 8) {
        if (Co == 64) return launch_c8_f16s<2>("conv3d_s1_c8_f16s", a, s);
        return launch_c8_f16s<1>("conv3d_s1_c8_f16s", a, s);
    }
    if (Ci == 16) {                                     // single 16-channel chunk, streamed weight groups of 3 K-steps
        if (Co == 64) return launch_f16s<2, 4, 32, 32, 2, 2, false, 1, false>("conv3d_s1_c16_f16s", a, s);
        return launch_f16s<2, 4, 32, 32, 2, 1, false, 1, false>("conv3d_s1_c16_f16s", a, s);
    }
    if (Co % 64 == 0) {
        // widths that are 16 mod 32 (240, 120, ...): 16-wide M-block rows leave no half-empty edge tile and a smaller halo
        if (W % 32 == 16 && H % 8 == 0) return launch_f16s<2, 8, 16, 16, 2, 2, true, 2, false>("conv3d_s1_f16s_co64", a, s);
        return launch_f16s<2, 4, 32, 32, 2, 2, true, 2, false>("conv3d_s1_f16s_co64", a, s);
    }
    // (2x8x16 tiles and swizzled 128-byte records were measured for this layer too: both 4 % slower than padded 2x4x32)
    if (Ci == 32) {                                     // single 32-channel chunk: sliding window along d
        const int rc = launch_f16s_slide<4, 32, 2, 1>("conv3d_s1_f16s_co32", a, s);
        if (rc >= 0) return rc;
    }
    return launch_f16s<2, 4, 32, 32, 2, 1, false, 2, false>("conv3d_s1_f16s_co32", a, s);
}

// First layer straight from the module's NCDHW volume (8 planes): conv3d_c8_f16s_kernel<NB, true>.  Always the tiled kernel
// (no small-layer direct path), so the summation order does not depend on the size.
extern "C" int msnet_conv3d_k3_c8_ncdhw_f16s(const float* x_ncdhw, const void* wpk_f16s, const float* scale, const float* shift,
                                             float* y, int N, int D, int H, int W, int Co, int relu, msnet_stream_t stream) {
    if (!x_ncdhw || !wpk_f16s || !y) return fail("msnet_conv3d_k3_c8_ncdhw_f16s: null pointer");
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0) return fail("msnet_conv3d_k3_c8_ncdhw_f16s: empty input");
    if (Co != 32 && Co != 64) return fail("msnet_conv3d_k3_c8_ncdhw_f16s: Co=%d (32 or 64)", Co);
    ConvArgs a{};
    a.x = x_ncdhw; a.wpk = reinterpret_cast<const f32x4*>(wpk_f16s); a.scale = scale; a.shift = shift; a.res = nullptr; a.y = y;
    a.N = N; a.D = D; a.H = H; a.W = W; a.Ci = 8; a.Co = Co; a.relu = relu; a.oflag = overflow_flag();
    a.OD = D; a.OH = H; a.OW = W;
    hipStream_t s = (hipStream_t)stream;
    if (Co == 64) return launch_c8_f16s<2, true>("conv3d_s1_c8_f16s", a, s);
    return launch_c8_f16s<1, true>("conv3d_s1_c8_f16s", a, s);
}

// First layer on a channels-last MODULE INPUT x: f32[N][D][H][W][8] (msnet_build_volume_ndhwc's layout): the NDHWC first-layer
// kernel plus the fp16-range check of the module input that the layout-conversion pass carries on the NCDHW route.
extern "C" int msnet_conv3d_k3_c8_in_f16s(const float* x_ndhwc, const void* wpk_f16s, const float* scale, const float* shift,
                                          float* y, int N, int D, int H, int W, int Co, int relu, msnet_stream_t stream) {
    if (!x_ndhwc || !wpk_f16s || !y) return fail("msnet_conv3d_k3_c8_in_f16s: null pointer");
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0) return fail("msnet_conv3d_k3_c8_in_f16s: empty input");
    if (Co != 32 && Co != 64) return fail("msnet_conv3d_k3_c8_in_f16s: Co=%d (32 or 64)", Co);
    ConvArgs a{};
    a.x = x_ndhwc; a.wpk = reinterpret_cast<const f32x4*>(wpk_f16s); a.scale = scale; a.shift = shift; a.res = nullptr; a.y = y;
    a.N = N; a.D = D; a.H = H; a.W = W; a.Ci = 8; a.Co = Co; a.relu = relu; a.oflag = overflow_flag();
    a.OD = D; a.OH = H; a.OW = W;
    hipStream_t s = (hipStream_t)stream;
    if (Co == 64) return launch_c8_f16s<2, false, true>("conv3d_s1_c8_f16s", a, s);
    return launch_c8_f16s<1, false, true>("conv3d_s1_c8_f16s", a, s);
}

// Winograd-depth form of a 32 -> 32 stride-1 layer: g36 = f32 [32][32][36] transformed (BN-folded, pre-scaled) weights, tap
// T = k*9 + kh*3 + kw with g0 = w[kd=0], g1 = (w0+w1+w2)/2, g2 = (w0-w1+w2)/2, g3 = w[kd=2]; packed = 36*32*32*2 fp16 (73,728 B).
extern "C" int msnet_pack_conv_weight_wd_f16s(const float* g36, void* packed, msnet_stream_t stream) {
    if (!g36 || !packed) return fail("msnet_pack_conv_weight_wd_f16s: null pointer");
    hipStream_t s = (hipStream_t)stream;
    LaunchScope ls("pack_weight_f16s", s, 0, 6.0 * 36 * 32 * 32);
    hipLaunchKernelGGL(pack_weight_wd_f16s_kernel, dim3(144), dim3(256), 0, s, g36, (_Float16*)packed);
    return check_launch("msnet_pack_conv_weight_wd_f16s");
}

// 1 when msnet_conv3d_k3_wd_f16s takes the layer: stride 1, 32 -> 32 channels, and large enough for the tiled kernels
extern "C" int msnet_conv3d_k3_wd_f16s_supported(int D, int H, int W, int Ci, int Co, int stride) {
    if (stride != 1 || Ci != 32 || Co != 32 || D < 2) return 0;
    // 32-bit byte offsets inside a sample (drained stores, loader descriptor): larger samples take the direct kernel
    if ((size_t)D * H * W * Co * 4 > 0xfffffff0u || (size_t)D * H * W * Ci * 4 > 0x7ffffff0u) return 0;
    ConvArgs a{};
    a.D = a.OD = D; a.H = a.OH = H; a.W = a.OW = W; a.Ci = Ci; a.Co = Co; a.N = 1;
    const size_t items = (size_t)cdiv(D, 2) * cdiv(H, 4) * cdiv(W, 32);
    return direct_eligible(a, items) ? 0 : 1;
}

extern "C" int msnet_conv3d_k3_wd_f16s(const float* x, const void* wpk_wd, const float* scale, const float* shift,
                                       const float* residual, float* y, int N, int D, int H, int W, int relu,
                                       msnet_stream_t stream) {
    if (!x || !wpk_wd || !y) return fail("msnet_conv3d_k3_wd_f16s: null pointer");
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0) return fail("msnet_conv3d_k3_wd_f16s: empty input");
    ConvArgs a{};
    a.x = x; a.wpk = reinterpret_cast<const f32x4*>(wpk_wd); a.scale = scale; a.shift = shift; a.res = residual; a.y = y;
    a.N = N; a.D = D; a.H = H; a.W = W; a.Ci = 32; a.Co = 32; a.relu = relu; a.oflag = overflow_flag();
    a.OD = D; a.OH = H; a.OW = W;
    const int rc = launch_wd_f16s("conv3d_s1_wd_f16s", a, (hipStream_t)stream);
    if (rc < 0) return fail("msnet_conv3d_k3_wd_f16s: a sample exceeds the kernel's 32-bit offset range");
    return rc;
}

// Ci = 64 with Co = 32 / 64 has the tiled kernel; the other shapes (and any small layer) run on the direct kernel.
extern "C" int msnet_deconv3d_k3s2_f16s_supported(int Ci, int Co) {
    return ((Ci == 32 || Ci == 64 || Ci == 128) && Co > 0 && Co % 32 == 0) ? 1 : 0;
}

// Deconv weights for the split-fp16 path: w f32[Ci][Co][3][3][3] -> packed (msnet_packed_weight_floats(Ci,Co) floats).
extern "C" int msnet_pack_deconv_weight_f16s(const float* w, void* packed, int Ci, int Co, msnet_stream_t stream) {
    if (!w || !packed) return fail("msnet_pack_deconv_weight_f16s: null pointer");
    if (!msnet_deconv3d_k3s2_f16s_supported(Ci, Co)) return fail("msnet_pack_deconv_weight_f16s: unsupported Ci=%d Co=%d", Ci, Co);
    const size_t total = (size_t)27 * Ci * Co * 2;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipStream_t s = (hipStream_t)stream;
    LaunchScope ls("pack_weight_f16s", s, 0, 6.0 * total);
    hipLaunchKernelGGL(pack_deconv_weight_f16s_kernel, dim3(blocks), dim3(256), 0, s, w, (_Float16*)packed, Ci, Co, Ci / 16, 1);
    return check_launch("msnet_pack_deconv_weight_f16s");
}

extern "C" int msnet_deconv3d_k3s2_f16s(const float* x, const void* wpk_f16s, const float* scale, const float* shift,
                                        const float* residual, float* y, int N, int D, int H, int W, int Ci, int Co,
                                        int relu, msnet_stream_t stream) {
    if (!x || !wpk_f16s || !y) return fail("msnet_deconv3d_k3s2_f16s: null pointer");
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0) return fail("msnet_deconv3d_k3s2_f16s: empty input");
    if (!msnet_deconv3d_k3s2_f16s_supported(Ci, Co)) return fail("msnet_deconv3d_k3s2_f16s: unsupported Ci=%d Co=%d", Ci, Co);
    ConvArgs a{};
    a.x = x; a.wpk = reinterpret_cast<const f32x4*>(wpk_f16s); a.scale = scale; a.shift = shift; a.res = residual; a.y = y;
    a.N = N; a.D = D; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co; a.relu = relu; a.oflag = overflow_flag();
    a.OD = 2 * D; a.OH = 2 * H; a.OW = 2 * W;
    const bool tiled_ok = Ci == 64 && (Co == 32 || Co == 64);
    const size_t items = (size_t)cdiv(D, 2) * cdiv(H, 4) * cdiv(W, 32) * (Co / 32);      // per sample (batch-invariant choice)
    if (!tiled_ok || direct_eligible(a, items)) {
        if (!direct_shape_ok(a)) return fail("msnet_deconv3d_k3s2_f16s: Ci=%d Co=%d at this size needs the fp32 kernel", Ci, Co);
        return launch_direct_f16s<true>("deconv3d_f16s", a, 2, Ci / 16, 1, (hipStream_t)stream);
    }
    return launch_deconv_f16s<4, 1>("deconv3d_f16s", a, (hipStream_t)stream);
}

// Shared by the split-fp16 conv translation units (conv3d_f16s*.hip): operand types, the fp16 MFMA wrapper, the hi/lo split,
// the work-item counter of the persistent kernels, and the host-side launch functions each unit exports.
//
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
#pragma once
#include <stdlib.h>

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
#define S2_SWZ true
#ifndef MSNET_A_AUX
#define MSNET_A_AUX 0           // cache-policy bits of the loaders' tile requests (2 = nt, measured: see DESIGN.md 4.1d)
#endif
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // first-class vector (HIP's uint4 is a class)

__device__ __forceinline__ f32x16 mfma16(half8 a, half8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}


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

// ---- (class, tap) order of the transposed conv (k3, s2, p1, op1): shared by the tiled deconv kernel, the direct kernel's
// transposed mode and the weight packer ---------------------------------------------------------------------------------------
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
static __constant__ DeconvTapTable kDeconvTaps = make_deconv_taps();

// ---- launch functions exported by the translation units (each instantiates its own kernels; -1 from the slide / Winograd
// launchers: shape not taken, the caller falls back) ----------------------------------------------------------------------------
int ws_launch_s2(const char* name, ConvArgs a, hipStream_t s);                      // conv3d_f16s_ws_s2.hip
int ws_launch_c16(bool co64, const char* name, ConvArgs a, hipStream_t s);          // conv3d_f16s_ws_c16.hip
int ws_launch_co64(bool w16, const char* name, ConvArgs a, hipStream_t s);          // conv3d_f16s_ws_co64.hip
int ws_launch_co32(const char* name, ConvArgs a, hipStream_t s);                    // conv3d_f16s_ws_co32.hip
int ws_launch_co32_slide(const char* name, ConvArgs a, hipStream_t s);              // conv3d_f16s_ws_co32.hip
int ws_launch_co32_inchk(const char* name, ConvArgs a, hipStream_t s);              // conv3d_f16s_ws_co32.hip
int c8_launch(int nb, bool ncs, bool inchk, const char* name, ConvArgs a, hipStream_t s);      // conv3d_f16s_c8.hip
int c8_pack_launch(const float* w, _Float16* packed, int Co, hipStream_t s);        // conv3d_f16s_c8.hip
int direct_launch(bool transposed, const char* name, ConvArgs a, int stride, int KS, int NBG, hipStream_t s);   // conv3d_f16s_direct.hip

// shapes the direct kernel can run at all
inline bool direct_shape_ok(const ConvArgs& a) {
    const int KK = a.Ci / 16;
    if (a.Ci % 16 || !(KK == 2 || KK == 4 || KK == 8) || a.Co % 32) return false;
    return (size_t)a.D * a.H * a.W * a.Ci * 4 <= 0xfffffff0u && (size_t)a.OD * a.OH * a.OW * a.Co * 4 <= 0xfffffff0u;     // per sample
}
// a layer is "small" when the tiled kernel would have work for fewer than a quarter of the CUs (measured: at 108 tiles the
// tiled kernel still wins, 49 vs 61 us; at 30-54 tiles the direct one does, 37 vs 75 and 23 vs 40 us)
inline bool direct_eligible(const ConvArgs& a, size_t tiled_items) {
    if (!direct_shape_ok(a)) return false;
    if (const char* e = getenv("MSNET_DIRECT")) {       // test hook: "0" never, "1" whenever the shape allows
        if (e[0] == '0') return false;
        if (e[0] == '1') return true;
    }
    return tiled_items < (size_t)num_cus() / 4;
}

}  // namespace msnet

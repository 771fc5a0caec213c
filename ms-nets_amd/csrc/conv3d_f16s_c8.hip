#include "conv_f16s.h"

namespace msnet {
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
// (Round 6: THREE workgroups per CU -- the last K-step's weights in registers so that 3 x 52,736 B fit the LDS, 170 registers per
// lane -- was built and measured: 27-60 spilled registers, layer +20 %, step -3.6 %; profiles/r06_c8_wg3.txt.)
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
                epilogue_store<32>(v, zero, sc, sh, rs_y, off, 0, a.Co * 4, a.relu, [&](int, int lw) { return rowok && lw < wlim; }, a.oflag);
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
                   4.0 * ((double)a.N * a.D * a.H * a.W * a.Ci + vox * a.Co * (a.res ? 2 : 1)), true);
    MSNET_LAUNCH(ls, (conv3d_c8_f16s_kernel<NB, NCS, INCHK>), dim3((unsigned)nblk), dim3(256), 0, s, a);
    return check_launch(name);
}

int c8_launch(int nb, bool ncs, bool inchk, const char* name, ConvArgs a, hipStream_t s) {
    if (nb == 2) {
        if (ncs) return launch_c8_f16s<2, true>(name, a, s);
        if (inchk) return launch_c8_f16s<2, false, true>(name, a, s);
        return launch_c8_f16s<2>(name, a, s);
    }
    if (ncs) return launch_c8_f16s<1, true>(name, a, s);
    if (inchk) return launch_c8_f16s<1, false, true>(name, a, s);
    return launch_c8_f16s<1>(name, a, s);
}

int c8_pack_launch(const float* w, _Float16* packed, int Co, hipStream_t s) {
    LaunchScope ls("pack_weight_f16s", s, 0, 6.0 * 28 * 8 * Co);
    hipLaunchKernelGGL(pack_weight_c8_f16s_kernel, dim3(64), dim3(256), 0, s, w, packed, Co);
    return check_launch("msnet_pack_conv_weight_f16s");
}

}  // namespace msnet

using namespace msnet;

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
    return c8_launch(Co == 64 ? 2 : 1, true, true, "conv3d_s1_c8_f16s", a, s);
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
    return c8_launch(Co == 64 ? 2 : 1, false, true, "conv3d_s1_c8_f16s", a, s);
}

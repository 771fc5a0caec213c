// Stride-2 conv (Ci = 32 / 64 -> Co = 64) with 8-CHANNEL chunks and two taps per K-step -- round-4 experiment for the
// `Conv3DBlock` stride-2 heads (gcnet_3dcnn.py:108-114) and the hourglass conv1 / conv3 (psmnet_3dcnn.py:69-75).
//
// Why: the wave-specialised stride-2 kernel (conv_f16s_ws.h, STRIDE = 2) stages 16 channels of a 5x5x65 (or 5x9x33) input tile
// = 128 output voxels, one M-block per MFMA wave: a K-step is 2 A + 4 B fragment reads for 6 MFMAs and a weight group only 18
// MFMAs long, its MFMA pipe is ~32 % busy.  Two M-blocks per wave need a 256-voxel tile, whose 5x9x65 input
// voxels only fit the LDS 8 channels at a time -- the first layer's scheme (conv3d_f16s_c8.hip): 32-byte records (hi | lo of 8
// channels), a 16-wide K-step = TWO taps x 8 channels (lane half hh reads the voxel shifted by tap 2s + hh), 14 K-steps per chunk.
//
// Shape: 2x4x32 output tile, 512 threads, no wave specialisation: wave w owns M-block pair w >> 1 (two output rows of one depth
// plane) and N-block w & 1 (32 of the 64 output channels): 64 accumulator registers, kept across the Ci / 8 chunk items of a
// tile.  Per item every thread stages its share of the NEXT item in registers (12 float4 of activations, 7 of packed weights)
// while the MFMAs of the current one run; between two barriers the registers are split and dropped into LDS (tile 95 KB with
// the columns of a row de-interleaved -- even columns first -- so that a tap reads consecutive records at stride 2; weights 56 KB).
//   packed weights (16-byte units): idx = (((c*14 + s)*2 + nb)*2 + hl)*64 + lane, element j of lane (r, hh):
//       W[co = nb*32 + r][ci = 8c + j][tap = 2s + hh]  (zero for tap 27);  hl = 0 hi, 1 lo.
#include "conv_f16s.h"

namespace msnet {

__global__ void pack_weight_s2c8_f16s_kernel(const float* __restrict__ w, _Float16* __restrict__ out, int Ci) {
    const int NC = Ci / 8;
    const size_t total = (size_t)NC * 14 * 2 * 2 * 64 * 8;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        size_t i = o;
        const int j = i & 7; i >>= 3;
        const int lane = i & 63; i >>= 6;
        const int hl = i & 1; i >>= 1;
        const int nb = i & 1; i >>= 1;
        const int sstep = (int)(i % 14);
        const int c = (int)(i / 14);
        const int tap = 2 * sstep + (lane >> 5), co = nb * 32 + (lane & 31), ci = 8 * c + j;
        const float v = tap < 27 ? w[((size_t)co * Ci + ci) * 27 + tap] : 0.f;
        const _Float16 h = (_Float16)v;
        out[o] = hl ? (_Float16)((v - (float)h) * kLoScale) : h;
    }
}

__global__ __launch_bounds__(512) void conv3d_s2c8_f16s_kernel(ConvArgs a) {
    constexpr int TD = 2, TH = 4, TW = 32, ID = 2 * TD + 1, IH = 2 * TH + 1, IW = 2 * TW + 1, CH = (IW + 1) / 2, RP = 2 * CH;
    constexpr int NPOS = ID * IH * IW, NREC = ID * IH * RP, NT = 512;
    constexpr int NL = (NPOS * 2 + NT - 1) / NT;                        // float4 (channel quads) per thread per item: 12
    constexpr int WB = 14 * 2 * 2 * 1024, NWB = WB / 16 / NT;           // weight bytes per chunk, 16-byte pieces per thread: 7
    static_assert(NWB * NT * 16 == WB, "weight pieces divide evenly");
    static_assert(NREC * 32 + WB <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) unsigned char lds_a[NREC * 32];
    __shared__ __attribute__((aligned(16))) unsigned char lds_b[WB];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int mp = wave >> 1, nb = wave & 1;                           // M-block pair (bd = mp >> 1, bh = (mp & 1)*2 + i), N-block
    const unsigned G = gridDim.x, lb = xcd_remap(blockIdx.x, G);
    const unsigned T = (unsigned)a.N * a.ntd * a.nth * a.ntw;
    const int nch = a.Ci / 8;
    const int ntiles = (T > lb) ? (int)((T - lb + G - 1) / G) : 0;
    const int nitems = ntiles * nch;
    if (nitems == 0) return;

    // loader role: slot = u*NT + tid -> (pos = slot >> 1, quad = slot & 1)
    const size_t isample = (size_t)a.D * a.H * a.W * a.Ci * 4;
    f32x4 av[NL];
    u32x4 bw[NWB];
    // slot coordinates are RE-COMPUTED where they are needed (two constant divisions per slot, ~250 VALU per item and thread against
    // ~2700 MFMA cycles): held in registers they would be 24 of the 256 a thread has, next to 76 of prefetched data and 64 accumulators
    auto slot_dhw = [&](int u, int t) {                  // (id << 16) | (ih << 8) | iw, or -1 past the tile's end
        const int pos = (u * NT + t) >> 1;
        const int row = pos / IW, iw = pos - row * IW, id = row / IH, ih = row - id * IH;
        return pos < NPOS ? ((id << 16) | (ih << 8) | iw) : -1;
    };
    TileCtr ctr, nxt;                                    // current item / the one being fetched; pos = channel chunk
    ctr.init(lb, G, 1, a.ntw, a.nth, a.ntd, nch);
    nxt = ctr;
    const auto rs_w = make_rsrc(a.wpk, (size_t)nch * WB);
    auto issue = [&](const TileCtr& c) {
        const int d0 = c.td * TD, h0 = c.th * TH, w0 = c.tw * TW;
        const int gd0 = 2 * d0 - 1, gh0 = 2 * h0 - 1, gw0 = 2 * w0 - 1;
        const unsigned base = (unsigned)(((((long)gd0 * a.H + gh0) * a.W + gw0) * a.Ci + c.pos * 8) * 4);   // may wrap; in-range slots bring it back
        const bool interior = gd0 >= 0 && gd0 + ID <= a.D && gh0 >= 0 && gh0 + IH <= a.H && gw0 >= 0 && gw0 + IW <= a.W;
        const auto rsrc = make_rsrc(a.x + (size_t)c.n * (isample / 4), isample);
        int tq = tid;
        asm volatile("" : "+v"(tq));                      // (keeps the slot arithmetic inside this call: not hoisted out of the item loop)
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int dhw = slot_dhw(u, tq), id = dhw >> 16, ih = (dhw >> 8) & 255, iw = dhw & 255, q = tq & 1;
            bool ok = dhw >= 0;
            if (!interior) ok = ok && (unsigned)(gd0 + id) < (unsigned)a.D && (unsigned)(gh0 + ih) < (unsigned)a.H && (unsigned)(gw0 + iw) < (unsigned)a.W;
            const unsigned rel = (unsigned)((((id * a.H + ih) * a.W + iw) * a.Ci + q * 4) * 4);
            av[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok ? base + rel : 0xffffffffu, 0, 0));
        }
        // (through a descriptor: one per-thread byte offset + scalar offsets -- 64-bit global addresses would cost 14 registers)
#pragma unroll
        for (int k = 0; k < NWB; ++k)
            bw[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)tq * 16u, (unsigned)(c.pos * WB + k * NT * 16), 0));
    };
    auto write_ab = [&]() {
        int tq = tid;
        asm volatile("" : "+v"(tq));
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int dhw = slot_dhw(u, tq);
            if (dhw >= 0) {
                const int id = dhw >> 16, ih = (dhw >> 8) & 255, iw = dhw & 255, q = tq & 1;
                const int rec = (id * IH + ih) * RP + (iw & 1) * CH + (iw >> 1);      // even columns first, then the odd ones
                half4 hi, lo;
                split4(av[u], hi, lo);
                const int sw = ((rec >> 3) & 1) * 16;
                *reinterpret_cast<half4*>(lds_a + rec * 32 + sw + q * 8) = hi;
                *reinterpret_cast<half4*>(lds_a + rec * 32 + (sw ^ 16) + q * 8) = lo;
            }
        }
        u32x4* dst = reinterpret_cast<u32x4*>(lds_b);
#pragma unroll
        for (int k = 0; k < NWB; ++k) dst[k * NT + tid] = bw[k];
    };
    // MFMA role: M-block i of the wave = output row (bd, bh_i); its lane r is output column r, i.e. input column 2r + kw
    int vox0[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) vox0[i] = ((2 * (mp >> 1)) * IH + 2 * ((mp & 1) * 2 + i)) * RP + r;
    const size_t osample = (size_t)a.OD * a.OH * a.OW * a.Co * 4;
    f32x16 acc0[2], acc1[2];
    const int co = nb * 32 + r;                          // this lane's output channel; its BN constants live in two registers
    const float sc = a.scale ? a.scale[co] : 1.f, sh = a.shift ? a.shift[co] : 0.f;

    issue(nxt);
    for (int it = 0; it < nitems; ++it) {
        const int n = ctr.n, d0 = ctr.td * TD, h0 = ctr.th * TH, w0 = ctr.tw * TW, chunk = ctr.pos;
        ctr.next();
        nxt.next();
        __syncthreads();                                // previous item fully consumed
        write_ab();
        __syncthreads();
        if (it + 1 < nitems) issue(nxt);                // in flight during the MFMAs (and the epilogue)
        if (chunk == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) { acc0[i][e] = 0.f; acc1[i][e] = 0.f; }
        }
        static_for<14>([&](auto sc_) {
            constexpr int sstep = decltype(sc_)::value;
            constexpr int t0 = 2 * sstep, t1 = 2 * sstep + 1 < 27 ? 2 * sstep + 1 : 26;
            constexpr int off0 = ((t0 / 9) * IH + (t0 / 3) % 3) * RP + ((t0 % 3) & 1) * CH + ((t0 % 3) >> 1);
            constexpr int off1 = ((t1 / 9) * IH + (t1 / 3) % 3) * RP + ((t1 % 3) & 1) * CH + ((t1 % 3) >> 1);
            half8 ah[2], al[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rec = vox0[i] + (hh ? off1 : off0);
                const int sw = ((rec >> 3) & 1) * 16;
                ah[i] = *reinterpret_cast<const half8*>(lds_a + rec * 32 + sw);
                al[i] = *reinterpret_cast<const half8*>(lds_a + rec * 32 + (sw ^ 16));
            }
            const half8 bh_ = *reinterpret_cast<const half8*>(lds_b + ((sstep * 2 + nb) * 2) * 1024 + lane * 16);
            const half8 bl = *reinterpret_cast<const half8*>(lds_b + ((sstep * 2 + nb) * 2 + 1) * 1024 + lane * 16);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc0[i] = mfma16(ah[i], bh_, acc0[i]);
                acc1[i] = mfma16(al[i], bh_, acc1[i]);
                acc1[i] = mfma16(ah[i], bl, acc1[i]);
            }
            __builtin_amdgcn_sched_barrier(0);          // fragment reads are not hoisted across K-steps (registers)
        });
        if (chunk == nch - 1) {
            // epilogue: lane = output channel, register e = voxel (e&3) + 8*(e>>2) + 4*hh of the 32-voxel row
            const auto rs_y = make_rsrc(a.y + (size_t)n * (osample / 4), osample);
            const auto rs_res = make_rsrc(a.res ? a.res + (size_t)n * (osample / 4) : nullptr, a.res ? osample : 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int od = d0 + (mp >> 1), oh = h0 + (mp & 1) * 2 + i, owb = w0 + 4 * hh;
                const bool rowok = od < a.OD && oh < a.OH;
                const int wlim = a.OW - owb;
                const unsigned off = (unsigned)((((size_t)od * a.OH + oh) * a.OW + owb) * a.Co + co) * 4u;
                f32x16 v, rv;
#pragma unroll
                for (int e = 0; e < 16; ++e) { v[e] = acc0[i][e] + acc1[i][e] * kLoInv; rv[e] = 0.f; }
                if (a.res) residual_prefetch<32>(rv, rs_res, off, 0, a.Co * 4, [&](int, int lw) { return rowok && lw < wlim; });
                epilogue_store<32>(v, rv, sc, sh, rs_y, off, 0, a.Co * 4, a.relu, [&](int, int lw) { return rowok && lw < wlim; }, a.oflag);
            }
        }
    }
}

static int launch_s2c8_f16s(const char* name, ConvArgs a, hipStream_t s) {
    a.ntd = cdiv(a.OD, 2); a.nth = cdiv(a.OH, 4); a.ntw = cdiv(a.OW, 32);
    const size_t ntiles = (size_t)a.N * a.ntd * a.nth * a.ntw;
    if (ntiles == 0 || ntiles > 0x7fffffffu) return fail("%s: bad tile count %zu", name, ntiles);
    if ((size_t)a.OD * a.OH * a.OW * a.Co * 4 > 0xfffffff0u || (size_t)a.D * a.H * a.W * a.Ci * 4 > 0xfffffff0u)
        return fail("%s: a sample exceeds the 4 GB buffer-descriptor range", name);
    const size_t nblk = ntiles < (size_t)num_cus() ? ntiles : (size_t)num_cus();
    const double vox = (double)a.N * a.OD * a.OH * a.OW;
    LaunchScope ls(name, s, 2.0 * 27.0 * a.Ci * a.Co * vox, 4.0 * ((double)a.N * a.D * a.H * a.W * a.Ci + vox * a.Co * (a.res ? 2 : 1)));
    hipLaunchKernelGGL(conv3d_s2c8_f16s_kernel, dim3((unsigned)nblk), dim3(512), 0, s, a);
    return check_launch(name);
}

}  // namespace msnet

using namespace msnet;

// 1 when msnet_conv3d_k3s2_c8_f16s takes the layer: Ci = 32 / 64 -> Co = 64, and large enough for a tiled kernel
extern "C" int msnet_conv3d_k3s2_c8_f16s_supported(int D, int H, int W, int Ci, int Co) {
    if ((Ci != 32 && Ci != 64) || Co != 64 || D <= 0 || H <= 0 || W <= 0) return 0;
    ConvArgs a{};
    a.D = D; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co; a.N = 1;
    a.OD = (D - 1) / 2 + 1; a.OH = (H - 1) / 2 + 1; a.OW = (W - 1) / 2 + 1;
    if ((size_t)a.OD * a.OH * a.OW * Co * 4 > 0xfffffff0u || (size_t)D * H * W * Ci * 4 > 0xfffffff0u) return 0;
    const size_t items = (size_t)cdiv(a.OD, 2) * cdiv(a.OH, 4) * cdiv(a.OW, 32);
    return direct_eligible(a, items) ? 0 : 1;
}

// w: f32 [64][Ci][3][3][3] (BN-folded, pre-scaled) -> packed: (Ci / 8) * 57,344 bytes
extern "C" int msnet_pack_conv_weight_s2c8_f16s(const float* w, void* packed, int Ci, int Co, msnet_stream_t stream) {
    if (!w || !packed) return fail("msnet_pack_conv_weight_s2c8_f16s: null pointer");
    if ((Ci != 32 && Ci != 64) || Co != 64) return fail("msnet_pack_conv_weight_s2c8_f16s: Ci=%d Co=%d (32 / 64 -> 64)", Ci, Co);
    hipStream_t s = (hipStream_t)stream;
    LaunchScope ls("pack_weight_f16s", s, 0, 6.0 * 28 * Ci * Co);
    hipLaunchKernelGGL(pack_weight_s2c8_f16s_kernel, dim3(256), dim3(256), 0, s, w, (_Float16*)packed, Ci);
    return check_launch("msnet_pack_conv_weight_s2c8_f16s");
}

extern "C" int msnet_conv3d_k3s2_c8_f16s(const float* x, const void* wpk_s2c8, const float* scale, const float* shift,
                                         const float* residual, float* y, int N, int D, int H, int W, int Ci, int Co, int relu,
                                         msnet_stream_t stream) {
    if (!x || !wpk_s2c8 || !y) return fail("msnet_conv3d_k3s2_c8_f16s: null pointer");
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0) return fail("msnet_conv3d_k3s2_c8_f16s: empty input");
    if ((Ci != 32 && Ci != 64) || Co != 64) return fail("msnet_conv3d_k3s2_c8_f16s: Ci=%d Co=%d (32 / 64 -> 64)", Ci, Co);
    ConvArgs a{};
    a.x = x; a.wpk = reinterpret_cast<const f32x4*>(wpk_s2c8); a.scale = scale; a.shift = shift; a.res = residual; a.y = y;
    a.N = N; a.D = D; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co; a.relu = relu; a.oflag = overflow_flag();
    a.OD = (D - 1) / 2 + 1; a.OH = (H - 1) / 2 + 1; a.OW = (W - 1) / 2 + 1;
    return launch_s2c8_f16s("conv3d_s2_f16s", a, (hipStream_t)stream);
}

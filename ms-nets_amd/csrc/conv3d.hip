// 3x3x3 convolutions of the cost-volume aggregators as implicit GEMMs on the fp32-input MFMA
// (v_mfma_f32_32x32x2_f32: exact f32 products, f32 accumulate -- the parity-safe matrix path on gfx950).
//
// Replaces the nn.Conv3d / nn.ConvTranspose3d + nn.BatchNorm3d (+ReLU, +skip) stacks of
//   /root/reference/src/models/gcnet_3dcnn.py:20-27,97-122  and  psmnet_3dcnn.py:22-25,41-89,96-122.
//
// Data layout in HBM: activations are channels-last fp32, [N][D][H][W][C]; one voxel's C channels are
// one contiguous 128/256/512-byte run, so every global access below is a full-line access.
//
// Implicit GEMM:  out[pos][co] = sum_{tap, ci} in[pos (+) tap][ci] * w[co][ci][tap]
//   M = 32 output voxels (an "M-block": BH x BW voxels of one depth slice), N = 32 output channels,
//   K = 27 taps x Ci, walked 8 input channels at a time.
// A workgroup stages the input tile + halo for one chunk of CC input channels in LDS once and re-reads it for
// all 27 taps and all output channels; the B operand (weights, pre-packed on the device into MFMA lane order
// by pack.hip) streams from L2 two steps ahead of the MFMAs.
//
// MFMA operand maps (cdna_hip_programming.md section 3): for v_mfma_f32_32x32x2_f32 lane l holds
// A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31]; D[i][j] sits in lane (j | 32*((i>>2)&1)),
// register (i&3) + 4*(i>>3).  The K order inside an 8-channel group is permuted so that one 16-byte
// ds_read_b128 / global_load_dwordx4 per lane feeds four consecutive MFMAs: lane half h reads
// channels 8q+4h .. 8q+4h+3 and MFMA t consumes channel 8q+4h+t from each half (the weights are
// packed with the same permutation, so the sum over K is unchanged).
#include "conv_common.h"

namespace msnet {

// ---------------------------------------------------------------------------------------------
// Wave-specialised, persistent form of the forward conv (the one the big layers use).
//   8 waves per workgroup, two per SIMD: waves 0-3 ("compute") run the software-pipelined MFMA loop and the
//   epilogue; waves 4-7 ("loaders") fetch the NEXT work item's input tile from
//   HBM/L2 into registers while the MFMAs run, then drop it into LDS between two barriers:
//        loader :  issue(k) ... wait        |A_k| write LDS |B_k| issue(k+1) ...
//        compute:  MFMA(k-1)                |A_k| epilogue(k-1) |B_k| MFMA(k) ...
//   so the global-load latency of the tile hides under the previous item's MFMAs and the LDS fill hides under
//   its epilogue.  The loaders have their own vmcnt, so their 20-odd outstanding tile loads never sit in front
//   of the compute waves' counted waits on the weight stream.
//   A workgroup walks work items (tile, Ci-chunk); tiles are dealt so that at any moment the 256 resident
//   workgroups cover one contiguous run of tiles, split per XCD (halo re-reads hit that XCD's L2).
// ---------------------------------------------------------------------------------------------
template <int CC, int STRIDE, int TD, int TH, int TW, int BW, int WM, int WN, int MB, int NB>
__global__ __launch_bounds__(512, 2) void conv3d_k3_mfma_ws(ConvArgs a) {
    constexpr int BH = 32 / BW;
    constexpr int ID = (TD - 1) * STRIDE + 3, IH = (TH - 1) * STRIDE + 3, IW = (TW - 1) * STRIDE + 3;
    constexpr int PS = CC + 4;
    constexpr int NQ = CC / 8;
    constexpr int MW = TW / BW, MH = TH / BH;
    constexpr int V = CC / 4;
    constexpr int NSLOT = ID * IH * IW * V;
    constexpr int NL = (NSLOT + 255) / 256;            // float4 per loader thread
    static_assert(TD * MH * MW == WM * MB, "M-block count mismatch");
    static_assert(WM * WN == 4, "4 compute waves per workgroup");
    __shared__ __attribute__((aligned(16))) float lds[ID * IH * IW * PS];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const unsigned G = gridDim.x;
    const unsigned lb = xcd_remap(blockIdx.x, G);
    const int nchunks = a.Ci / CC;
    const unsigned T = (unsigned)a.N * a.ntd * a.nth * a.ntw * a.ngroups;
    const int my_tiles = (T > lb) ? (int)((T - lb + G - 1) / G) : 0;
    const int nitems = my_tiles * nchunks;
    if (nitems == 0) return;

    auto decode = [&](int it, int& g, int& n, int& od0, int& oh0, int& ow0, int& chunk) {
        unsigned t = lb + (unsigned)(it / nchunks) * G;
        chunk = it % nchunks;
        g = t % a.ngroups; t /= a.ngroups;
        ow0 = (t % a.ntw) * TW; t /= a.ntw;
        oh0 = (t % a.nth) * TH; t /= a.nth;
        od0 = (t % a.ntd) * TD;
        n = t / a.ntd;
    };

    if (wave >= 4) {
        // ------------------------------ loader waves ------------------------------
        const int lt = tid - 256;
        for (int it = 0; it < nitems; ++it) {
            int g, n, od0, oh0, ow0, chunk;
            decode(it, g, n, od0, oh0, ow0, chunk);
            const int id0 = od0 * STRIDE - 1, ih0 = oh0 * STRIDE - 1, iw0 = ow0 * STRIDE - 1;
            const float* xc = a.x + chunk * CC;
            f32x4 v[NL];
#pragma unroll
            for (int u = 0; u < NL; ++u) {
                const int slot = u * 256 + lt;
                const int pos = slot / V, c4 = slot % V;
                const int iw = pos % IW, ih = (pos / IW) % IH, id = pos / (IW * IH);
                const int gd = id0 + id, gh = ih0 + ih, gw = iw0 + iw;
                v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (slot < NSLOT && (unsigned)gd < (unsigned)a.D && (unsigned)gh < (unsigned)a.H &&
                    (unsigned)gw < (unsigned)a.W) {
                    const size_t vox = (((size_t)n * a.D + gd) * a.H + gh) * a.W + gw;
                    v[u] = *reinterpret_cast<const f32x4*>(xc + vox * a.Ci + c4 * 4);
                }
            }
            MSNET_LDS_BARRIER();                       // A_it: compute waves are done reading the previous tile
#pragma unroll
            for (int u = 0; u < NL; ++u) {
                const int slot = u * 256 + lt;
                if (slot < NSLOT) *reinterpret_cast<f32x4*>(lds + (slot / V) * PS + (slot % V) * 4) = v[u];
            }
            MSNET_LDS_BARRIER();                       // B_it: tile is in LDS
        }
        return;
    }

    // ------------------------------ compute waves ------------------------------
    const int wm = wave % WM, wn = wave / WM;
    const int r = lane & 31, hh = lane >> 5;
    int abase[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int mb = wm * MB + i;
        const int bw = mb % MW, bh = (mb / MW) % MH, bd = mb / (MW * MH);
        const int lh = bh * BH + r / BW, lw = bw * BW + r % BW;
        abase[i] = ((bd * STRIDE * IH + lh * STRIDE) * IW + lw * STRIDE) * PS + 4 * hh;
    }
    const int nci8 = a.Ci >> 3;
    const size_t wtap = (size_t)nci8 * a.nbtot * 64;
    const size_t wq = (size_t)a.nbtot * 64;
    const int stride_w = a.Co, stride_h = a.OW * a.Co;

    f32x16 acc[MB][NB];
    MSNET_LDS_BARRIER();                               // A_0
    MSNET_LDS_BARRIER();                               // B_0
    for (int it = 0; it < nitems; ++it) {
        int g, n, od0, oh0, ow0, chunk;
        decode(it, g, n, od0, oh0, ow0, chunk);
        const int nb0 = (g * WN + wn) * NB;
        if (chunk == 0) {
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        }
        const f32x4* wbase = a.wpk + ((size_t)(chunk * NQ) * a.nbtot + nb0) * 64 + lane;

        constexpr int S = 27 * NQ;
        f32x4 av[2][MB], bv[3][NB];
        auto a_load = [&](f32x4 (&dst)[MB], int st) {
            const int tap = st / NQ, q = st % NQ;
            const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
            const int toff = ((kd * IH + kh) * IW + kw) * PS + q * 8;
#pragma unroll
            for (int i = 0; i < MB; ++i) dst[i] = *reinterpret_cast<const f32x4*>(lds + abase[i] + toff);
        };
        auto b_load = [&](f32x4 (&dst)[NB], int st) {
            const int tap = st / NQ, q = st % NQ;
            load_b<NB>(dst, wbase + tap * wtap + q * wq);
        };
        b_load(bv[0], 0);
        if (S > 1) b_load(bv[1], 1);
        a_load(av[0], 0);
#pragma unroll
        for (int st = 0; st < S; ++st) {
            if (st + 2 < S) b_load(bv[(st + 2) % 3], st + 2);
            if (st + 1 < S) a_load(av[(st + 1) & 1], st + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        acc[i][j] = mfma32(av[st & 1][i][t], bv[st % 3][j][t], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        }

        const bool more = it + 1 < nitems;
        if (more) MSNET_LDS_BARRIER();                 // A_{it+1}: LDS may be overwritten now
        if (chunk == nchunks - 1) {
            const bool full_hw = (oh0 + TH <= a.OH) && (ow0 + TW <= a.OW);
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                const int mb = wm * MB + i;
                const int bw = mb % MW, bh = (mb / MW) % MH, bd = mb / (MW * MH);
                const int od = od0 + bd;
                if (od >= a.OD) continue;
                const int ohb = oh0 + bh * BH, owb = ow0 + bw * BW + 4 * hh;
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int co = (nb0 + j) * 32 + r;
                    const float sc = a.scale ? a.scale[co] : 1.f;
                    const float sh = a.shift ? a.shift[co] : 0.f;
                    const size_t base = ((((size_t)n * a.OD + od) * a.OH + ohb) * a.OW + owb) * a.Co + co;
                    epilogue_block<BW>(acc[i][j], sc, sh, a.res, a.y, base, stride_h, stride_w, a.relu, full_hw,
                                       [&](int lh, int lw) { return ohb + lh < a.OH && owb + lw < a.OW; }, a.oflag);
                }
            }
        }
        if (more) MSNET_LDS_BARRIER();                 // B_{it+1}
    }
}

// ---------------------------------------------------------------------------------------------
// Transposed conv, kernel 3, stride 2, pad 1, output_padding 1 (out = 2 x in), as 8 output-parity
// classes sharing one LDS tile.  o = 2i - 1 + k per dim: an even output 2j takes only (k=1, i=j); an odd
// output 2j+1 takes (k=2, i=j) and (k=0, i=j+1).  Class (pd,ph,pw) therefore has 2^(pd+ph+pw) taps; the
// 27 (class, tap) pairs do exactly the 27*Ci*Co MACs per input voxel of the dense definition.
// The whole Ci is resident in LDS (no chunk loop) so classes can be finished one after another with a
// single live accumulator set.  Weights use the same packed order, indexed by the ConvTranspose3d tap.
// ---------------------------------------------------------------------------------------------
// One output-parity class (PD,PH,PW) of the transposed conv for this wave's MB x NB blocks: accumulate its
// 2^(PD+PH+PW) taps from the LDS tile, then run the epilogue.  Steps (delta, q) use the same software pipeline
// as the forward conv (A one step ahead, B two steps ahead, sched_barrier-pinned).
template <int CI, int IH, int IW, int BW, int MH, int MW, int MB, int NB, int PD, int PH, int PW>
__device__ __forceinline__ void deconv_class(const float* lds, const int (&abase)[MB], const f32x4* __restrict__ wbase,
                                             size_t wtap, size_t wq, const ConvArgs& a, int n, int d0, int h0, int w0,
                                             int wm, int nb0, int r, int hh) {
    constexpr int BH = 32 / BW, PS = CI + 4, NQ = CI / 8;
    constexpr int NT = (PD + 1) * (PH + 1) * (PW + 1);
    constexpr int S = NT * NQ;
    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    f32x4 av[2][MB], bv[3][NB];
    auto a_load = [&](f32x4 (&dst)[MB], int st) {
        const int tp = st / NQ, q = st % NQ;
        const int dw = tp % (PW + 1), dh = (tp / (PW + 1)) % (PH + 1), dd = tp / ((PW + 1) * (PH + 1));
        const int toff = ((dd * IH + dh) * IW + dw) * PS + q * 8;
#pragma unroll
        for (int i = 0; i < MB; ++i) dst[i] = *reinterpret_cast<const f32x4*>(lds + abase[i] + toff);
    };
    auto b_load = [&](f32x4 (&dst)[NB], int st) {
        const int tp = st / NQ, q = st % NQ;
        const int dw = tp % (PW + 1), dh = (tp / (PW + 1)) % (PH + 1), dd = tp / ((PW + 1) * (PH + 1));
        const int kd = PD ? (dd ? 0 : 2) : 1;
        const int kh = PH ? (dh ? 0 : 2) : 1;
        const int kw = PW ? (dw ? 0 : 2) : 1;
        load_b<NB>(dst, wbase + ((kd * 3 + kh) * 3 + kw) * wtap + q * wq);
    };
    b_load(bv[0], 0);
    if (S > 1) b_load(bv[1], 1);
    a_load(av[0], 0);
#pragma unroll
    for (int st = 0; st < S; ++st) {
        if (st + 2 < S) b_load(bv[(st + 2) % 3], st + 2);
        if (st + 1 < S) a_load(av[(st + 1) & 1], st + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    acc[i][j] = mfma32(av[st & 1][i][t], bv[st % 3][j][t], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
    }

    constexpr int TH_ = MH * BH, TW_ = MW * BW;
    const bool full_hw = (h0 + TH_ <= a.H) && (w0 + TW_ <= a.W);
    const int stride_w = 2 * a.Co, stride_h = 2 * a.OW * a.Co;
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int mb = wm * MB + i;
        const int bw = mb % MW, bh = (mb / MW) % MH, bd = mb / (MW * MH);
        const int id = d0 + bd;
        if (id >= a.D) continue;
        const int ihb = h0 + bh * BH, iwb = w0 + bw * BW + 4 * hh;        // input voxel of (c_e = 0, half hh)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int co = (nb0 + j) * 32 + r;
            const float sc = a.scale ? a.scale[co] : 1.f;
            const float sh = a.shift ? a.shift[co] : 0.f;
            const size_t base = ((((size_t)n * a.OD + 2 * id + PD) * a.OH + 2 * ihb + PH) * a.OW + 2 * iwb + PW) * a.Co + co;
            epilogue_block<BW>(acc[i][j], sc, sh, a.res, a.y, base, stride_h, stride_w, a.relu, full_hw,
                               [&](int lh, int lw) { return ihb + lh < a.H && iwb + lw < a.W; }, a.oflag);
        }
    }
}

// 8 waves per workgroup, two per SIMD: waves 0-3 and waves 4-7 both cover all WM x WN blocks of the tile but
// take complementary halves of the 8 parity classes (13 vs 14 of the 27 taps), so one set's epilogue (residual
// reads + strided stores, memory-bound) runs under the other set's MFMAs.
template <int CI, int TD, int TH, int TW, int BW, int WM, int WN, int MB, int NB>
__global__ __launch_bounds__(512, 2) void deconv3d_k3s2_mfma(ConvArgs a) {
    constexpr int BH = 32 / BW;
    constexpr int ID = TD + 1, IH = TH + 1, IW = TW + 1;
    constexpr int PS = CI + 4;
    constexpr int NQ = CI / 8;
    constexpr int MW = TW / BW, MH = TH / BH;
    static_assert(TD * MH * MW == WM * MB, "M-block count mismatch");
    static_assert(WM * WN == 4, "4 waves per workgroup");
    __shared__ __attribute__((aligned(16))) float lds[ID * IH * IW * PS];

    unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int g = bid % a.ngroups; bid /= a.ngroups;
    const int tw = bid % a.ntw; bid /= a.ntw;
    const int th = bid % a.nth; bid /= a.nth;
    const int td = bid % a.ntd;
    const int n = bid / a.ntd;
    const int d0 = td * TD, h0 = th * TH, w0 = tw * TW;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cset = wave8 >> 2, wave = wave8 & 3;
    const int wm = wave % WM, wn = wave / WM;
    const int r = lane & 31, hh = lane >> 5;

    stage_tile<CI, ID, IH, IW, 512>(lds, a.x, n, a.D, a.H, a.W, a.Ci, 0, d0, h0, w0, tid);
    __syncthreads();

    int abase[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int mb = wm * MB + i;
        const int bw = mb % MW, bh = (mb / MW) % MH, bd = mb / (MW * MH);
        const int lh = bh * BH + r / BW, lw = bw * BW + r % BW;
        abase[i] = ((bd * IH + lh) * IW + lw) * PS + 4 * hh;
    }
    const int nb0 = (g * WN + wn) * NB;
    const size_t wtap = (size_t)NQ * a.nbtot * 64;
    const size_t wq = (size_t)a.nbtot * 64;
    const f32x4* wbase = a.wpk + (size_t)nb0 * 64 + lane;

#define MSNET_DECONV_CLASS(PD, PH, PW) \
    deconv_class<CI, IH, IW, BW, MH, MW, MB, NB, PD, PH, PW>(lds, abase, wbase, wtap, wq, a, n, d0, h0, w0, wm, nb0, r, hh)
    if (cset == 0) {                    // 8 + 2 + 2 + 1 = 13 taps
        MSNET_DECONV_CLASS(1, 1, 1);
        MSNET_DECONV_CLASS(0, 0, 1);
        MSNET_DECONV_CLASS(0, 1, 0);
        MSNET_DECONV_CLASS(0, 0, 0);
    } else {                            // 4 + 4 + 4 + 2 = 14 taps
        MSNET_DECONV_CLASS(1, 1, 0);
        MSNET_DECONV_CLASS(1, 0, 1);
        MSNET_DECONV_CLASS(0, 1, 1);
        MSNET_DECONV_CLASS(1, 0, 0);
    }
#undef MSNET_DECONV_CLASS
}

// ---------------------------------------------------------------------------------------------
// Host dispatch
// ---------------------------------------------------------------------------------------------
template <int CC, int STRIDE, int TD, int TH, int TW, int BW, int WM, int WN, int MB, int NB>
static int launch_conv_ws(const char* name, ConvArgs a, hipStream_t s) {
    a.ntd = cdiv(a.OD, TD); a.nth = cdiv(a.OH, TH); a.ntw = cdiv(a.OW, TW);
    a.ngroups = a.Co / (32 * WN * NB);
    a.nbtot = a.Co / 32;
    const size_t ntiles = (size_t)a.N * a.ntd * a.nth * a.ntw * a.ngroups;
    if (ntiles == 0 || ntiles > 0x7fffffffu) return fail("%s: bad tile count %zu", name, ntiles);
    const size_t nblk = ntiles < (size_t)num_cus() ? ntiles : (size_t)num_cus();   // persistent: one workgroup per CU
    const double vox = (double)a.N * a.OD * a.OH * a.OW;
    LaunchScope ls(name, s, 2.0 * 27.0 * a.Ci * a.Co * vox,
                   4.0 * ((double)a.N * a.D * a.H * a.W * a.Ci + vox * a.Co * (a.res ? 2 : 1)));
    hipLaunchKernelGGL((conv3d_k3_mfma_ws<CC, STRIDE, TD, TH, TW, BW, WM, WN, MB, NB>), dim3((unsigned)nblk),
                       dim3(512), 0, s, a);
    return check_launch(name);
}

template <int CI, int TD, int TH, int TW, int BW, int WM, int WN, int MB, int NB>
static int launch_deconv(const char* name, ConvArgs a, hipStream_t s) {
    a.ntd = cdiv(a.D, TD); a.nth = cdiv(a.H, TH); a.ntw = cdiv(a.W, TW);
    a.ngroups = a.Co / (32 * WN * NB);
    a.nbtot = a.Co / 32;
    const size_t nblk = (size_t)a.N * a.ntd * a.nth * a.ntw * a.ngroups;
    if (nblk == 0 || nblk > 0x7fffffffu) return fail("%s: bad grid %zu", name, nblk);
    const double ivox = (double)a.N * a.D * a.H * a.W;
    LaunchScope ls(name, s, 2.0 * 27.0 * a.Ci * a.Co * ivox,
                   4.0 * (ivox * a.Ci + 8.0 * ivox * a.Co * (a.res ? 2 : 1)));
    hipLaunchKernelGGL((deconv3d_k3s2_mfma<CI, TD, TH, TW, BW, WM, WN, MB, NB>), dim3((unsigned)nblk), dim3(512),
                       0, s, a);
    return check_launch(name);
}

// Pick the M-block shape that wastes the fewest lanes on the W axis: 1x32 voxels when OW is a multiple of
// 32 (or large), else 2x16.
static inline bool prefer_bw16(int ow) {
    const int w32 = cdiv(ow, 32) * 32, w16 = cdiv(ow, 16) * 16;
    return w16 < w32;
}

}  // namespace msnet

using namespace msnet;

extern "C" int msnet_conv3d_k3(const float* x, const float* wpk, const float* scale, const float* shift,
                               const float* residual, float* y, int N, int D, int H, int W, int Ci, int Co,
                               int stride, int relu, msnet_stream_t stream) {
    if (!x || !wpk || !y) return fail("msnet_conv3d_k3: null pointer");
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0) return fail("msnet_conv3d_k3: empty input %dx%dx%dx%d", N, D, H, W);
    if (stride != 1 && stride != 2) return fail("msnet_conv3d_k3: stride %d not in {1,2}", stride);
    if (Co % 32 != 0 || Co <= 0) return fail("msnet_conv3d_k3: Co=%d must be a positive multiple of 32", Co);
    if (!(Ci == 8 || (Ci > 0 && Ci % 16 == 0))) return fail("msnet_conv3d_k3: Ci=%d must be 8 or a multiple of 16", Ci);
    ConvArgs a{};
    a.x = x; a.wpk = reinterpret_cast<const f32x4*>(wpk); a.scale = scale; a.shift = shift; a.res = residual; a.y = y;
    a.N = N; a.D = D; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co; a.relu = relu; a.oflag = overflow_flag();
    a.OD = (D - 1) / stride + 1; a.OH = (H - 1) / stride + 1; a.OW = (W - 1) / stride + 1;
    hipStream_t s = (hipStream_t)stream;
    const bool two = (Co % 64 == 0);
    const bool b16 = prefer_bw16(a.OW);
    if (stride == 1) {
        if (Ci == 8) {
            //            CC S TD TH TW  BW WM WN MB NB
            if (two) return launch_conv_ws<8, 1, 2, 4, 32, 32, 4, 1, 2, 2>("conv3d_s1_c8", a, s);
            return launch_conv_ws<8, 1, 2, 8, 32, 32, 4, 1, 4, 1>("conv3d_s1_c8", a, s);
        }
        if (Ci % 32 == 0) {
            if (b16) {
                if (two) return launch_conv_ws<32, 1, 2, 8, 16, 16, 4, 1, 2, 2>("conv3d_s1", a, s);
                return launch_conv_ws<32, 1, 2, 8, 16, 16, 4, 1, 2, 1>("conv3d_s1", a, s);
            }
            if (two) return launch_conv_ws<32, 1, 2, 4, 32, 32, 4, 1, 2, 2>("conv3d_s1", a, s);
            return launch_conv_ws<32, 1, 2, 4, 32, 32, 4, 1, 2, 1>("conv3d_s1", a, s);
        }
        if (two) return launch_conv_ws<16, 1, 2, 8, 16, 16, 4, 1, 2, 2>("conv3d_s1", a, s);
        return launch_conv_ws<16, 1, 2, 8, 16, 16, 4, 1, 2, 1>("conv3d_s1", a, s);
    }
    // stride 2: the input halo tile is (2T+1)^3 voxels, so stage 16 channels at a time and keep the tile at
    // 2x4x16 outputs (4 M-blocks); with Co >= 64 the four waves split 2 (M) x 2 (N).
    if (Ci % 16 != 0) return fail("msnet_conv3d_k3: stride 2 needs Ci %% 16 == 0 (got %d)", Ci);
    if (two) return launch_conv_ws<16, 2, 2, 4, 16, 16, 2, 2, 2, 1>("conv3d_s2", a, s);
    return launch_conv_ws<16, 2, 2, 4, 16, 16, 4, 1, 1, 1>("conv3d_s2", a, s);
}

extern "C" int msnet_deconv3d_k3s2(const float* x, const float* wpk, const float* scale, const float* shift,
                                   const float* residual, float* y, int N, int D, int H, int W, int Ci, int Co,
                                   int relu, msnet_stream_t stream) {
    if (!x || !wpk || !y) return fail("msnet_deconv3d_k3s2: null pointer");
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0) return fail("msnet_deconv3d_k3s2: empty input");
    if (Co % 32 != 0 || Co <= 0) return fail("msnet_deconv3d_k3s2: Co=%d must be a positive multiple of 32", Co);
    ConvArgs a{};
    a.x = x; a.wpk = reinterpret_cast<const f32x4*>(wpk); a.scale = scale; a.shift = shift; a.res = residual; a.y = y;
    a.N = N; a.D = D; a.H = H; a.W = W; a.Ci = Ci; a.Co = Co; a.relu = relu; a.oflag = overflow_flag();
    a.OD = 2 * D; a.OH = 2 * H; a.OW = 2 * W;
    hipStream_t s = (hipStream_t)stream;
    const bool two = (Co % 64 == 0);
    const bool b16 = true;   // 2x16-voxel M-blocks measured faster than 1x32 on every deconv shape (r01: 0.82 vs 1.05 ms on deconvbn3)
    switch (Ci) {
    case 32:
        //                              CI TD TH TW  BW WM WN MB NB
        if (b16) { if (two) return launch_deconv<32, 2, 8, 16, 16, 4, 1, 2, 2>("deconv3d", a, s);
                   return launch_deconv<32, 2, 8, 16, 16, 4, 1, 2, 1>("deconv3d", a, s); }
        if (two) return launch_deconv<32, 2, 4, 32, 32, 4, 1, 2, 2>("deconv3d", a, s);
        return launch_deconv<32, 2, 4, 32, 32, 4, 1, 2, 1>("deconv3d", a, s);
    case 64:
        if (b16) { if (two) return launch_deconv<64, 2, 8, 16, 16, 4, 1, 2, 2>("deconv3d", a, s);
                   return launch_deconv<64, 2, 8, 16, 16, 4, 1, 2, 1>("deconv3d", a, s); }
        if (two) return launch_deconv<64, 2, 4, 32, 32, 4, 1, 2, 2>("deconv3d", a, s);
        return launch_deconv<64, 2, 4, 32, 32, 4, 1, 2, 1>("deconv3d", a, s);
    case 128:
        if (two) return launch_deconv<128, 1, 4, 16, 16, 2, 2, 1, 1>("deconv3d", a, s);
        return fail("msnet_deconv3d_k3s2: Ci=128 needs Co %% 64 == 0 (got %d)", Co);
    default:
        return fail("msnet_deconv3d_k3s2: Ci=%d not in {32,64,128}", Ci);
    }
}

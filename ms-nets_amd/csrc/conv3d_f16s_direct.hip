#include "conv_f16s.h"

namespace msnet {
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
                   4.0 * ((double)a.N * a.D * a.H * a.W * a.Ci + (double)total * a.Co * (a.res ? 2 : 1)), true);
    const dim3 g((unsigned)nblocks), b(256);
    if (KK == 2)      MSNET_LAUNCH(ls, (conv3d_direct_f16s_kernel<TR, 2>), g, b, 0, s, a, stride, KS, NBG);
    else if (KK == 4) MSNET_LAUNCH(ls, (conv3d_direct_f16s_kernel<TR, 4>), g, b, 0, s, a, stride, KS, NBG);
    else              MSNET_LAUNCH(ls, (conv3d_direct_f16s_kernel<TR, 8>), g, b, 0, s, a, stride, KS, NBG);
    return check_launch(name);
}

int direct_launch(bool transposed, const char* name, ConvArgs a, int stride, int KS, int NBG, hipStream_t s) {
    if (transposed) return launch_direct_f16s<true>(name, a, stride, KS, NBG, s);
    return launch_direct_f16s<false>(name, a, stride, KS, NBG, s);
}

}  // namespace msnet

#include "conv_f16s.h"

namespace msnet {
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
            goff_[u] = (unsigned)(((ih * a.W + iw) * a.Ci + c4 * 4) * 4);      // a.Ci: channels per input voxel RECORD (32, or 64 for a half of a 64-channel tensor)
            loff_[u] = (ih * 4 * IW + iw) * RB + (c4 & 1) * 8 + (((c4 >> 1) ^ ((iw >> 1) & 7)) << 4);   // (+ k * IW * RB: plane k)
            mask0 |= (sl < PSLOT ? 1u : 0u) << u;
        }
        const size_t sample_bytes = (size_t)a.D * a.H * a.W * a.Ci * 4;
        // Raw planes of the tile whose q planes are being built: p0, p1 in one register set, p2, p3 in the other.  Inside a column
        // the next tile's p0, p1 ARE this tile's p2, p3, so the two sets swap roles from tile to tile (the tile body exists once per
        // parity: no register copies) and only two planes are fetched per tile.
        f32x4 S[2][2][PL];
        // request input plane (od0 - 1 + pl) of the tile at c; out-of-range slots are out-of-range offsets (zeros come back)
        auto issue = [&](f32x4 (&dst)[PL], const Coord& c, int pl, bool live) {
            const int gd = c.od0 - 1 + pl, ih0 = c.oh0 - 1, iw0 = c.ow0 - 1;
            const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x) + (size_t)c.n * (sample_bytes / 4), 0,
                                                                (int)sample_bytes, 0x00020000);
            const unsigned base = (unsigned)((((long)gd * a.H + ih0) * a.W + iw0) * a.Ci) * 4u;
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
        // weight pieces through a buffer descriptor: one per-thread byte offset (lt * 16) + a compile-time scalar offset per piece,
        // instead of 64-bit addresses in VGPRs (36 of them, which hipcc hoists out of the tile loop and -- once anything else
        // needs the registers -- spills; a spill reload waits vmcnt(0), i.e. for every tile request in flight)
        const auto rs_w = make_rsrc(a.wpk, (size_t)6 * GB);
        const unsigned lt16 = (unsigned)lt * 16u;
#define WD_LOAD_B(GRP, K) __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, lt16, (((GRP) % 6) * PG + (K) * LT) * 16, 0))
#define WD_ISSUE_B(GRP, SET)                                                                                        \
    do { SET.v0 = WD_LOAD_B(GRP, 0); SET.v1 = WD_LOAD_B(GRP, 1); SET.v2 = WD_LOAD_B(GRP, 2);                          \
         SET.v3 = WD_LOAD_B(GRP, 3); SET.v4 = WD_LOAD_B(GRP, 4); SET.v5 = WD_LOAD_B(GRP, 5); } while (0)
#define WD_WRITE_B(GRP, SET)                                                                                        \
    do { u32x4* dst_ = reinterpret_cast<u32x4*>(lds_b + ((GRP) & 1) * GB) + lt;                                     \
         dst_[0] = SET.v0; dst_[LT] = SET.v1; dst_[2 * LT] = SET.v2; dst_[3 * LT] = SET.v3; dst_[4 * LT] = SET.v4; dst_[5 * LT] = SET.v5; } while (0)
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
        TileCtr cur = ctr0, nxt = ctr0, nxt2 = ctr0;    // this tile, the next one, the one after
        nxt.next();
        nxt2.next(); nxt2.next();
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
            MSNET_LDS_BARRIER();                        // b1: the MFMA waves are done with the previous tile (plane 3, both weight buffers)
            // column start: its q0..q2 are built here, and the next tile's p2 is requested into this tile's p0 registers (dead once
            // q0 is written).  For a continuing tile that request went out six slots ago (behind g3 of the previous item).
            if (!early) {
                write_q(I0{}, S[P], S[1 - P]); write_q(I1{}, S[P], S[1 - P]); write_q(I2{}, S[P], S[1 - P]);
                issue(S[P][0], nx, 2, more);
            }
            write_q(I3{}, S[P], S[1 - P]);             // (weight group 0 was copied under the previous tile's last group)
            MSNET_LDS_BARRIER();                        // b2: tile and weight group 0 are in LDS
            // next tile: its p2, p3 always go into this tile's p0 / p1 registers (dead since the window); a column start also
            // fetches its own p0, p1 into this tile's p2 / p3 registers and builds all its q planes in its window
            // (the weight request first: vmcnt counts in order, so the copy of group 2 one slot on must not have to wait for the
            // plane requests -- HBM -- that would otherwise sit in front of it)
            WD_WRITE_B(1, bw[0]); WD_ISSUE_B(2, bw[0]);
            issue(S[P][1], nx, 3, more);
            if (!ncont) issue(S[1 - P][1], nx, 1, more);            // (a column start's p0 follows behind g3, see there)
            MSNET_LDS_BARRIER();                        // g0
            WD_WRITE_B(2, bw[0]); WD_ISSUE_B(3, bw[0]);
            MSNET_LDS_BARRIER();                        // g1: taps 0..11 done, q0 is dead
            if (ncont) write_q(I0{}, S[1 - P], S[P]);
            WD_WRITE_B(3, bw[0]); WD_ISSUE_B(4, bw[0]);
            MSNET_LDS_BARRIER();                        // g2: taps ..17 done, q1 is dead
            if (ncont) write_q(I1{}, S[1 - P], S[P]);
            WD_WRITE_B(4, bw[0]); WD_ISSUE_B(5, bw[0]);
            MSNET_LDS_BARRIER();                        // g3
            WD_WRITE_B(5, bw[0]); WD_ISSUE_B(6, bw[0]);      // (group 6 = the next tile's group 0)
            // The registers of p2 (= the next tile's p0) are dead since q0' was written behind g1: the p2 of the tile AFTER next goes into
            // them, six slots before its first use (q0'' behind the next g1) instead of two and a half -- the
            // q writes no longer wait for HBM.  At a column start (no q0' here) the same request fetches the next tile's p0.
            {
                const Coord nx2 = coord_of(ncont ? nxt2 : nxt);
                issue(S[1 - P][0], nx2, ncont ? 2 : 0, ncont ? it + 2 < nitems : more);
            }
            MSNET_LDS_BARRIER();                        // g4: taps ..29 done, q2 is dead
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
    for (int it = 0; it < nitems; ++it) {
        const Coord c = coord_of(ctr);
        ctr.next();
        MSNET_LDS_BARRIER();                            // b1
        if (pending) {
            if (!a.res) park(pn, pod0, poh0, pow0);
            else epilogue(pn, pod0, poh0, pow0);
            pending = false;
        }
        MSNET_LDS_BARRIER();                            // b2
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
            al[s % 3] = *reinterpret_cast<const half8*>(abase_lo[kw][ks] + IMM);
        };
        auto frag_b = [&](auto sc_) {
            constexpr int s = decltype(sc_)::value, T = s / 2, ks = s % 2, g = T / 6, t = T % 6;
            constexpr int IMM = (g & 1) * GB + ((t * 2 + ks) * 2) * 1024;
            bh_[s % 3] = *reinterpret_cast<const half8*>(bbase + IMM);
            bl[s % 3] = *reinterpret_cast<const half8*>(bbase + IMM + 1024);
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
                MSNET_READER_BARRIER();                 // g_g (reads in flight: A fragments of live q planes only)
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
    // (a.Ci / a.Co are record strides here; the kernel always contracts 32 x 32 channels)
    LaunchScope ls(name, s, 2.0 * 27.0 * 32 * 32 * vox, 4.0 * ((double)a.N * a.D * a.H * a.W * 32 + vox * 32 * (a.res ? 2 : 1)), true);
    MSNET_LAUNCH(ls, conv3d_wd_f16s_kernel, dim3((unsigned)nblk), dim3(512), 0, s, a);
    return check_launch(name);
}

}  // namespace msnet

using namespace msnet;

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

// The same kernel on 32-channel SLICES of wider channels-last tensors: x points at the first of its 32 input channels inside
// records of x_channels floats, y (and residual) at the first of its 32 output channels inside records of y_channels floats.
// Experiment entry (a 64 -> 64 layer as four Winograd-depth launches, partial sums handed over through `residual`).
extern "C" int msnet_conv3d_k3_wd_f16s_strided(const float* x, const void* wpk_wd, const float* scale, const float* shift,
                                               const float* residual, float* y, int N, int D, int H, int W, int x_channels,
                                               int y_channels, int relu, msnet_stream_t stream) {
    if (!x || !wpk_wd || !y) return fail("msnet_conv3d_k3_wd_f16s_strided: null pointer");
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0) return fail("msnet_conv3d_k3_wd_f16s_strided: empty input");
    if (x_channels < 32 || x_channels % 32 || y_channels < 32 || y_channels % 32)
        return fail("msnet_conv3d_k3_wd_f16s_strided: record widths %d / %d (multiples of 32)", x_channels, y_channels);
    ConvArgs a{};
    a.x = x; a.wpk = reinterpret_cast<const f32x4*>(wpk_wd); a.scale = scale; a.shift = shift; a.res = residual; a.y = y;
    a.N = N; a.D = D; a.H = H; a.W = W; a.Ci = x_channels; a.Co = y_channels; a.relu = relu; a.oflag = overflow_flag();
    a.OD = D; a.OH = H; a.OW = W;
    // (the descriptors span D*H*W records from the slice's first channel: the last record's tail beyond the slice is never touched)
    const int rc = launch_wd_f16s("conv3d_s1_wd_f16s", a, (hipStream_t)stream);
    if (rc < 0) return fail("msnet_conv3d_k3_wd_f16s_strided: a sample exceeds the kernel's 32-bit offset range");
    return rc;
}

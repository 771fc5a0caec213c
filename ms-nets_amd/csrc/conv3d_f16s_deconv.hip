#include "conv_f16s.h"

namespace msnet {
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
#define MSNET_DBAR() MSNET_LDS_BARRIER()
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
// group G: copy group G+1's weights, request group G+4's, then this group's share of the next tile (APG loads), barrier
#define MSNET_DGROUP(G, SET)                                                        \
    MSNET_WRITE_B(k0 + (G) + 1, SET);                                               \
    MSNET_ISSUE_B(k0 + (G) + 1 + 3, SET);                                           \
    if constexpr ((G) * APG < NL) issue_part((G) * APG, (G) * APG + APG);           \
    MSNET_DBAR();
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
            write_a();
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
                        acc0[i][j] = mfma16(ah[ks & 1][i], bh_[ks & 1][j], acc0[i][j]);
                        acc1[i][j] = mfma16(al[ks & 1][i], bh_[ks & 1][j], acc1[i][j]);
                        acc1[i][j] = mfma16(ah[ks & 1][i], bl[ks & 1][j], acc1[i][j]);
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
    LaunchScope ls(name, s, 2.0 * 27.0 * a.Ci * a.Co * ivox, 4.0 * (ivox * a.Ci + 8.0 * ivox * a.Co * (a.res ? 2 : 1)), true);
    MSNET_LAUNCH(ls, (deconv3d_k3s2_f16s_ws<KS, NB, DEC_LOADER_WAVES>), dim3((unsigned)nblk), dim3(256 + 64 * DEC_LOADER_WAVES), 0, s, a);
    return check_launch(name);
}

}  // namespace msnet

using namespace msnet;

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
        return direct_launch(true, "deconv3d_f16s", a, 2, Ci / 16, 1, (hipStream_t)stream);
    }
    return launch_deconv_f16s<4, 1>("deconv3d_f16s", a, (hipStream_t)stream);
}

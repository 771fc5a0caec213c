// The wave-specialised persistent split-fp16 conv kernel (stride 1 and 2, plain tiles and the sliding window) and its launch
// templates; included by the conv3d_f16s_ws_*.hip units, each of which instantiates one group of shapes.
#pragma once
#include "conv_f16s.h"

namespace msnet {

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
// INCHK (round 6): the input IS a module input (PSMNet's 64-plane volume handed over channels-last): the loaders fold the magnitude
// bits of every staged value into the fp16-range check of the module input (bit 1 of the overflow word, like the first-layer
// kernel) -- the separate read-only pass over the volume (msnet_check_input_range, 0.08 ms at 48x136x240x64) is not needed then.
template <int TD, int TH, int TW, int BW, int MB, int NB, bool SWZ, int KS, bool RESB, int STRIDE, bool SLIDE = false, int LW = 4,
          bool INCHK = false>
__global__ __launch_bounds__(256 + 64 * LW, (256 + 64 * LW) / 256) void conv3d_k3s1_f16s_ws(ConvArgs a) {
    constexpr int LT = 64 * LW;                          // loader threads
    static_assert(!INCHK || (!RESB && !SLIDE && STRIDE == 1), "input check: plain stride-1 tiles (every staged value passes write_a's split)");
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
    constexpr bool KHS = B3 && KS == 2 && MB == 2 && BW == 32 && !SWZ && (TH / (32 / BW)) % 2 == 0 && LW == 4;
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
        [[maybe_unused]] unsigned in_amax = 0u;         // INCHK: running max of the staged values' magnitude BITS (inf / NaN compare high)
        auto write_a = [&](const f32x4 (&src)[PL], int pslot, int u0, int u1, auto prec) {
            constexpr bool PRE = decltype(prec)::value;
            struct H2 { half4 a, b; };
#pragma unroll
            for (int u = 0; u < PL; ++u) {
                if (u < u0 || u >= u1) continue;
                if (u * LT + lt < PSLOT) {
                    half4 hi, lo;
                    if constexpr (PRE) { const H2 t = __builtin_bit_cast(H2, src[u]); hi = t.a; lo = t.b; }
                    else split4(src[u], hi, lo);
                    if constexpr (INCHK && !PRE)
                        in_amax = max(max(in_amax, max(magnitude_bits(src[u][0]), magnitude_bits(src[u][1]))),
                                      max(magnitude_bits(src[u][2]), magnitude_bits(src[u][3])));
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
        SET.v0 = src_[bi_[0]]; SET.v1 = src_[bi_[1]];                                                                  \
        if constexpr (NLB > 2) SET.v2 = src_[bi_[2]];                                                              \
        if constexpr (NLB > 3) { SET.v3 = src_[3 * LT + lt]; SET.v4 = src_[4 * LT + lt]; SET.v5 = src_[5 * LT + lt]; }    \
    } while (0)
#define MSNET_WRITE_B(K, SET)                                                                                      \
    do {                                                                                                           \
        u32x4* dst_ = reinterpret_cast<u32x4*>(lds_b + (B3 ? ((K) - k0) % 3 : ((K) & 1)) * GB);                    \
        dst_[bi_[0]] = SET.v0; dst_[bi_[1]] = SET.v1;                                                                  \
        if constexpr (NLB > 2) dst_[bi_[2]] = SET.v2;                                                              \
        if constexpr (NLB > 3) { dst_[3 * LT + lt] = SET.v3; dst_[4 * LT + lt] = SET.v4; dst_[5 * LT + lt] = SET.v5; }    \
    } while (0)
// slot of group G: copy group G + BA (BA = 2 with three LDS buffers, else 1) and request the group NSETS later into the freed set
// The weight copy + request of a slot come FIRST in it (right behind the barrier that opens it), the tile requests and plane copies
// behind them: vmcnt counts in order, so a weight copy NSETS slots on then waits for tile requests up to the slot BEFORE its own
// request, not including that slot's (HBM) requests.
#define MSNET_GROUP_B(G, PAR)                                                       \
    MSNET_WRITE_B(k0 + (G) + BA, bw[MSNET_SETI((G) + BA, PAR)]);                    \
    MSNET_ISSUE_B((G) + BA + NSETS, bw[MSNET_SETI((G) + BA, PAR)]);
#define MSNET_GROUP_FIRST(PAR) MSNET_GROUP_B(0, PAR)
#define MSNET_GROUP(G, PAR)                                                         \
    MSNET_LDS_BARRIER();                                                            \
    if constexpr ((G) < 7) { MSNET_GROUP_B((G) + 1, PAR) }
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
            for (int it = 0; it < nitems; ++it) {
                const int k0 = it * 9;
                const bool more = it + 1 < nitems;
                const bool cs = cur.pos == 0;           // the current item starts a column: its planes 2,3 (0,1) are not resident
                if (!cs) rot ^= 1;
                const bool ncont = more && nxt.pos != 0;
                const Coord nx = coord_of(nxt);
                const int p0 = ncont ? 2 : 0;           // first missing plane of the next item
                MSNET_LDS_BARRIER();                    // b1: MFMA waves are done with the previous tile
                if (cs) {                               // (LDS copies only inside the branches)
                    if (!early) { write_a(av[0], (2 * rot) & 3, 0, PL, RAW); write_a(av[1], (2 * rot + 1) & 3, 0, PL, RAW); }
                    write_a(av[2], (2 * rot + 2) & 3, 0, PL, RAW); write_a(av[3], (2 * rot + 3) & 3, 0, PL, RAW);
                }
                MSNET_WINDOW_B(0)
                MSNET_LDS_BARRIER();                    // b2: tile and group 0 are in LDS
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
        if constexpr (PRESPLIT) { presplit(av[2]); presplit(av[3]); presplit(av[4]); }   // first item: its window expects hi|lo quads
        // The loader shares each SIMD with an MFMA wave and runs ~3x slower than alone, so its per-item work (28 loads,
        // 28 split+copy, 27 weight pieces) is spread evenly over the nine group slots instead of bunched at the barriers.
        constexpr int H0 = (PL + 2) / 3, H1 = (2 * PL + 2) / 3, HH = (PL + 1) / 2;
        auto item = [&](auto parc, const int it) {
            [[maybe_unused]] constexpr int PAR = decltype(parc)::value;  // it & 1 (two sets); unused with three (9 % 3 == 0: group k0+g uses set g % 3)
            const int k0 = it * 9;
            const bool more = it + 1 < nitems;
            MSNET_LDS_BARRIER();                        // b1: MFMA waves are done with the previous tile
            if (!early) { write_a(av[0], 0, 0, PL, RAW); write_a(av[1], 1, 0, PL, RAW); }
            if constexpr (PRESPLIT) { write_a(av[2], 2, 0, PL, SPLIT); write_a(av[3], 3, 0, PL, SPLIT); write_a(av[4], 4, 0, PL, SPLIT); }
            else { write_a(av[2], 2, 0, PL, RAW); write_a(av[3], 3, 0, PL, RAW); }
            MSNET_WINDOW_B(PAR)
            MSNET_LDS_BARRIER();                        // b2: tile and group 0 are in LDS
            MSNET_GROUP_FIRST(PAR)
            // group g+1 is copied to LDS (and group g+4 requested) while group g is multiplied; barrier g_g ends it.
            // The next tile is requested during groups 0-2; its planes 0 / 1 are copied as soon as they are dead.  (Past the
            // last item the requests are dead -- `more` = false -- and the copies put zeros into planes nobody reads again.)
            const Coord nx = coord_of(nxt);
            // Stride 2 (five planes, 104 KB per item and CU): the next tile's requests go out evenly over all eight slots, SPS per
            // thread and slot.  What limits this kernel is the rate at which a CU can take in lines that miss its L1 -- ~12 B/clk,
            // i.e. ~13 KB per group: with everything in groups 0-2 (or a plane per group in 0-4) single buffer loads took
            // 300-500 cycles to ISSUE, the loader waves reached the group barriers late and the MFMA waves sat there; a
            // probe (a build that waited vmcnt(0) behind each slot's requests) shows the data back ~400 cycles after the last request of a slot has been accepted.
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
            if constexpr (SPREAD) MSNET_SEQ(0);
            else { issue_a(av[0], nx, 0, more, 0, PL); issue_a(av[1], nx, 1, more, 0, HH); }
            MSNET_GROUP(0, PAR)
            if constexpr (SPREAD) MSNET_SEQ(1);
            else { issue_a(av[1], nx, 1, more, HH, PL); issue_a(av[2], nx, 2, more, 0, PL); }
            MSNET_GROUP(1, PAR)
            if constexpr (SPREAD) MSNET_SEQ(2);
            else issue_a(av[3], nx, 3, more, 0, PL);
            MSNET_GROUP(2, PAR)                         // g_2 passed: kd = 0 groups done, plane 0 is dead
            if constexpr (SPREAD) MSNET_SEQ(3);
            write_a(av[0], 0, 0, H0, RAW);
            MSNET_GROUP(3, PAR)
            if constexpr (SPREAD) MSNET_SEQ(4);
            write_a(av[0], 0, H0, H1, RAW);
            MSNET_GROUP(4, PAR)
            if constexpr (SPREAD) MSNET_SEQ(5);
            write_a(av[0], 0, H1, PL, RAW);
            MSNET_GROUP(5, PAR)                         // g_5 passed: kd = 1 groups done, plane 1 is dead
            if constexpr (SPREAD) MSNET_SEQ(6);
            write_a(av[1], 1, 0, HH, RAW);
            if constexpr (PRESPLIT) presplit(av[2]);
            MSNET_GROUP(6, PAR)
            if constexpr (SPREAD) MSNET_SEQ(7);
            write_a(av[1], 1, HH, PL, RAW);
            if constexpr (PRESPLIT) presplit(av[3]);
            MSNET_GROUP(7, PAR)
#undef MSNET_SEQ
            MSNET_TAIL_B(PAR)
            if constexpr (PRESPLIT) presplit(av[4]);    // (its last request went out in slot 7: split behind g_7, before b1)
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
        if constexpr (INCHK) {
            if (a.oflag && in_amax >= kSplitMaxBits) atomicOr(a.oflag, 2u);
        }
        return;
    }

    // ------------------------------ MFMA waves ------------------------------
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
            {   // branch-free: with nothing parked (plw == 0) the offset is out of range and the store is dropped
                static_assert(BW == 32 || !DRAIN, "drained stores assume one-row M-blocks");
                const bool ok = lw < plw[i];
                const unsigned o = ok ? pbase[i][j] + (unsigned)(lh * stride_h + lw * stride_w) * 4u : 0xffffffffu;
                float val = pend[i][j][e] * psc[j] + psh[j];      // (bit_cast applied to the vector element itself reads element 0)
                if (a.relu) val = fmaxf(val, 0.f);
                pamax = fmaxf(pamax, ok ? fabsf(val) : 0.f);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), pend_rs, o, 0, 0);
            }
        }
    };

    TileCtr ctr = ctr0;
    for (int it = 0; it < nitems; ++it) {
        int n, od0, oh0, ow0, chunk, cg;
        coords(ctr, n, od0, oh0, ow0, chunk, cg);
        if (SLIDE && ctr.pos != 0) rot ^= 1;             // next tile of the same column
        ctr.next();
        MSNET_LDS_BARRIER();                            // b1
        if (pending) {
            if (DRAIN && !a.res) park(pn, pod0, poh0, pow0, pcg);
            else epilogue(pn, pod0, poh0, pow0, pcg);
            pending = false;
        }
        MSNET_LDS_BARRIER();                            // b2
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
                    acc0[0][0] = mfma16(rh[S & 1][kh], qh[q % 3], acc0[0][0]);
                    acc1[0][0] = mfma16(rl[S & 1][kh], qh[q % 3], acc1[0][0]);
                    acc1[0][0] = mfma16(rh[S & 1][kh], ql[q % 3], acc1[0][0]);
                    acc0[1][0] = mfma16(rh[S & 1][kh + 1], qh[q % 3], acc0[1][0]);
                    acc1[1][0] = mfma16(rl[S & 1][kh + 1], qh[q % 3], acc1[1][0]);
                    acc1[1][0] = mfma16(rh[S & 1][kh + 1], ql[q % 3], acc1[1][0]);
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
                if constexpr (g < 8) {
                    // g_g: in flight are the A rows and B fragments of group g+1's first steps -- live tile planes and the weight
                    // buffer published at g_(g-1), neither of which a loader writes before g_(g+1): no drain (see do_group)
#ifdef EXP_FULL_GROUP_BARRIER
                    MSNET_LDS_BARRIER();
#else
                    MSNET_READER_BARRIER();
#endif
                }
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
        // FIRST / LAST (compile-time: is this group 0 / group 8 of the item): the rolled loop over groups 1..7 then has NO
        // run-time condition around its fragment prefetches.  With `if (g < 8)` / `if (g == 0)` inside one rolled body the
        // compiler's waitcnt pass had to assume the path WITHOUT the next group's prefetch burst, so the last step of every group
        // waited with lgkmcnt(4) .. lgkmcnt(0) -- i.e. for the 12 reads just issued for the NEXT group -- and every group paid an
        // LDS round trip (~450 of ~1050 cycles per 18-MFMA group of the stride-2 kernel in the per-wave stamps).
        auto do_group = [&](int g, auto firstc, auto lastc, auto drain) {
            constexpr bool FIRST = decltype(firstc)::value, LAST = decltype(lastc)::value;
            const int g3 = g - 3 * ((g * 11) >> 5);     // g % 3 (g < 9)
            const unsigned char* bb = lds_b + (RESB ? g : (B3 ? g3 : ((gg0 + g) & 1))) * GB + lane * 16;
            constexpr bool BEARLY = B3;
            [[maybe_unused]] const unsigned char* bb_next = lds_b + (g3 == 2 ? 0 : g3 + 1) * GB + lane * 16;   // B3: group g+1's buffer
#pragma unroll
            for (int i = 0; i < MB; ++i) { goff[i] = goff_next[i]; goff_next[i] = grp_off(g + 1, i); }   // (kd, kh) rows, in voxels
            if (!BEARLY || FIRST) {                     // (B3: the previous group read these before its barrier)
#pragma unroll
                for (int q = 0; q < PF; ++q) frag_b(q, q, bb);
            }
            static_for<NS>([&](auto sc_) {
                constexpr int s = decltype(sc_)::value;
                if (s + PF < NS) { frag_a(s + PF, (s + PF) % R, goff); frag_b(s + PF, (s + PF) % R, bb); }
                else if constexpr (!LAST) {
                    frag_a(s + PF - NS, (s + PF) % R, goff_next);
                    if constexpr (BEARLY) frag_b(s + PF - NS, (s + PF) % R, bb_next);
                }
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        acc0[i][j] = mfma16(ah[s % R][i], bh_[s % R][j], acc0[i][j]);
                        acc1[i][j] = mfma16(al[s % R][i], bh_[s % R][j], acc1[i][j]);
                        acc1[i][j] = mfma16(ah[s % R][i], bl[s % R][j], acc1[i][j]);
                    }
                drain(sc_);
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
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (!RESB && !LAST) {
                // g_g.  The reads in flight here are fragment prefetches for group g+1: A fragments of tile planes that are still
                // live (the loaders overwrite a plane only after the barrier that ends its LAST group, and group g+1 never reads a
                // plane that dies at g_g), and -- three buffers -- B fragments of buffer g+1, which is next written two barriers
                // later.  The buffer the loaders refill after g_g (group g's) was consumed by this group's MFMAs.  So no drain.
#ifdef EXP_FULL_GROUP_BARRIER
                MSNET_LDS_BARRIER();
#else
                MSNET_READER_BARRIER();
#endif
            }
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

template <int TD, int TH, int TW, int BW, int MB, int NB, bool SWZ, int KS, bool RESB, int STRIDE = 1, int LW = 4, bool INCHK = false>
int launch_f16s(const char* name, ConvArgs a, hipStream_t s) {
    a.ntd = cdiv(a.OD, TD); a.nth = cdiv(a.OH, TH); a.ntw = cdiv(a.OW, TW);
    a.ngroups = a.Co / (32 * NB); a.nbtot = a.Co / 32;
    const size_t ntiles = (size_t)a.N * a.ntd * a.nth * a.ntw * a.ngroups;
    if (ntiles == 0 || ntiles > 0x7fffffffu) return fail("%s: bad tile count %zu", name, ntiles);
    if ((size_t)a.D * a.H * a.W * a.Ci * 4 > 0x7ffffff0u || (size_t)a.OD * a.OH * a.OW * a.Co * 4 > 0xfffffff0u)
        return fail("%s: a sample exceeds the range of the kernel's buffer descriptors (2 GB in, 4 GB out; use the fp32 path)", name);
    const size_t nblk = ntiles < (size_t)num_cus() ? ntiles : (size_t)num_cus();
    const double vox = (double)a.N * a.OD * a.OH * a.OW;
    LaunchScope ls(name, s, 2.0 * 27.0 * a.Ci * a.Co * vox,
                   4.0 * ((double)a.N * a.D * a.H * a.W * a.Ci + vox * a.Co * (a.res ? 2 : 1)), true);
    MSNET_LAUNCH(ls, (conv3d_k3s1_f16s_ws<TD, TH, TW, BW, MB, NB, SWZ, KS, RESB, STRIDE, false, LW, INCHK>), dim3((unsigned)nblk), dim3(256 + 64 * LW), 0, s, a);
    return check_launch(name);
}

// Sliding-window launch for single-chunk stride-1 layers.  A tile column (all d at one (h, w) tile) is cut into `nseg`
// segments that are dealt to the persistent workgroups; within a segment every tile after the first stages two planes
// instead of four (measured: 0.85 of a tile's time), so longer segments are cheaper per tile but balance worse.  Returns -1
// when plain tiles are estimated to be no slower (the caller then launches the ordinary kernel).
template <int TH, int TW, int MB, int NB>
int launch_f16s_slide(const char* name, ConvArgs a, hipStream_t s) {
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
                   4.0 * ((double)a.N * a.D * a.H * a.W * a.Ci + vox * a.Co * (a.res ? 2 : 1)), true);
    MSNET_LAUNCH(ls, (conv3d_k3s1_f16s_ws<TD, TH, TW, 32, MB, NB, false, 2, false, 1, true, SLIDE_LOADER_WAVES>), dim3((unsigned)nblk), dim3(256 + 64 * SLIDE_LOADER_WAVES), 0, s, a);
    return check_launch(name);
}


}  // namespace msnet

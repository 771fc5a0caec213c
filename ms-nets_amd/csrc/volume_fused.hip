// Fast path of msnet_build_volume for the reference's own parameters (windows 11 / 3 / 5 / 5, cbmv_generator.py:434-462):
// three launches from the two bordered uint8 images to the [8, D', H', W'] float32 volume.
//
//   vprep_kernel       per-pixel tables from LDS-staged image tiles: census bit images, NCC window sums + 1/sqrt terms,
//                      ZSAD window means, Sobel images                                  (matchers.cpp:268-300,71-147,472-485,515-554)
//   sadsob_band_kernel the float32 integral image of |SL - SR_shift| NEVER goes to HBM: one workgroup owns one
//                      disparity and one band of rows, runs the reference's sequential vertical pass (one thread per
//                      column, from row 0: the rows above the band are re-accumulated, they cost two cached loads each),
//                      keeps the band in LDS, runs the sequential horizontal pass (one lane per row) and evaluates the
//                      5x5 boxes.  Same additions in the same order as matchers.cpp:388-423, so bit-identical.
//   features_kernel    one thread = one cropped pixel of one matcher; the D' raw costs stay in REGISTERS (the d loop is
//                      fully unrolled, right-image data comes from LDS strips), so each output element is written once
//                      and expf is evaluated once per element:
//                        pass 1  raw cost c_d for every d -> normalised cost channel (cbmv_generator.py:283-287), min
//                        pass 2  e_d = exp(-(c_d - m)^2 / sigma) (aml_numerator, common.h), den += e_d in d order (featextract.cpp:444-447)
//                        pass 3  likelihood channel e_d / den (featextract.cpp:452-458)
//
// HBM traffic per map: the 8 output channels (401 MB at 960x544, D=192) + the parked Sobel-SAD raw costs (50 MB written,
// 50 MB read back) + the small tables.  Built with -ffp-contract=off like volume.hip: float32 operation order is part
// of the reference's result.
#include <stdlib.h>

#include <map>
#include <memory>
#include <mutex>
#include <type_traits>

#include "common.h"

namespace msnet {

typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
constexpr int kCW = 11, kNW = 3, kSW = 5, kZW = 5;      // census / NCC / Sobel-SAD / ZSAD windows of the fast path
constexpr int kMaxBands = 28;                           // Sobel-SAD row bands per image (host-checked)
constexpr int kBandRMin = 11;                           // smallest band height any configuration uses (workspace sizing)

struct FastArgs {
    const uint8_t* l; const uint8_t* r;
    uint4* lb; uint4* rb;               // census bit images [Hb*Wb], 121 bits in 4 words (order private to this file)
    double2* ncl; double2* ncr;         // NCC tables {window sum A, 1/sqrt term C} per pixel
    float* ml; float* mr;               // ZSAD window means
    float* sobl; float* sobr;           // Sobel images
    float* out;
    float kexp[4];                      // aml_scale(sigma) per matcher (common.h), computed on the host
    int Hb, Wb, nd, bh, bw, Hc, Wc;
};

// ------------------------------------------------------------------------------------------------ prep
// 64 x 4 pixel tiles, blockIdx.z: 0 census L, 1 census R, 2 tables (NCC, ZSAD means, Sobel) of both images.
__global__ __launch_bounds__(256) void vprep_kernel(FastArgs a) {
    __shared__ uint8_t tile[2][14][80];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 4;
    const int x = x0 + tx, y = y0 + ty;
    const int W = a.Wb, H = a.Hb;
    if (blockIdx.z < 2) {
        const uint8_t* img = blockIdx.z == 0 ? a.l : a.r;
        for (int k = threadIdx.x; k < 14 * 74; k += 256) {
            const int r = k / 74, c = k % 74;
            const int gy = y0 - 5 + r, gx = x0 - 5 + c;
            tile[0][r][c] = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? img[gy * W + gx] : 0;
        }
        __syncthreads();
        if (x >= W || y >= H) return;
        uint32_t wd[4] = {0u, 0u, 0u, 0u};
        const int i = y - 5, j = x - 5;
        if (i >= 0 && j >= 0 && i < H - kCW && j < W - kCW) {     // matchers.cpp:283 (i < H - wsize, not <=)
            const int c = tile[0][ty + 5][tx + 5];
#pragma unroll
            for (int wh = 0; wh < kCW; ++wh)
#pragma unroll
                for (int ww = 0; ww < kCW; ++ww) {
                    const int b = wh * kCW + ww;
                    const uint32_t lt = (uint32_t)(c - (int)tile[0][ty + wh][tx + ww]) >> 31;   // center < pixel
                    wd[b >> 5] = (wd[b >> 5] << 1) | lt;
                }
        }
        (blockIdx.z == 0 ? a.lb : a.rb)[(size_t)y * W + x] = make_uint4(wd[0], wd[1], wd[2], wd[3]);
        return;
    }
    for (int k = threadIdx.x; k < 2 * 8 * 68; k += 256) {
        const int im = k / (8 * 68), q = k % (8 * 68), r = q / 68, c = q % 68;
        const int gy = y0 - 2 + r, gx = x0 - 2 + c;
        const uint8_t* img = im == 0 ? a.l : a.r;
        tile[im][r][c] = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? img[gy * W + gx] : 0;
    }
    __syncthreads();
    if (x >= W || y >= H) return;
    const size_t p = (size_t)y * W + x;
    {   // NCC 3x3 (window top-left i = y-1, j = x-1)
        const int i = y - 1, j = x - 1;
        double2 tl = make_double2(0.0, 0.0), tr = make_double2(0.0, 0.0);
        if (i >= 0 && j >= 0 && i < H - kNW && j < W - kNW) {
            unsigned sl = 0, sr = 0, ql = 0, qr = 0;
#pragma unroll
            for (int wh = 0; wh < kNW; ++wh)
#pragma unroll
                for (int ww = 0; ww < kNW; ++ww) {
                    const unsigned u = tile[0][ty + 1 + wh][tx + 1 + ww], v = tile[1][ty + 1 + wh][tx + 1 + ww];
                    sl += u; sr += v; ql += u * u; qr += v * v;
                }
            const unsigned long long sq = (unsigned long long)(kNW * kNW);
            tl.x = (double)sl; tr.x = (double)sr;
            tl.y = 1.0 / sqrt((double)(sq * ql) - (double)sl * (double)sl);
            tr.y = 1.0 / sqrt((double)(sq * qr) - (double)sr * (double)sr);
        }
        a.ncl[p] = tl; a.ncr[p] = tr;
    }
    {   // ZSAD means 5x5: sequential float sum of <= 25 bytes is exact, then one division
        const int i = y - 2, j = x - 2;
        float ml = 0.f, mr = 0.f;
        if (i >= 0 && j >= 0 && i < H - kZW && j < W - kZW) {
            unsigned sl = 0, sr = 0;
#pragma unroll
            for (int wh = 0; wh < kZW; ++wh)
#pragma unroll
                for (int ww = 0; ww < kZW; ++ww) { sl += tile[0][ty + wh][tx + ww]; sr += tile[1][ty + wh][tx + ww]; }
            ml = (float)sl / (float)(kZW * kZW);
            mr = (float)sr / (float)(kZW * kZW);
        }
        a.ml[p] = ml; a.mr[p] = mr;
    }
    {   // Sobel, written at (i+1, j+1) for i < H-3, j < W-3
        const int i = y - 1, j = x - 1;
        float vl = 0.f, vr = 0.f;
        if (i >= 0 && j >= 0 && i < H - 3 && j < W - 3) {
#pragma unroll
            for (int im = 0; im < 2; ++im) {
                const int s = -(int)tile[im][ty + 1][tx + 1] + (int)tile[im][ty + 1][tx + 3] - 2 * (int)tile[im][ty + 2][tx + 1] +
                              2 * (int)tile[im][ty + 2][tx + 3] - (int)tile[im][ty + 3][tx + 1] + (int)tile[im][ty + 3][tx + 3];
                (im == 0 ? vl : vr) = (float)s;
            }
        }
        a.sobl[p] = vl; a.sobr[p] = vr;
    }
}

// ------------------------------------------------------------------------------------------------ Sobel-SAD
// The reference's vertical pass is one sequential float32 chain per (disparity, column) over all rows -- but every term
// |SL - SR_shift| is an integer <= 2040 (Sobel responses are integers in [-1020, 1020]) and a column sums to at most
// Hb * 2040 < 2^24 (checked on the host: Hb <= 8000), so every partial sum is an exactly representable integer and the ORDER
// of the vertical additions cannot change a bit.  That makes the vertical pass parallel:
//   sadsob_bandsum_kernel : bs[(d * nbands + b) * LS + c] = sum of |SL - SR_shift| over the image rows of band slot b
//                           (slot 0: rows 0 .. first-1, slot b: the R rows in front of band b's first kept integral row)
//   sadsob_band_kernel    : start value of band b = bs[0] + ... + bs[b], then its own R + 5 rows.
// The horizontal pass is NOT exact (sums reach 3e8) and keeps the reference's strictly sequential order (matchers.cpp:406-411).
template <int R>
__global__ __launch_bounds__(256) void sadsob_bandsum_kernel(FastArgs a, float* __restrict__ bs, int LS, int nbands) {
    const int c = blockIdx.x * 256 + threadIdx.x;          // integral column <-> image column c - 1
    const int b = blockIdx.y, d = blockIdx.z;
    if (c >= LS) return;
    const int W = a.Wb, H = a.Hb;
    const int j = c - 1;
    float sum = 0.f;
    if (c <= W && j >= d) {
        const int first = a.bh - kSW / 2 - 1;              // image rows in front of band 0's first kept integral row
        const int r0 = b == 0 ? 0 : first + (b - 1) * R, r1 = first + b * R;      // image rows [r0, r1)
        const float* pl = a.sobl + j;
        const float* pr = a.sobr + j - d;
        float vl[R], vr[R];                                // (raw loads, a scheduling fence, then the arithmetic: see sadsob_band_kernel)
#pragma unroll
        for (int k = 0; k < R; ++k) { const int row = min(r0 + k, H - 1); vl[k] = pl[row * W]; vr[k] = pr[row * W]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < R; ++k) if (r0 + k < r1) sum += fabsf(vl[k] - vr[k]);
    }
    bs[((size_t)d * nbands + b) * LS + c] = sum;
}

// Workgroup = (disparity d, band of R cropped rows).  LDS holds the R + 5 integral rows the band's boxes touch, row
// stride LS with LS % 8 == 4 so that `lane = row` ds_read_b128 / ds_write_b128 accesses are conflict-free: vertical pass
// (thread = column), horizontal pass (lane = row, strictly sequential from column d+1), then the 5x5 boxes.  `park` receives
// the raw box costs [nd][Hc][Wc] (kSentinel where the reference leaves RAND_MAX).  The integral image never goes to HBM.
template <int R, int NT>
__global__ __launch_bounds__(NT) void sadsob_band_kernel(FastArgs a, const float* __restrict__ bs, float* __restrict__ park, int LS,
                                                         int nbands, int skip) {
    extern __shared__ __attribute__((aligned(16))) float S[];     // [R + 5][LS]
    constexpr int NR = R + kSW;
    static_assert(NR <= 64 && NR % 8 == 0, "band rows: one lane per row, vertical pass in batches of 8");
    const int d = blockIdx.x / nbands;
    const int band = blockIdx.x % nbands;
    const int yc0 = band * R;                              // first cropped row of the band
    const int W = a.Wb, H = a.Hb;
    const int i_lo = yc0 + a.bh - kSW / 2;                 // first integral row kept  (= window top of the band's first row)
    const int tid = threadIdx.x;

    // vertical pass: S[i][c] = S[i-1][c] + |SL[i-1][c-1] - SR[i-1][c-1-d]| for the NR integral rows i_lo .. i_lo+NR-1 (rows
    // past the image repeat the last row's term: they are never read by a box)
    if (!(skip & 1))
    for (int c = tid; c < LS; c += NT) {
        const int j = c - 1;
        float run = 0.f;
        const bool live = c <= W && j >= d;                // other columns stay zero, as the reference leaves them
        if (live) {
            const float* q = bs + (size_t)d * nbands * LS + c;
            float t[kMaxBands];                            // all loads in flight at once (a dependent chain would pay 26 L2 latencies)
#pragma unroll
            for (int b = 0; b < kMaxBands; ++b) t[b] = q[(size_t)min(b, band) * LS];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < kMaxBands; ++b) if (b <= band) run += t[b];       // exact integers (see above)
        }
        const float* pl = a.sobl + (live ? j : 0);
        const float* pr = a.sobr + (live ? j - d : 0);
#pragma unroll
        for (int i0 = 0; i0 < NR; i0 += 16) {
            float vl[16], vr[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (i0 + k < NR) { const int row = min(i_lo - 1 + i0 + k, H - 1); vl[k] = pl[row * W]; vr[k] = pr[row * W]; }
            }
            // all 32 loads of the batch are requested before the first dependent add: without the fence the machine scheduler is free
            // to pair each load with its add (load, load, s_waitcnt vmcnt(0), add ...) -- one L2 round trip per row instead of one per
            // batch -- and did so the moment an unrelated edit moved its register-pressure estimate (45 -> 62 us per map)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (i0 + k < NR) { run = live ? fabsf(vl[k] - vr[k]) + run : 0.f; S[(i0 + k) * LS + c] = run; }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();

    // horizontal pass: 16 columns per step; the reads of the next step are issued before this step's 16 dependent additions
    const int c_last = min(W, W - a.bw + kSW);             // last integral column any cropped box reads
    if (!(skip & 2) && tid < NR) {
        float* row = S + tid * LS;
        float run = 0.f;
        int c = (d + 1) & ~3;                              // columns <= d hold zeros: adding them keeps run == 0
        auto rd = [&](int cc) { return *reinterpret_cast<const f32x4*>(row + min(cc, LS - 4)); };
        auto step = [&](f32x4 (&v)[4], f32x4 (&nx)[4], int cc) {
#pragma unroll
            for (int q = 0; q < 4; ++q) nx[q] = rd(cc + 16 + 4 * q);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v[q][0] = v[q][0] + run; v[q][1] = v[q][1] + v[q][0]; v[q][2] = v[q][2] + v[q][1]; v[q][3] = v[q][3] + v[q][2];
                run = v[q][3];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (cc + 4 * q < LS) *reinterpret_cast<f32x4*>(row + cc + 4 * q) = v[q];
        };
        f32x4 va[4], vb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) va[q] = rd(c + 4 * q);
        for (; c <= c_last; c += 32) {                     // two steps per iteration: the buffers swap roles without copies
            step(va, vb, c);
            if (c + 16 <= c_last) step(vb, va, c + 16);
            else break;
        }
    }
    __syncthreads();

    // boxes: cost(y, x) = S[b][r] - S[b][l] - S[t][r] + S[t][l], window top-left (y-2, x-2); four outputs per thread in flight
    const size_t plane = (size_t)a.Hc * a.Wc;
    const int rows = min(R, a.Hc - yc0);
    const int nout = rows * a.Wc;
    float* const obase = park + (size_t)d * plane + (size_t)yc0 * a.Wc;     // the band's outputs are contiguous: [rows][Wc]
    const int off = a.bw - kSW / 2;
    int rr = 0, x = tid;
    while (x >= a.Wc) { x -= a.Wc; ++rr; }
    if (!(skip & 4))
    for (int p0 = tid; p0 < nout; p0 += 4 * NT) {
        float tl[4], tr[4], bl[4], br[4];
        int jj[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool ok = p0 + u * NT < nout;
            const int j = x + off;
            const float* t = S + (ok ? rr : 0) * LS + (ok ? j : 0);
            tl[u] = t[0]; tr[u] = t[kSW]; bl[u] = t[kSW * LS]; br[u] = t[kSW * LS + kSW];
            jj[u] = j;
            x += NT;
            while (x >= a.Wc) { x -= a.Wc; ++rr; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (p0 + u * NT < nout) {
                float c = kSentinel;
                if (jj[u] >= d) {
                    float q = br[u] - bl[u];
                    q = q - tr[u];
                    q = q + tl[u];
                    c = q;
                }
                obase[p0 + u * NT] = c;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ features
// RN(a / b) from rb = RN(1 / b) in three operations (Markstein): q = RN(a * rb) is within one ulp of a / b, the residual
// r = a - q * b is then exact in one fma, and RN(q + r * rb) is the correctly rounded quotient -- the same bits as the IEEE
// division the reference performs (checked exhaustively for the census costs in tests/test_oracle_matchers.py and bit for
// bit against the dividing kernels of volume.hip in tests/test_gpu_volume.py), at a third of its instructions.
__device__ __forceinline__ float div_rn(float a, float b, float rb) {
    const float q = a * rb;
    const float r = __builtin_fmaf(-q, b, a);
    return __builtin_fmaf(r, rb, q);
}

__device__ __forceinline__ float norm_cost(int M, float c) {
    if (M == 0) return div_rn(fminf(fmaxf(c, 0.f), 120.f), 120.f, 1.f / 120.f);    // cbmv_generator.py:283
    if (M == 1) { float t = fminf(fmaxf(c, -1.f), 1.f); t = 1.f + t; return t * 0.5f; }   // :285 (x / 2 is exact)
    return fminf(fmaxf(c, 0.f), 8192.f) * (1.f / 8192.f);                           // :286-287 (x / 2^13 is exact)
}

__device__ __forceinline__ float aml_e(float c, float m, float kexp) { return aml_numerator(c, m, kexp); }

// M: 0 census, 1 NCC, 2 Sobel-SAD (raw costs parked by sadsob_band_kernel), 3 ZSAD.
// ND: compile-time bound on the disparities (register array); nd <= ND, nd % 8 == 0.
// TR: rows of the pixel tile whose right-image strips are staged together (4: features4_kernel, 1: features_cl_kernel).

// LDS bytes of matcher M's right-image strips for a TR-row tile
template <int M, int ND, int TR> constexpr size_t strip_bytes() {
    constexpr size_t SW_ = 64 + ND - 1;
    return M == 0 ? TR * SW_ * 16 : M == 1 ? TR * SW_ * 16 + (TR + 2) * (SW_ + 2) : M == 3 ? TR * SW_ * 4 + (TR + 4) * (SW_ + 4) * 4 : 0;
}

// Stage the right-image data of the tile rows yb0 .. yb0+TR-1 in LDS: strip column s <-> image column xb0 - (ND-1) + s.
// Called by NT threads with t = 0..NT-1 (the caller synchronises them afterwards).
template <int M, int ND, int TR>
__device__ __forceinline__ void stage_right(const FastArgs& a, unsigned char* smem, int xb0, int yb0, int t, int NT) {
    constexpr int SW_ = 64 + ND - 1;
    const int W = a.Wb, H = a.Hb;
    if (M == 0) {
        uint4* sb = reinterpret_cast<uint4*>(smem);        // [TR][SW_]
        for (int k = t; k < TR * SW_; k += NT) {
            const int r = k / SW_, s = k % SW_;
            const int gy = yb0 + r, gx = xb0 - (ND - 1) + s;
            sb[k] = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? a.rb[(size_t)gy * W + gx] : make_uint4(0, 0, 0, 0);
        }
    } else if (M == 1) {
        double2* st = reinterpret_cast<double2*>(smem);    // [TR][SW_] tables
        uint8_t* si = smem + TR * SW_ * 16;                 // [TR + 2][SW_ + 2] pixels, image column xb0 - 1 - (ND-1) + s
        for (int k = t; k < TR * SW_; k += NT) {
            const int r = k / SW_, s = k % SW_;
            const int gy = yb0 + r, gx = xb0 - (ND - 1) + s;
            st[k] = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? a.ncr[(size_t)gy * W + gx] : make_double2(0.0, 0.0);
        }
        for (int k = t; k < (TR + 2) * (SW_ + 2); k += NT) {
            const int r = k / (SW_ + 2), s = k % (SW_ + 2);
            const int gy = yb0 - 1 + r, gx = xb0 - 1 - (ND - 1) + s;
            si[k] = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? a.r[gy * W + gx] : 0;
        }
    } else if (M == 3) {
        float* sm = reinterpret_cast<float*>(smem);        // [TR][SW_] means
        float* si = sm + TR * SW_;                          // [TR + 4][SW_ + 4] pixels as float, image column xb0 - 2 - (ND-1) + s
        for (int k = t; k < TR * SW_; k += NT) {
            const int r = k / SW_, s = k % SW_;
            const int gy = yb0 + r, gx = xb0 - (ND - 1) + s;
            sm[k] = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? a.mr[(size_t)gy * W + gx] : 0.f;
        }
        for (int k = t; k < (TR + 4) * (SW_ + 4); k += NT) {
            const int r = k / (SW_ + 4), s = k % (SW_ + 4);
            const int gy = yb0 - 2 + r, gx = xb0 - 2 - (ND - 1) + s;
            si[k] = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? (float)a.r[gy * W + gx] : 0.f;
        }
    }
}

// ZSAD raw costs of disparities [D0, D1) of one pixel (matchers.cpp:442-512: 25 window terms, fp32, fixed order) into c[D0..D1).
// A separate function so that a second wave can take the upper half of the range (features_cl_kernel): the sliding 5x5 window
// starts at any D0, and every c[d] is the same chain of operations whichever wave forms it.
template <int ND, int TR, int D0, int D1>
__device__ __forceinline__ void zsad_costs(const FastArgs& a, const unsigned char* smem, int tx, int ty, int xb, int yb, int nd,
                                           float (&c)[ND]) {
    constexpr int SW_ = 64 + ND - 1;
    static_assert(D0 % 8 == 0 && D1 % 8 == 0 && D0 < D1 && D1 <= ND, "disparity range in groups of eight");
    const int W = a.Wb;
    const size_t pl = (size_t)yb * W + xb;
    const float mlv = a.ml[pl];
    const float* sm = reinterpret_cast<const float*>(smem) + ty * SW_ + tx + (ND - 1);
    const float* si = reinterpret_cast<const float*>(smem) + TR * SW_ + ty * (SW_ + 4) + tx + (ND - 1);   // window top-left, d = 0
    float lm[25];
#pragma unroll
    for (int k = 0; k < 25; ++k) lm[k] = (float)a.l[(yb - 2 + k / 5) * W + xb - 2 + k % 5] - mlv;
    const int jmax = xb - kZW / 2;
    // sliding 5x5 window of the right image: relative column q (step d needs -d .. -d+4) lives in slot q mod 5
    float win[5][5];
#pragma unroll
    for (int wh = 0; wh < 5; ++wh)
#pragma unroll
        for (int q = 1; q < 5; ++q) win[wh][(((q - D0) % 5) + 5) % 5] = si[wh * (SW_ + 4) + q - D0];      // columns 1..4 of step d = D0
#pragma unroll
    for (int d0 = D0; d0 < D1; d0 += 8) {
        if (d0 < nd) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int d = d0 + u;
                // new leftmost column (relative column -d) goes to slot (-d) mod 5 = (5 - d % 5) % 5
                const int s0 = (5 - d % 5) % 5;
#pragma unroll
                for (int wh = 0; wh < 5; ++wh) win[wh][s0] = si[wh * (SW_ + 4) - d];
                const float mrv = sm[-d];
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 25; ++k) {
                    float t = lm[k] - win[k / 5][(s0 + k % 5) % 5];
                    t = t + mrv;
                    acc = acc + fabsf(t);
                }
                c[d] = (d <= jmax) ? acc : kSentinel;
                __builtin_amdgcn_sched_barrier(0);      // keeps the LDS reads of later steps from being hoisted (registers)
            }
        }
    }
}

// Pass 1: the nd raw costs of bordered pixel (yb, xb) = tile row ty, tile column tx, into registers.  The border (>= 6) makes
// every window of a cropped pixel spatially valid, so only the disparity range is tested: census d <= xb - 5
// (matchers.cpp:318), the others d <= window-left column.  M == 2 reads the parked Sobel-SAD costs [nd][Hc][Wc] through
// `park` (descriptor) at byte offset pix4 + d * plane4.
template <int M, int ND, int TR>
__device__ __forceinline__ void raw_costs(const FastArgs& a, const unsigned char* smem, int tx, int ty, int xb, int yb, int nd,
                                          __amdgpu_buffer_rsrc_t park, unsigned pix4, unsigned plane4, float (&c)[ND]) {
    constexpr int SW_ = 64 + ND - 1;
    const int W = a.Wb;
    const size_t pl = (size_t)yb * W + xb;
    if (M == 0) {
        const uint4 lw = a.lb[pl];
        const uint4* sb = reinterpret_cast<const uint4*>(smem) + ty * SW_ + tx + (ND - 1);
        const int jmax = xb - kCW / 2;
#pragma unroll
        for (int d0 = 0; d0 < ND; d0 += 8) {
            if (d0 < nd) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int d = d0 + u;
                    const uint4 rw = sb[-d];
                    const int cnt = __popc(lw.x ^ rw.x) + __popc(lw.y ^ rw.y) + __popc(lw.z ^ rw.z) + __popc(lw.w ^ rw.w);
                    c[d] = (d <= jmax) ? (float)cnt : kSentinel;
                }
            }
        }
    } else if (M == 1) {
        const double2 tl = a.ncl[pl];
        const double Al = tl.x, Cl = tl.y;
        const bool lfin = isfinite(Cl);
        const double2* st = reinterpret_cast<const double2*>(smem) + ty * SW_ + tx + (ND - 1);
        const uint8_t* si = smem + TR * SW_ * 16 + ty * (SW_ + 2) + tx + (ND - 1);     // window top-left of d = 0
        unsigned lv[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) lv[k] = a.l[(yb - 1 + k / 3) * W + xb - 1 + k % 3];
        const int jmax = xb - kNW / 2;
        // sliding 3x3 window of the right image: the window of step d covers relative columns -d .. -d+2; relative
        // column q lives in slot q mod 3, so a step loads only its new leftmost column (d is a constant after unrolling)
        unsigned win[3][3];
#pragma unroll
        for (int wh = 0; wh < 3; ++wh) { win[wh][1] = si[wh * (SW_ + 2) + 1]; win[wh][2] = si[wh * (SW_ + 2) + 2]; }
#pragma unroll
        for (int d0 = 0; d0 < ND; d0 += 8) {
            if (d0 < nd) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int d = d0 + u;
                    const int s0 = (3 - d % 3) % 3;
#pragma unroll
                    for (int wh = 0; wh < 3; ++wh) win[wh][s0] = si[wh * (SW_ + 2) - d];
                    unsigned LR = 0;
#pragma unroll
                    for (int k = 0; k < 9; ++k) LR += lv[k] * win[k / 3][(s0 + k % 3) % 3];
                    const double2 tr = st[-d];
                    float v;
                    if (lfin && isfinite(tr.y)) {
                        const double num = 9.0 * (double)LR - Al * tr.x;
                        double t = -num;
                        t = t * Cl;
                        t = t * tr.y;
                        v = (float)t;
                    } else {
                        v = 1.f;
                    }
                    c[d] = (d <= jmax) ? v : kSentinel;
                }
            }
        }
    } else if (M == 2) {
#pragma unroll
        for (int d0 = 0; d0 < ND; d0 += 8) {
            if (d0 < nd) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    c[d0 + u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(park, pix4, (unsigned)(d0 + u) * plane4, 0));
            }
        }
    } else {
        zsad_costs<ND, TR, 0, ND>(a, smem, tx, ty, xb, yb, nd, c);
    }
}

template <int M, int ND>
__device__ __forceinline__ void features_px(const FastArgs& a, unsigned char* smem) {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int xc0 = blockIdx.x * 64, yc0 = blockIdx.y * 4;
    const int xb0 = xc0 + a.bw, yb0 = yc0 + a.bh;
    const int nd = a.nd;

    stage_right<M, ND, 4>(a, smem, xb0, yb0, threadIdx.x, 256);
    if (M != 2) __syncthreads();

    const int x = xc0 + tx, y = yc0 + ty;
    if (x >= a.Wc || y >= a.Hc) return;
    const int xb = x + a.bw, yb = y + a.bh;
    const size_t plane = (size_t)a.Hc * a.Wc;
    // One buffer descriptor per output channel: a store is {descriptor, per-lane byte offset of the pixel, SCALAR byte
    // offset of the disparity plane} -- no per-store 64-bit address arithmetic on the vector unit.
    const unsigned pix4 = (unsigned)(y * a.Wc + x) * 4u;
    const unsigned plane4 = (unsigned)plane * 4u;
    const unsigned chan_bytes = (unsigned)nd * plane4;     // <= 4 GB checked on the host
    const __amdgpu_buffer_rsrc_t o_cost = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)M * nd * plane, 0, (int)chan_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t o_aml = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)(4 + M) * nd * plane, 0, (int)chan_bytes, 0x00020000);

    float c[ND];
    float m = kSentinel;
    // ---- pass 1: raw costs (the Sobel-SAD costs were parked in this matcher's likelihood channel by sadsob_band_kernel)
    raw_costs<M, ND, 4>(a, smem, tx, ty, xb, yb, nd, o_aml, pix4, plane4, c);

    // normalised cost channel + min
#pragma unroll
    for (int d0 = 0; d0 < ND; d0 += 8) {
        if (d0 < nd) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int d = d0 + u;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, norm_cost(M, c[d])), o_cost, pix4, (unsigned)d * plane4, 0);
                if (c[d] < m) m = c[d];
            }
        }
    }
    // ---- pass 2: likelihood numerators, denominator accumulated in d order
    const float kexp = a.kexp[M];
    float den = 0.f;
#pragma unroll
    for (int d0 = 0; d0 < ND; d0 += 8) {
        if (d0 < nd) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int d = d0 + u;
                const float e = aml_e(c[d], m, kexp);
                den += e;
                c[d] = e;
            }
        }
    }
    // ---- pass 3
    const bool dead = (m == kSentinel);                    // all-sentinel row -> zeros (featextract.cpp:452)
    const float rden = 1.f / den;                          // den >= 1: the minimum contributes expf(0)
#pragma unroll
    for (int d0 = 0; d0 < ND; d0 += 8) {
        if (d0 < nd) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int d = d0 + u;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dead ? 0.f : div_rn(c[d], den, rden)), o_aml, pix4, (unsigned)d * plane4, 0);
            }
        }
    }
}

template <int ND> constexpr size_t features_smem_bytes() {
    constexpr size_t SW_ = 64 + ND - 1;
    constexpr size_t m0 = 4 * SW_ * 16, m1 = 4 * SW_ * 16 + 6 * (SW_ + 2), m3 = 4 * SW_ * 4 + 8 * (SW_ + 4) * 4;
    return (m0 > m1 ? (m0 > m3 ? m0 : m3) : (m1 > m3 ? m1 : m3)) + 16;
}

// One kernel for the four matchers at three waves per SIMD (blockIdx.z + zbase: ZSAD, NCC, census, Sobel-SAD): ZSAD keeps 96
// costs + 25 left terms + a 5x5 sliding window in registers (155 with the per-step sched_barrier), the others ~120; the VALU-bound
// ZSAD waves share the CUs with the store-bound ones (135 us vs 84 + 51 us as two launches at 4 / 2 waves per SIMD)
template <int ND>
__global__ __launch_bounds__(256, 3) void features4_kernel(FastArgs a, int zbase) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[features_smem_bytes<ND>()];
    switch (blockIdx.z + zbase) {
    case 0: features_px<3, ND>(a, smem); break;
    case 1: features_px<1, ND>(a, smem); break;
    case 2: features_px<0, ND>(a, smem); break;
    default: features_px<2, ND>(a, smem); break;
    }
}
// ------------------------------------------------------------------------------------------------ channels-last features
// features_cl_kernel writes the volume as [D'][H'][W'][8] -- one 32-byte voxel record per (d, y, x), the first conv layer's
// own input layout -- so the 802 MB NCDHW -> NDHWC pass between the volume build and the aggregator does not exist on this
// path.  A workgroup owns a 64-pixel row segment and ALL FOUR matchers: wave w computes matcher (w + block) & 3 for the 64
// pixels with the D' raw costs in registers exactly as features_px does (same code: stage_right / raw_costs), then the four
// waves meet per group of eight disparities in an LDS tile [8 d][8 ch][64 px] and the 256 threads write it out as whole
// voxels: every store instruction of a wave covers 1 KB of contiguous output.  Both channels of a matcher must exist at the
// same time for that, so the likelihood numerator exp(-(c_d - m)^2 / sigma) is evaluated twice (once for the denominator,
// once for the value) instead of being kept in the cost's register -- same bits, four more instructions per element (common.h).
constexpr int kClPitch = 72;       // floats per LDS tile row: 64 pixels + 8, so that the two half-voxel readers of a pixel (rows
                                   // 4 apart: 4 * 72 = 288 = 32 mod 64) fall on disjoint bank halves
template <int ND> constexpr size_t features_cl_strip_bytes() {
    constexpr size_t m0 = strip_bytes<0, ND, 1>(), m1 = strip_bytes<1, ND, 1>(), m3 = strip_bytes<3, ND, 1>();
    return ((m0 > m1 ? (m0 > m3 ? m0 : m3) : (m1 > m3 ? m1 : m3)) + 15) & ~(size_t)15;
}

// ZSPLIT (ND = 96): the ZSAD wave's raw-cost pass -- 25 window terms x 3 dependent fp32 operations per disparity -- is twice any
// other matcher's and the other three waves wait for it at the first tile barrier.  The Sobel-SAD wave, whose own pass 1 is 96
// loads of parked costs, first forms the ZSAD costs of the UPPER half of the disparity range from the ZSAD wave's strips
// (`zstrip`) and hands them over through the second tile buffer, which no wave writes before the first tile barrier; the ZSAD
// wave forms the lower half.  Every cost is the same chain of operations whichever wave runs it: bit-identical output.
template <int M, int ND>
__device__ __forceinline__ void features_cl_wave(const FastArgs& a, unsigned char* strip, float* tbuf, const float* park_base,
                                                 const unsigned char* zstrip) {
    constexpr bool ZSPLIT = ND == 96;
    const int tid = threadIdx.x, lane = tid & 63;
    const int xc0 = blockIdx.x * 64, y = blockIdx.y;
    const int nd = a.nd;
    stage_right<M, ND, 1>(a, strip, xc0 + a.bw, y + a.bh, lane, 64);
    __syncthreads();
    const int x = min(xc0 + lane, a.Wc - 1);               // lanes past the row end recompute its last pixel (never stored)
    const size_t plane = (size_t)a.Hc * a.Wc;
    const unsigned pix4 = (unsigned)(y * a.Wc + x) * 4u;
    const unsigned plane4 = (unsigned)plane * 4u;
    const __amdgpu_buffer_rsrc_t park = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(park_base), 0, (int)((unsigned)nd * plane4), 0x00020000);
    const __amdgpu_buffer_rsrc_t o_all = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)((unsigned)nd * plane4 * 8u), 0x00020000);

    float c[ND];
    float* const hand = tbuf + 64 * kClPitch;              // the second tile buffer: [ND / 2][64] floats of hand-over
    if constexpr (ZSPLIT && M == 3) {
        zsad_costs<ND, 1, 0, ND / 2>(a, strip, x - xc0, 0, x + a.bw, y + a.bh, nd, c);
    } else if constexpr (ZSPLIT && M == 2) {
        zsad_costs<ND, 1, ND / 2, ND>(a, zstrip, x - xc0, 0, x + a.bw, y + a.bh, nd, c);
#pragma unroll
        for (int d = ND / 2; d < ND; ++d)
            if (d < nd) hand[(d - ND / 2) * 64 + lane] = c[d];
        raw_costs<M, ND, 1>(a, strip, x - xc0, 0, x + a.bw, y + a.bh, nd, park, pix4, plane4, c);
    } else {
        raw_costs<M, ND, 1>(a, strip, x - xc0, 0, x + a.bw, y + a.bh, nd, park, pix4, plane4, c);
    }
    if constexpr (ZSPLIT) {
        __syncthreads();                                   // the upper half of the ZSAD costs is in `hand`
        if constexpr (M == 3) {
#pragma unroll
            for (int d = ND / 2; d < ND; ++d)
                if (d < nd) c[d] = hand[(d - ND / 2) * 64 + lane];
        }
    }
    float m = kSentinel;
#pragma unroll
    for (int d = 0; d < ND; ++d)
        if (d < nd && c[d] < m) m = c[d];
    // ---- pass 2: the denominator, accumulated in d order (featextract.cpp:444-447)
    const float kexp = a.kexp[M];
    float den = 0.f;
#pragma unroll
    for (int d0 = 0; d0 < ND; d0 += 8) {
        if (d0 < nd) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                den += aml_e(c[d0 + u], m, kexp);
                if (u & 1) __builtin_amdgcn_sched_barrier(0);      // two expf in flight, not eight: all ND costs stay live here
            }
        }
    }
    const bool dead = (m == kSentinel);                    // all-sentinel row -> zeros (featextract.cpp:452)
    const float rden = 1.f / den;                          // den >= 1: the minimum contributes expf(0)
    // ---- pass 3: eight disparities at a time through the LDS tile, written out as whole voxels
    // store role: float4 number f = tid + 256 k of the tile's 1024 (8 d x 64 px x 2 halves): d = f >> 7, px = (f & 127) >> 1
    unsigned voff[4];
    int srow[4];
    int tq = tid;
    float m3 = m;
    // Opaque copies: (i) the store role's coordinates are computed BEHIND pass 2 -- during the raw-cost pass (ZSAD: 155
    // registers) they would be spilled; (ii) pass 3 recomputes (c_d - m)^2 / sigma from its own copy of m -- otherwise the
    // compiler keeps pass 2's ND quotients alive next to the ND costs (192 registers + spills) instead of re-deriving them.
    asm volatile("" : "+v"(tq), "+v"(den), "+v"(m3));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int f = tq + 256 * k, u = f >> 7, px = (f & 127) >> 1, half = f & 1;
        srow[k] = (u * 8 + half * 4) * kClPitch + px;
        voff[k] = xc0 + px < a.Wc ? (unsigned)(((size_t)u * plane + (size_t)y * a.Wc + xc0 + px) * 32u + half * 16u) : 0xffffffffu;
    }
#pragma unroll
    for (int d0 = 0; d0 < ND; d0 += 8) {
        if (d0 < nd) {
            float* buf = tbuf + ((d0 >> 3) & 1) * (64 * kClPitch);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int d = d0 + u;
                const float e = aml_e(c[d], m3, kexp);
                buf[(u * 8 + M) * kClPitch + lane] = norm_cost(M, c[d]);
                buf[(u * 8 + 4 + M) * kClPitch + lane] = dead ? 0.f : div_rn(e, den, rden);
                if (u & 1) __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();                               // (the other tile buffer is free again: its readers passed this barrier)
            const unsigned dbase = (unsigned)d0 * plane4 * 8u;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = buf[srow[k] + j * kClPitch];
                const unsigned off = voff[k] == 0xffffffffu ? 0xffffffffu : voff[k] + dbase;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), o_all, off, 0, 0);
            }
        }
    }
}

template <int ND>
__global__ __launch_bounds__(256, 3) void features_cl_kernel(FastArgs a, const float* park) {
    constexpr size_t SB = features_cl_strip_bytes<ND>();
    __shared__ __attribute__((aligned(16))) unsigned char strips[4 * SB];
    __shared__ __attribute__((aligned(16))) float tbuf[2 * 64 * kClPitch];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char* strip = strips + wave * SB;
    const unsigned char* zstrip = strips + ((0 - blockIdx.x - blockIdx.y) & 3) * SB;      // the strips of this block's ZSAD wave
    // the long-running ZSAD wave rotates over the four SIMDs from block to block
    switch ((wave + blockIdx.x + blockIdx.y) & 3) {
    case 0: features_cl_wave<3, ND>(a, strip, tbuf, park, zstrip); break;
    case 1: features_cl_wave<1, ND>(a, strip, tbuf, park, zstrip); break;
    case 2: features_cl_wave<0, ND>(a, strip, tbuf, park, zstrip); break;
    default: features_cl_wave<2, ND>(a, strip, tbuf, park, zstrip); break;
    }
}

// ------------------------------------------------------------------------------------------------ host
static int band_ls(int Wb) {
    int LS = Wb + 1;
    while (LS % 8 != 4) ++LS;
    return LS;
}

bool volume_fast_supported(const msnet_volume_params& p, int Hb, int Wb, int nd) {
    if (p.censw != kCW || p.nccw != kNW || p.sobelw != kSW || p.sadw != kZW) return false;
    if (p.border_h < 6 || p.border_w < 6) return false;    // every window of a cropped pixel is inside the image
    if (nd % 8 != 0 || nd > 96) return false;
    if ((size_t)nd * (Hb - 2 * p.border_h) * (Wb - 2 * p.border_w) * 4 > 0xfffffff0u) return false;   // one channel per buffer descriptor
    if (Wb + 8 > 2500) return false;                        // LDS band of the Sobel-SAD kernel (16 rows x (Wb + 8) floats)
    if (cdiv(Hb - 2 * p.border_h, kBandRMin) > kMaxBands) return false;   // band table of the Sobel-SAD kernels
    if (Hb > 8000) return false;                            // exact vertical sums: Hb * 2040 < 2^24 (sadsob_bandsum_kernel)
    // slot 0 of sadsob_bandsum_kernel sums the border_h - 3 image rows above band 0 with at most R terms (R = 11 for the wide
    // images that need 16-row bands, else 27): a taller top border takes the generic path
    const bool wide = (size_t)(27 + kSW) * band_ls(Wb) * sizeof(float) > 160 * 1024;
    if (p.border_h - kSW / 2 - 1 > (wide ? 11 : 27)) return false;
    return true;
}

// bytes of workspace the fast path uses: census bits, NCC tables, four float images, Sobel-SAD checkpoints
size_t volume_fast_workspace_bytes(int Hb, int Wb, int nd) {
    const size_t img = (size_t)Hb * Wb;
    const int Hc = Hb > 12 ? Hb - 12 : 1;                  // at least border 6; more border = fewer bands
    // + the parked Sobel-SAD raw costs [nd][Hc][Wc] of the channels-last build (the NCDHW build parks them in its output)
    return img * (2 * 16 + 2 * 16 + 4 * 4) + (size_t)nd * cdiv(Hc, kBandRMin) * band_ls(Wb) * sizeof(float) + 256 +
           (size_t)nd * Hc * (Wb > 12 ? Wb - 12 : 1) * sizeof(float) + 256;
}

// Second stream of the build: the Sobel-SAD kernels (one wave per workgroup busy most of the time, LDS-capacity-bound) run
// beside the feature kernel of the other three matchers (VALU- and store-bound) instead of in front of it.  One helper
// stream and two events per (device, caller stream), created on first use: builds issued on DIFFERENT streams (two host
// threads, two VolumeBuilders) never share an event; builds on one stream are ordered by that stream.  fork / join are
// ordinary event waits, so the build stays asynchronous on the caller's stream and capturable.
struct VolAux {
    hipStream_t s = nullptr; hipEvent_t fork = nullptr, join = nullptr; bool ok = false; unsigned long long used = 0;
    VolAux() = default;
    VolAux(const VolAux&) = delete;
    VolAux& operator=(const VolAux&) = delete;
    ~VolAux() {                                             // HIP lets work queued on a destroyed stream finish first
        if (fork) (void)hipEventDestroy(fork);
        if (join) (void)hipEventDestroy(join);
        if (s) (void)hipStreamDestroy(s);
    }
};
constexpr size_t kMaxVolAux = 16;      // helper streams kept alive at once (callers that build on many short-lived streams)
// Entries are handed out as shared_ptr: an entry that the LRU eviction drops from the table while another thread's launch
// sequence still uses it stays alive until that sequence returns (the references handed out before were freed under it).
static std::shared_ptr<VolAux> vol_aux(hipStream_t caller) {
    static std::map<std::pair<int, hipStream_t>, std::shared_ptr<VolAux>> aux;
    static std::mutex mu;
    static unsigned long long tick = 0;
    std::lock_guard<std::mutex> lk(mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const auto key = std::make_pair(dev, caller);
    auto it = aux.find(key);
    if (it == aux.end()) {
        if (aux.size() >= kMaxVolAux) {
            // least recently used entry leaves the table; a caller stream whose entry was evicted gets a new one on its next build
            auto old = aux.begin();
            for (auto j = aux.begin(); j != aux.end(); ++j) if (j->second->used < old->second->used) old = j;
            aux.erase(old);
        }
        auto x = std::make_shared<VolAux>();
        x->ok = hipStreamCreateWithFlags(&x->s, hipStreamNonBlocking) == hipSuccess &&
                hipEventCreateWithFlags(&x->fork, hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&x->join, hipEventDisableTiming) == hipSuccess;
        it = aux.emplace(key, std::move(x)).first;
    }
    it->second->used = ++tick;
    return it->second;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device attribute of a kernel: set once per (device, kernel), under a lock
template <class K>
static void allow_big_lds(K kernel) {
    static std::mutex mu;
    static unsigned long long done = 0;                     // bit = device ordinal (this static exists once per kernel type)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); return; }
    std::lock_guard<std::mutex> lk(mu);
    if (!(done >> dev & 1ull)) {
        (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        done |= 1ull << dev;
    }
}

// channels_last: out is [nd][Hc][Wc][8] (features_cl_kernel) instead of [8][nd][Hc][Wc]
int volume_fast_launch(const uint8_t* l, const uint8_t* r, int Hb, int Wb, int nd, const msnet_volume_params& p, void* workspace,
                       float* out, hipStream_t s, bool channels_last) {
    const size_t img = (size_t)Hb * Wb;
    FastArgs a{};
    a.l = l; a.r = r;
    unsigned char* w = (unsigned char*)workspace;          // carve (all 16-byte aligned): bits 2*img*16 | tables 2*img*16 | 4 float images | checkpoints
    a.lb = (uint4*)w; a.rb = a.lb + img;
    a.ncl = (double2*)(a.rb + img); a.ncr = a.ncl + img;
    a.ml = (float*)(a.ncr + img); a.mr = a.ml + img; a.sobl = a.mr + img; a.sobr = a.sobl + img;
    float* ck = a.sobr + img;
    const int Hc_ws = Hb > 12 ? Hb - 12 : 1;               // the carve below uses the workspace function's own bounds
    float* park_ws = (float*)(((uintptr_t)(ck + (size_t)nd * cdiv(Hc_ws, kBandRMin) * band_ls(Wb)) + 255) & ~(uintptr_t)255);
    a.out = out;
    a.kexp[0] = aml_scale(p.cens_sigma); a.kexp[1] = aml_scale(p.ncc_sigma); a.kexp[2] = aml_scale(p.sad_sigma); a.kexp[3] = aml_scale(p.sad_sigma);
    a.Hb = Hb; a.Wb = Wb; a.nd = nd; a.bh = p.border_h; a.bw = p.border_w;
    a.Hc = Hb - 2 * p.border_h; a.Wc = Wb - 2 * p.border_w;
    const size_t plane = (size_t)a.Hc * a.Wc;
    const dim3 gpix(cdiv(a.Wc, 64), cdiv(a.Hc, 4), 1);
#ifdef EXP_VOLUME_KNOBS      // diagnostic builds only: the shipped library takes nothing from the environment here
    static const int band_skip = [] { const char* e = getenv("MSNET_BAND_SKIP"); return e ? atoi(e) : 0; }();     // skip phases (wrong results)
    static const int band_cfg = [] { const char* e = getenv("MSNET_BAND_CFG"); return e ? atoi(e) : 0; }();       // band height / threads
    static const bool want_overlap = [] { const char* e = getenv("MSNET_VOL_STREAMS"); return !(e && e[0] == '0'); }();
#else
    constexpr int band_skip = 0, band_cfg = 0;
    constexpr bool want_overlap = true;
#endif
    if (channels_last && (size_t)nd * plane * 32 > 0xfffffff0u)
        return fail("msnet_build_volume_ndhwc: the volume exceeds the 4 GB buffer-descriptor range");
    // (the channels-last feature launch needs all four matchers at once: nothing to overlap the Sobel-SAD kernels with, no helper stream)
    static const std::shared_ptr<VolAux> none = std::make_shared<VolAux>();
    const std::shared_ptr<VolAux> aux_ref = (channels_last || !want_overlap) ? none : vol_aux(s);     // alive until this call returns
    VolAux& aux = *aux_ref;
    const bool overlap = want_overlap && aux.ok && !channels_last;
    hipStream_t sb = overlap ? aux.s : s;                  // stream of the Sobel-SAD kernels

    // the whole build on the caller's stream: timed from the start of its FIRST kernel to the end of its LAST one, both on the
    // caller's stream, through events attached to those two kernels' dispatch packets (common.h: no barrier packets in the stream)
    LaunchScope whole("vol_build", s, 0, 4.0 * 8.0 * nd * (double)plane + 2.0 * img, true);
    hipEvent_t ev_first = nullptr, ev_last = nullptr;
    const bool timed = whole.events(&ev_first, &ev_last);
    whole.launched = false;        // the stop event rides on the LAST kernel: an early return before it drops the row (common.h)
    {
        LaunchScope ls("volk_prep", s, 0, 2.0 * img + 72.0 * img);
        if (timed) hipExtLaunchKernelGGL(vprep_kernel, dim3(cdiv(Wb, 64), cdiv(Hb, 4), 3), dim3(256), 0, s, ev_first, nullptr, 0, a);
        else hipLaunchKernelGGL(vprep_kernel, dim3(cdiv(Wb, 64), cdiv(Hb, 4), 3), dim3(256), 0, s, a);
    }
    if (overlap) {
        if (hipEventRecord(aux.fork, s) != hipSuccess || hipStreamWaitEvent(aux.s, aux.fork, 0) != hipSuccess)
            return fail("msnet_build_volume: stream fork failed");
    }
    {
        const int LS = band_ls(Wb);
        float* park = channels_last ? park_ws : out + (size_t)6 * nd * plane;     // NCDHW: channel 6 = likelihood of the Sobel-SAD cost
        LaunchScope ls("volk_sadsob", sb, 0, 4.0 * nd * (double)plane);
        auto launch = [&](auto rc, auto ntc) -> int {
            constexpr int R = decltype(rc)::value, NT = decltype(ntc)::value;
            const int nbands = cdiv(a.Hc, R);
            if (nbands > kMaxBands) return fail("msnet_build_volume: %d Sobel-SAD bands (max %d)", nbands, kMaxBands);
            const size_t lds = (size_t)(R + kSW) * LS * sizeof(float);
            if (lds > 160 * 1024) return fail("msnet_build_volume: image width %d too large for the Sobel-SAD band", Wb);
            if (lds > 64 * 1024) allow_big_lds(sadsob_band_kernel<R, NT>);
            hipLaunchKernelGGL(sadsob_bandsum_kernel<R>, dim3(cdiv(LS, 256), nbands, nd), dim3(256), 0, sb, a, ck, LS, nbands);
            hipLaunchKernelGGL((sadsob_band_kernel<R, NT>), dim3(nd * nbands), dim3(NT), lds, sb, a, ck, park, LS, nbands, band_skip);
            return 0;
        };
        int rc = 0;
        const bool wide = (size_t)(27 + kSW) * LS * sizeof(float) > 160 * 1024;     // full-resolution KITTI widths: 16-row bands
        // the channels-last build runs these kernels ALONE (nothing to overlap them with): 35-row bands -- 8 x 96 = 768 workgroups
        // at config #2 instead of 1056, three full rounds of one per CU -- measured 45 vs 55 us (profiles/r04_band_cfg.txt)
        const bool tall = channels_last && band_cfg == 0 && (size_t)(35 + kSW) * LS * sizeof(float) <= 160 * 1024 &&
                          a.bh - kSW / 2 - 1 <= 35;
        if (wide) rc = launch(std::integral_constant<int, 11>{}, std::integral_constant<int, 256>{});
        else if (tall) rc = launch(std::integral_constant<int, 35>{}, std::integral_constant<int, 512>{});
        else if (band_cfg == 1) rc = launch(std::integral_constant<int, 27>{}, std::integral_constant<int, 256>{});
        else if (band_cfg == 2) rc = launch(std::integral_constant<int, 11>{}, std::integral_constant<int, 256>{});
        else if (band_cfg == 3) rc = launch(std::integral_constant<int, 59>{}, std::integral_constant<int, 512>{});
        else if (band_cfg == 4) rc = launch(std::integral_constant<int, 19>{}, std::integral_constant<int, 512>{});
        else if (band_cfg == 5) rc = launch(std::integral_constant<int, 19>{}, std::integral_constant<int, 256>{});
        else if (band_cfg == 6) rc = launch(std::integral_constant<int, 35>{}, std::integral_constant<int, 512>{});
        else rc = launch(std::integral_constant<int, 27>{}, std::integral_constant<int, 512>{});
        if (rc) return rc;
    }
    if (overlap && hipEventRecord(aux.join, aux.s) != hipSuccess) return fail("msnet_build_volume: stream join failed");
    // (last = true: the build's last kernel carries the whole-build stop event)
    auto features = [&](int zbase, int nz, bool last) {
        const dim3 g(gpix.x, gpix.y, nz);
        hipEvent_t stop = (last && timed) ? ev_last : nullptr;
        if (stop) whole.launched = true;
        if (nd <= 32) { if (stop) hipExtLaunchKernelGGL(features4_kernel<32>, g, dim3(256), 0, s, nullptr, stop, 0, a, zbase);
                        else hipLaunchKernelGGL(features4_kernel<32>, g, dim3(256), 0, s, a, zbase); }
        else          { if (stop) hipExtLaunchKernelGGL(features4_kernel<96>, g, dim3(256), 0, s, nullptr, stop, 0, a, zbase);
                        else hipLaunchKernelGGL(features4_kernel<96>, g, dim3(256), 0, s, a, zbase); }
    };
    if (channels_last) {
        LaunchScope ls("volk_features", s, 0, 4.0 * 8.0 * nd * (double)plane);
        const dim3 g(gpix.x, a.Hc, 1);
        hipEvent_t stop = timed ? ev_last : nullptr;
        if (stop) whole.launched = true;
        if (nd <= 32) { if (stop) hipExtLaunchKernelGGL(features_cl_kernel<32>, g, dim3(256), 0, s, nullptr, stop, 0, a, (const float*)park_ws);
                        else hipLaunchKernelGGL(features_cl_kernel<32>, g, dim3(256), 0, s, a, park_ws); }
        else          { if (stop) hipExtLaunchKernelGGL(features_cl_kernel<96>, g, dim3(256), 0, s, nullptr, stop, 0, a, (const float*)park_ws);
                        else hipLaunchKernelGGL(features_cl_kernel<96>, g, dim3(256), 0, s, a, park_ws); }
    } else if (overlap) {
        {
            LaunchScope ls("volk_features", s, 0, 4.0 * 6.0 * nd * (double)plane);
            features(0, 3, false);                          // ZSAD, NCC, census: no dependence on the Sobel-SAD stream
        }
        if (hipStreamWaitEvent(s, aux.join, 0) != hipSuccess) return fail("msnet_build_volume: stream join failed");
        LaunchScope ls("volk_features_sobel", s, 0, 4.0 * 2.0 * nd * (double)plane);
        features(3, 1, true);
    } else {
        LaunchScope ls("volk_features", s, 0, 4.0 * 8.0 * nd * (double)plane);
        features(0, 4, true);
    }
    return check_launch("msnet_build_volume");
}

}  // namespace msnet

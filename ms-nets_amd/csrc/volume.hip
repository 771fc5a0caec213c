// Matching-space cost volume on the GPU: the four hand-crafted matchers, the per-pixel likelihood (AML)
// features and the 8-channel assembly.  Replaces the CPU path
//   /root/reference/src/cpp/matchers/matchers.cpp      census :232-353, nccNister :47-228, sadsob :356-438,
//                                                      zsad :442-512, sobel :515-554
//   /root/reference/src/cpp/featextract/featextract.cpp swap_axes :49-76, extract_aml_testing :415-462
//   /root/reference/src/dataloader/cbmv_generator.py    get_costs :27-79, extract_features_left :258-308
//
// Built with -ffp-contract=off: zsad / sadsob / the normalisation are float32 arithmetic whose operation
// ORDER is part of the reference's result, so nothing here may be fused into an FMA or
// re-associated.  Integer paths (census, sobel, the NCC window sums) are exact by construction.
// Entries the reference never writes keep its fill value RAND_MAX -> 2^31 (kSentinel).
#include <stdlib.h>

#include "common.h"

namespace msnet {

// ---------------------------------------------------------------- census -------------------------------
// Bit b = wh*wsize + ww of pixel (y,x) is [center < window(wh,ww)]; the reference pads the 121 bits to 128
// with always-equal lanes, so cost = popcount(L ^ R_shift) over the wsize^2 real bits.
constexpr int kCensusWords = 8;   // up to 16x16 windows

__global__ void census_transform_kernel(const uint8_t* __restrict__ img, uint32_t* __restrict__ bits, int H, int W,
                                        int ws, int nwords) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const int wc = ws / 2;
    const int i = y - wc, j = x - wc;
    uint32_t wd[kCensusWords];
#pragma unroll
    for (int k = 0; k < kCensusWords; ++k) wd[k] = 0;
    if (i >= 0 && j >= 0 && i < H - ws && j < W - ws) {
        const int c = img[y * W + x];
        int b = 0;
        for (int wh = 0; wh < ws; ++wh)
            for (int ww = 0; ww < ws; ++ww, ++b)
                if (c < (int)img[(i + wh) * W + j + ww]) {
#pragma unroll
                    for (int k = 0; k < kCensusWords; ++k)
                        if (k == (b >> 5)) wd[k] |= 1u << (b & 31);
                }
    }
    for (int k = 0; k < nwords; ++k) bits[((size_t)y * W + x) * nwords + k] = wd[k];
}

// DMAJOR = false: out[y][x][d] (reference layout of census);  true: out[d][y][x] (fused path).
template <bool DMAJOR>
__global__ void census_cost_kernel(const uint32_t* __restrict__ lb, const uint32_t* __restrict__ rb,
                                   float* __restrict__ out, int H, int W, int nd, int ws, int nwords) {
    const int wc = ws / 2;
    const size_t total = (size_t)H * W * nd;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        int d, x, y;
        if (DMAJOR) { x = o % W; y = (o / W) % H; d = (int)(o / ((size_t)W * H)); }
        else        { d = o % nd; x = (o / nd) % W; y = (int)(o / ((size_t)nd * W)); }
        const int i = y - wc, j = x - wc;
        float v = kSentinel;
        if (i >= 0 && j >= 0 && i < H - ws && j < W - ws && d <= j) {
            const uint32_t* a = lb + ((size_t)y * W + x) * nwords;
            const uint32_t* b = rb + ((size_t)y * W + x - d) * nwords;
            int cnt = 0;
            for (int k = 0; k < nwords; ++k) cnt += __popc(a[k] ^ b[k]);
            v = (float)cnt;
        }
        out[o] = v;
    }
}

// ---------------------------------------------------------------- sobel --------------------------------
__global__ void sobel_kernel(const uint8_t* __restrict__ img, float* __restrict__ out, int H, int W) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const int i = y - 1, j = x - 1;
    float v = 0.f;
    if (i >= 0 && j >= 0 && i < H - 3 && j < W - 3) {
        const uint8_t* p = img + i * W + j;
        const int s = -(int)p[0] + (int)p[2] - 2 * (int)p[W] + 2 * (int)p[W + 2] - (int)p[2 * W] + (int)p[2 * W + 2];
        v = (float)s;
    }
    out[y * W + x] = v;
}

// ---------------------------------------------------------------- NCC ----------------------------------
// All window sums are exact integers (the reference's u32/u64/double integral images are exact too), so
// direct summation gives the same values; only 1/sqrt and the two final multiplies round, in double, in
// the reference's order:  tmp = ((-(double)(n*lD - Al*Ar)) * Cl) * Cr;  cost = (float)tmp.
__global__ void ncc_kernel(const uint8_t* __restrict__ l, const uint8_t* __restrict__ r, float* __restrict__ out,
                           int H, int W, int nd, int ws) {
    const int wc = ws / 2, sq = ws * ws;
    const size_t total = (size_t)nd * H * W;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        const int x = o % W, y = (o / W) % H, d = (int)(o / ((size_t)W * H));
        const int i = y - wc, j = x - wc;
        float v = kSentinel;
        if (i >= 0 && j >= d && i < H - ws && j < W - ws) {
            unsigned long long Al = 0, Ar = 0, Bl = 0, Br = 0, LR = 0;
            for (int wh = 0; wh < ws; ++wh) {
                const uint8_t* lp = l + (i + wh) * W + j;
                const uint8_t* rp = r + (i + wh) * W + j - d;
                for (int ww = 0; ww < ws; ++ww) {
                    const unsigned a = lp[ww], b = rp[ww];
                    Al += a; Ar += b; Bl += a * a; Br += b * b; LR += a * b;
                }
            }
            const double Cl = 1.0 / sqrt((double)((unsigned long long)sq * Bl) - (double)Al * (double)Al);
            const double Cr = 1.0 / sqrt((double)((unsigned long long)sq * Br) - (double)Ar * (double)Ar);
            if (isfinite(Cl) && isfinite(Cr)) {
                const double num = (double)sq * (double)LR - (double)(Al * Ar);
                double t = -num;
                t = t * Cl;
                t = t * Cr;
                v = (float)t;
            } else {
                v = 1.f;
            }
        }
        out[o] = v;
    }
}

// ---------------------------------------------------------------- ZSAD ---------------------------------
__device__ __forceinline__ float window_mean(const uint8_t* __restrict__ p, int W, int ws) {
    float s = 0.f;   // sequential float sum of <= 256 bytes: exact
    for (int wh = 0; wh < ws; ++wh)
        for (int ww = 0; ww < ws; ++ww) s += (float)p[wh * W + ww];
    return s / (float)(ws * ws);
}

__global__ void zsad_kernel(const uint8_t* __restrict__ l, const uint8_t* __restrict__ r, float* __restrict__ out,
                            int H, int W, int nd, int ws) {
    const int wc = ws / 2;
    const size_t total = (size_t)nd * H * W;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        const int x = o % W, y = (o / W) % H, d = (int)(o / ((size_t)W * H));
        const int i = y - wc, j = x - wc;
        float v = kSentinel;
        if (i >= 0 && j >= d && i < H - ws && j < W - ws) {
            const uint8_t* lp = l + i * W + j;
            const uint8_t* rp = r + i * W + j - d;
            const float ml = window_mean(lp, W, ws), mr = window_mean(rp, W, ws);
            float acc = 0.f;
            for (int wh = 0; wh < ws; ++wh)
                for (int ww = 0; ww < ws; ++ww) {
                    float t = (float)lp[wh * W + ww] - ml;
                    t = t - (float)rp[wh * W + ww];
                    t = t + mr;
                    acc = acc + fabsf(t);
                }
            v = acc;
        }
        out[o] = v;
    }
}

// ---------------------------------------------------------------- SAD of Sobel -------------------------
// The reference builds, per disparity, a FLOAT32 integral image of |SL - SR_shift| by a sequential vertical
// pass followed by a sequential horizontal pass; values exceed 2^24 so the rounding depends on that order
// and it is reproduced literally: one thread owns one column (pass V) or one row (pass H) and adds in the
// same sequence.  ws layout: [nd][H+1][W+1].
__global__ void sadsob_vertical_kernel(const float* __restrict__ sl, const float* __restrict__ sr, float* __restrict__ ws,
                                       int H, int W, int nd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;   // integral column 0..W
    const int d = blockIdx.y;
    if (c > W) return;
    float* col = ws + (size_t)d * (H + 1) * (W + 1) + c;
    col[0] = 0.f;
    const int j = c - 1;                                    // image column
    if (j < d) {                                            // columns the reference leaves at zero
        for (int i = 1; i <= H; ++i) col[(size_t)i * (W + 1)] = 0.f;
        return;
    }
    // the running sum is a strictly sequential float32 chain (the reference's order); only the loads are batched so
    // that their latency is paid once per 8 rows instead of once per row
    float run = 0.f;
    int i = 1;
    for (; i + 7 <= H; i += 8) {
        float a[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = fabsf(sl[(i - 1 + k) * W + j] - sr[(i - 1 + k) * W + j - d]);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            run = a[k] + run;
            col[(size_t)(i + k) * (W + 1)] = run;
        }
    }
    for (; i <= H; ++i) {
        const float a = fabsf(sl[(i - 1) * W + j] - sr[(i - 1) * W + j - d]);
        run = a + run;
        col[(size_t)i * (W + 1)] = run;
    }
}

// One wave scans 64 rows; column tiles of 64 go through LDS so global traffic stays coalesced.
__global__ __launch_bounds__(64) void sadsob_horizontal_kernel(float* __restrict__ ws, int H, int W, int nd) {
    __shared__ float tile[64][65];
    const int lane = threadIdx.x;
    const int row0 = blockIdx.x * 64;           // integral rows 0..H
    const int d = blockIdx.y;
    float* base = ws + (size_t)d * (H + 1) * (W + 1);
    const int nrows = min(64, H + 1 - row0);
    float run = 0.f;                            // slice[row][d] == 0
    for (int c0 = d + 1; c0 <= W; c0 += 64) {
        const int ncols = min(64, W + 1 - c0);
        if (nrows == 64 && ncols == 64) {       // full tile: all 64 row loads in flight before the first LDS write
            float t[64];
#pragma unroll
            for (int rr = 0; rr < 64; ++rr) t[rr] = base[(size_t)(row0 + rr) * (W + 1) + c0 + lane];
#pragma unroll
            for (int rr = 0; rr < 64; ++rr) tile[rr][lane] = t[rr];
        } else {
            for (int rr = 0; rr < nrows; ++rr)
                if (lane < ncols) tile[rr][lane] = base[(size_t)(row0 + rr) * (W + 1) + c0 + lane];
        }
        __syncthreads();
        if (lane < nrows) {
            if (ncols == 64) {                  // row segment through registers: the add chain no longer waits on LDS per step
                float v[64];
#pragma unroll
                for (int k = 0; k < 64; ++k) v[k] = tile[lane][k];
#pragma unroll
                for (int k = 0; k < 64; ++k) { run = v[k] + run; v[k] = run; }
#pragma unroll
                for (int k = 0; k < 64; ++k) tile[lane][k] = v[k];
            } else {
                for (int k = 0; k < ncols; ++k) {
                    run = tile[lane][k] + run;
                    tile[lane][k] = run;
                }
            }
        }
        __syncthreads();
#pragma unroll 8
        for (int rr = 0; rr < nrows; ++rr)
            if (lane < ncols) base[(size_t)(row0 + rr) * (W + 1) + c0 + lane] = tile[rr][lane];
        __syncthreads();
    }
}

__global__ void sadsob_box_kernel(const float* __restrict__ ws, float* __restrict__ out, int H, int W, int nd, int wsz) {
    const int wc = wsz / 2;
    const size_t total = (size_t)nd * H * W;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        const int x = o % W, y = (o / W) % H, d = (int)(o / ((size_t)W * H));
        const int i = y - wc, j = x - wc;
        float v = kSentinel;
        if (i >= 0 && j >= d && i < H - wsz && j < W - wsz) {
            const float* s = ws + (size_t)d * (H + 1) * (W + 1);
            const float* t = s + (size_t)i * (W + 1);
            const float* b = s + (size_t)(i + wsz) * (W + 1);
            float r = b[j + wsz] - b[j];
            r = r - t[j + wsz];
            r = r + t[j];
            v = r;
        }
        out[o] = v;
    }
}

// ---------------------------------------------------------------- swap_axes ----------------------------
// [D][S] -> [S][D] (S = H*W) through a 64x64 LDS tile.
__global__ __launch_bounds__(256) void swap_axes_kernel(const float* __restrict__ in, float* __restrict__ out, int D, long S) {
    __shared__ float tile[64][65];
    const long s0 = (long)blockIdx.x * 64;
    const int d0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int k = ty; k < 64; k += 4)
        if (d0 + k < D && s0 + tx < S) tile[k][tx] = in[(size_t)(d0 + k) * S + s0 + tx];
    __syncthreads();
    for (int k = ty; k < 64; k += 4)
        if (s0 + k < S && d0 + tx < D) out[(size_t)(s0 + k) * D + d0 + tx] = tile[tx][k];
}

// ---------------------------------------------------------------- AML ----------------------------------
// extract_aml_testing: per row of D costs, m = min; den = sum_k expf(-((c_k-m)^2)/sigma) accumulated
// sequentially in float32; out_k = expf(-((c_k-m)^2)/sigma)/den, or 0 for an all-sentinel row.
__device__ __forceinline__ float aml_term(float c, float m, float sigma) {
    return aml_numerator(c, m, aml_scale(sigma));          // (the scale is loop-invariant at every call site: one double division per thread)
}

// vol/out [P][D]; 64 rows per wave staged through LDS (row stride D+1) so global accesses are coalesced.
__global__ __launch_bounds__(64) void aml_rows_kernel(const float* __restrict__ vol, float* __restrict__ out, long P, int D,
                                                      float sigma) {
    extern __shared__ float rows[];   // [64][D+1]
    const int lane = threadIdx.x;
    const long p0 = (long)blockIdx.x * 64;
    const int n = (int)((P - p0 < 64) ? (P - p0) : 64);
    const int LS = D + 1;
    const size_t base = (size_t)p0 * D;
    for (int k = lane; k < n * D; k += 64) rows[(k / D) * LS + k % D] = vol[base + k];
    __syncthreads();
    if (lane < n) {
        float* rp = rows + lane * LS;
        float m = kSentinel;
        for (int k = 0; k < D; ++k) if (rp[k] < m) m = rp[k];
        float den = 0.f;
        for (int k = 0; k < D; ++k) den += aml_term(rp[k], m, sigma);
        for (int k = 0; k < D; ++k) rp[k] = (m == kSentinel) ? 0.f : aml_term(rp[k], m, sigma) / den;
    }
    __syncthreads();
    for (int k = lane; k < n * D; k += 64) out[base + k] = rows[(k / D) * LS + k % D];
}

// extract_features_left for ONE matcher on the reference's row layout: vol [P][D] (P = H'*W' pixels) ->
//   out_cost[d][p] = normalised cost, out_aml[d][p] = likelihood  (i.e. already transposed to [D][H'][W']).
__device__ __forceinline__ float normalise_cost(int ch, float c);
__global__ __launch_bounds__(64) void features_rows_kernel(const float* __restrict__ vol, float* __restrict__ out_cost,
                                                           float* __restrict__ out_aml, long P, int D, float sigma, int ch) {
    extern __shared__ float rows[];   // [64][D+1]
    const int lane = threadIdx.x;
    const long p0 = (long)blockIdx.x * 64;
    const int n = (int)((P - p0 < 64) ? (P - p0) : 64);
    const int LS = D + 1;
    const size_t base = (size_t)p0 * D;
    for (int k = lane; k < n * D; k += 64) rows[(k / D) * LS + k % D] = vol[base + k];
    __syncthreads();
    if (lane < n) {
        const float* rp = rows + lane * LS;
        float m = kSentinel;
        for (int k = 0; k < D; ++k) if (rp[k] < m) m = rp[k];
        float den = 0.f;
        for (int k = 0; k < D; ++k) den += aml_term(rp[k], m, sigma);
        for (int k = 0; k < D; ++k) {
            const float c = rp[k];
            out_cost[(size_t)k * P + p0 + lane] = normalise_cost(ch, c);
            out_aml[(size_t)k * P + p0 + lane] = (m == kSentinel) ? 0.f : aml_term(c, m, sigma) / den;
        }
    }
}

// ---------------------------------------------------------------- assembly -----------------------------
// extract_features_left on d-major raw costs [nd][Hb][Wb]: crop the border, write
//   out[ch][d][y][x]    = normalised cost     (ch 0..3 = census, ncc, sobel-SAD, zsad)
//   out[4+ch][d][y][x]  = AML(raw cost, sigma_ch)
// Thread = one pixel, lanes = consecutive x, so every access is a coalesced 256-byte run per d.
struct AssembleArgs {
    const float* raw[4];
    float sigma[4];
    float* out;
    int Hb, Wb, nd, bh, bw, Hc, Wc;
};

__device__ __forceinline__ float normalise_cost(int ch, float c) {
    switch (ch) {
    case 0: return fminf(fmaxf(c, 0.f), 120.f) / 120.f;                  // cbmv_generator.py:283
    case 1: { float t = fminf(fmaxf(c, -1.f), 1.f); t = 1.f + t; return t / 2.f; }   // :285
    default: return fminf(fmaxf(c, 0.f), 8192.f) / 8192.f;               // :286-287
    }
}

__global__ __launch_bounds__(256) void assemble_kernel(AssembleArgs a) {
    const int ch = blockIdx.z;
    const int y = blockIdx.y;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= a.Wc) return;
    const size_t plane_in = (size_t)a.Hb * a.Wb;
    const float* src = a.raw[ch] + (size_t)(y + a.bh) * a.Wb + (x + a.bw);
    const size_t plane_out = (size_t)a.Hc * a.Wc;
    float* o_cost = a.out + ((size_t)ch * a.nd) * plane_out + (size_t)y * a.Wc + x;
    float* o_aml = a.out + ((size_t)(4 + ch) * a.nd) * plane_out + (size_t)y * a.Wc + x;
    const float sigma = a.sigma[ch];
    float m = kSentinel;
    for (int d = 0; d < a.nd; ++d) { const float c = src[d * plane_in]; if (c < m) m = c; }
    float den = 0.f;
    for (int d = 0; d < a.nd; ++d) den += aml_term(src[d * plane_in], m, sigma);
    for (int d = 0; d < a.nd; ++d) {
        const float c = src[d * plane_in];
        o_cost[d * plane_out] = normalise_cost(ch, c);
        o_aml[d * plane_out] = (m == kSentinel) ? 0.f : aml_term(c, m, sigma) / den;
    }
}

// ---------------------------------------------------------------- fused features -----------------------
// One kernel per matcher goes from the images (plus small per-pixel tables) straight to the two volume channels
// it owns: thread = one cropped pixel, lanes along x (256-byte coalesced rows per d); raw cost for every d, min,
// sequential likelihood sum, then the normalised cost and the likelihood.  The d-major raw-cost volumes over the
// bordered image (4 x 56 MB written, read three times, plus the crop) never exist; only sadsob keeps its float32
// integral image, whose sequential passes cannot be fused.  Arithmetic is the per-matcher kernels' arithmetic.
struct FusedArgs {
    const uint8_t* l; const uint8_t* r;
    const uint32_t* lb; const uint32_t* rb;      // census bit images
    const double* nal; const double* ncl; const double* nar; const double* ncr;   // NCC: window sum A and 1/sqrt term C
    const float* ml; const float* mr;            // ZSAD window means
    const float* integ;                          // sadsob integral images [nd][Hb+1][Wb+1]
    float* out;
    float sigma[4];
    int Hb, Wb, nd, bh, bw, Hc, Wc;
    int censw, nccw, sadw, sobelw, nwords;
};

// per-pixel tables: NCC (A, C) of both images and ZSAD means of both images, at window-centre coordinates
__device__ __forceinline__ void pixel_tables_px(const FusedArgs& a, int x, int y) {
    const int W = a.Wb, H = a.Hb;
    const size_t c = (size_t)y * W + x;
    {   // NCC
        const int ws = a.nccw, wc = ws / 2, i = y - wc, j = x - wc;
        double Al = 0, Cl = 0, Ar = 0, Cr = 0;
        if (i >= 0 && j >= 0 && i < H - ws && j < W - ws) {
            unsigned long long sl = 0, sr = 0, ql = 0, qr = 0;
            for (int wh = 0; wh < ws; ++wh)
                for (int ww = 0; ww < ws; ++ww) {
                    const unsigned p = a.l[(i + wh) * W + j + ww], q = a.r[(i + wh) * W + j + ww];
                    sl += p; sr += q; ql += p * p; qr += q * q;
                }
            const unsigned long long sq = (unsigned long long)(ws * ws);
            Al = (double)sl; Ar = (double)sr;
            Cl = 1.0 / sqrt((double)(sq * ql) - (double)sl * (double)sl);
            Cr = 1.0 / sqrt((double)(sq * qr) - (double)sr * (double)sr);
        }
        const_cast<double*>(a.nal)[c] = Al;
        const_cast<double*>(a.ncl)[c] = Cl; const_cast<double*>(a.nar)[c] = Ar; const_cast<double*>(a.ncr)[c] = Cr;
    }
    {   // ZSAD means
        const int ws = a.sadw, wc = ws / 2, i = y - wc, j = x - wc;
        float ml = 0.f, mr = 0.f;
        if (i >= 0 && j >= 0 && i < H - ws && j < W - ws) {
            ml = window_mean(a.l + i * W + j, W, ws);
            mr = window_mean(a.r + i * W + j, W, ws);
        }
        const_cast<float*>(a.ml)[c] = ml; const_cast<float*>(a.mr)[c] = mr;
    }
}

// Per-pixel cost generators.  WS > 0 fixes the window at compile time (the defaults 11/3/5/5): the loops unroll and
// everything that does not depend on d (left window, left tables) stays in registers; WS == 0 takes the window from
// the arguments.  `emit(d, c)` is called for d = 0..nd-1 in order.
template <int WS, class F>
__device__ __forceinline__ void census_costs(const FusedArgs& a, int yb, int xb, F emit) {
    const int W = a.Wb, H = a.Hb, ws = WS > 0 ? WS : a.censw, wc = ws / 2, i = yb - wc, j = xb - wc;
    constexpr int NW = WS > 0 ? (WS * WS + 31) / 32 : kCensusWords;
    const int nwords = WS > 0 ? NW : a.nwords;
    const bool ok = i >= 0 && j >= 0 && i < H - ws && j < W - ws;
    uint32_t lw[NW];
    const uint32_t* p = a.lb + ((size_t)yb * W + xb) * nwords;
#pragma unroll
    for (int k = 0; k < NW; ++k) lw[k] = (ok && k < nwords) ? p[k] : 0u;
    for (int d = 0; d < a.nd; ++d) {
        float c = kSentinel;
        if (ok && j >= d) {
            const uint32_t* q = a.rb + ((size_t)yb * W + xb - d) * nwords;
            int cnt = 0;
#pragma unroll
            for (int k = 0; k < NW; ++k)
                if (k < nwords) cnt += __popc(lw[k] ^ q[k]);
            c = (float)cnt;
        }
        emit(d, c);
    }
}

template <int WS, class F>
__device__ __forceinline__ void ncc_costs(const FusedArgs& a, int yb, int xb, F emit) {
    const int W = a.Wb, H = a.Hb, ws = WS > 0 ? WS : a.nccw, wc = ws / 2, i = yb - wc, j = xb - wc;
    const bool ok = i >= 0 && j >= 0 && i < H - ws && j < W - ws;
    const size_t cl = (size_t)yb * W + xb;
    const double Cl = ok ? a.ncl[cl] : 0.0, Al = ok ? a.nal[cl] : 0.0;
    const double sq = (double)(ws * ws);
    constexpr int NL = WS > 0 ? WS * WS : 1;
    unsigned lv[NL];
    if (WS > 0) {
#pragma unroll
        for (int k = 0; k < NL; ++k) lv[k] = ok ? a.l[(i + k / (WS > 0 ? WS : 1)) * W + j + k % (WS > 0 ? WS : 1)] : 0u;
    }
    for (int d = 0; d < a.nd; ++d) {
        float c = kSentinel;
        if (ok && j >= d) {
            const double Cr = a.ncr[cl - d];
            if (!(isfinite(Cl) && isfinite(Cr))) {
                c = 1.f;
            } else {
                unsigned long long LR = 0;
                if (WS > 0) {
                    unsigned acc = 0;      // WS*WS*255*255 < 2^32 for WS <= 16
#pragma unroll
                    for (int k = 0; k < NL; ++k)
                        acc += lv[k] * (unsigned)a.r[(i + k / (WS > 0 ? WS : 1)) * W + j - d + k % (WS > 0 ? WS : 1)];
                    LR = acc;
                } else {
                    for (int wh = 0; wh < ws; ++wh) {
                        const uint8_t* lp = a.l + (i + wh) * W + j;
                        const uint8_t* rp = a.r + (i + wh) * W + j - d;
                        for (int ww = 0; ww < ws; ++ww) LR += (unsigned)lp[ww] * (unsigned)rp[ww];
                    }
                }
                const double num = sq * (double)LR - Al * a.nar[cl - d];
                double t = -num;
                t = t * Cl;
                t = t * Cr;
                c = (float)t;
            }
        }
        emit(d, c);
    }
}

template <int WS, class F>
__device__ __forceinline__ void sobel_costs(const FusedArgs& a, int yb, int xb, F emit) {
    const int W = a.Wb, H = a.Hb, ws = WS > 0 ? WS : a.sobelw, wc = ws / 2, i = yb - wc, j = xb - wc;
    const bool ok = i >= 0 && j >= 0 && i < H - ws && j < W - ws;
    const size_t plane = (size_t)(H + 1) * (W + 1);
    const float* t = a.integ + (size_t)(ok ? i : 0) * (W + 1) + (ok ? j : 0);
    const float* b = t + (size_t)ws * (W + 1);
    for (int d = 0; d < a.nd; ++d, t += plane, b += plane) {
        float c = kSentinel;
        if (ok && j >= d) {
            float r = b[ws] - b[0];
            r = r - t[ws];
            r = r + t[0];
            c = r;
        }
        emit(d, c);
    }
}

template <int WS, class F>
__device__ __forceinline__ void zsad_costs(const FusedArgs& a, int yb, int xb, F emit) {
    const int W = a.Wb, H = a.Hb, ws = WS > 0 ? WS : a.sadw, wc = ws / 2, i = yb - wc, j = xb - wc;
    const bool ok = i >= 0 && j >= 0 && i < H - ws && j < W - ws;
    const size_t cl = (size_t)yb * W + xb;
    const float ml = ok ? a.ml[cl] : 0.f;
    constexpr int NL = WS > 0 ? WS * WS : 1;
    constexpr int WD = WS > 0 ? WS : 1;
    float lm[NL];                                       // (float)L - mean_L: first step of the reference's expression
    if (WS > 0) {
#pragma unroll
        for (int k = 0; k < NL; ++k) lm[k] = ok ? (float)a.l[(i + k / WD) * W + j + k % WD] - ml : 0.f;
    }
    if constexpr (WS > 0) {
        // WS disparities per round share one load of the 2*WS-1 right-image columns they touch (window of step u = columns
        // WS-1-u .. 2*WS-2-u of the strip): 9 byte loads per disparity instead of 25 for the 5x5 window -- the pass was
        // bound by the texture-address path, not the VALU.  Same (wh, ww) summation order as before.
        constexpr int NC = 2 * WS - 1;
        for (int d0 = 0; d0 < a.nd; d0 += WS) {
            float rs[WS][NC];
            const int c0 = j - d0 - (WS - 1);                       // image column of strip column 0
#pragma unroll
            for (int wh = 0; wh < WS; ++wh)
#pragma unroll
                for (int q = 0; q < NC; ++q) {
                    const int col = c0 + q;
                    rs[wh][q] = (ok && col >= 0) ? (float)a.r[(i + wh) * W + col] : 0.f;
                }
#pragma unroll
            for (int u = 0; u < WS; ++u) {
                const int d = d0 + u;
                if (d < a.nd) {
                    float c = kSentinel;
                    if (ok && j >= d) {
                        const float mr = a.mr[cl - d];
                        float acc = 0.f;
#pragma unroll
                        for (int k = 0; k < NL; ++k) {
                            float t = lm[k] - rs[k / WD][WS - 1 - u + k % WD];
                            t = t + mr;
                            acc = acc + fabsf(t);
                        }
                        c = acc;
                    }
                    emit(d, c);
                }
            }
        }
    } else {
        for (int d = 0; d < a.nd; ++d) {
            float c = kSentinel;
            if (ok && j >= d) {
                const float mr = a.mr[cl - d];
                const uint8_t* rp = a.r + i * W + j - d;
                const uint8_t* lp = a.l + i * W + j;
                float acc = 0.f;
                for (int wh = 0; wh < ws; ++wh)
                    for (int ww = 0; ww < ws; ++ww) {
                        float t = (float)lp[wh * W + ww] - ml;
                        t = t - (float)rp[wh * W + ww];
                        t = t + mr;
                        acc = acc + fabsf(t);
                    }
                c = acc;
            }
            emit(d, c);
        }
    }
}

// The raw costs are parked in the cost channel's own output slots (written and re-read by the same thread, so they
// stay in L2 and need no fence), then overwritten with the normalised cost in the last pass.
template <int M, int WS>
__device__ __forceinline__ void features_fused(const FusedArgs& a) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.Wc || y >= a.Hc) return;
    const int yb = y + a.bh, xb = x + a.bw;
    const size_t plane_out = (size_t)a.Hc * a.Wc;
    float* o_cost = a.out + ((size_t)M * a.nd) * plane_out + (size_t)y * a.Wc + x;
    float* o_aml = a.out + ((size_t)(4 + M) * a.nd) * plane_out + (size_t)y * a.Wc + x;
    float m = kSentinel;
    auto emit = [&](int d, float c) {
        o_cost[d * plane_out] = c;
        if (c < m) m = c;
    };
    if (M == 0) census_costs<WS>(a, yb, xb, emit);
    else if (M == 1) ncc_costs<WS>(a, yb, xb, emit);
    else if (M == 2) sobel_costs<WS>(a, yb, xb, emit);
    else zsad_costs<WS>(a, yb, xb, emit);
    const float sigma = a.sigma[M];
    float den = 0.f;
#pragma unroll 8
    for (int d = 0; d < a.nd; ++d) den += aml_term(o_cost[d * plane_out], m, sigma);
#pragma unroll 8
    for (int d = 0; d < a.nd; ++d) {
        const float c = o_cost[d * plane_out];
        o_cost[d * plane_out] = normalise_cost(M, c);
        o_aml[d * plane_out] = (m == kSentinel) ? 0.f : aml_term(c, m, sigma) / den;
    }
}

// All four matchers in one launch (blockIdx.z picks the matcher, most expensive first): a matcher alone has only
// Hc*Wc = 130 K threads, two waves per SIMD, and is bound by the latency of its dependent loads; together they overlap.
template <int CW, int NW, int SW, int ZW>
__global__ __launch_bounds__(256) void features_all_kernel(FusedArgs a) {
    switch (blockIdx.z) {
    case 0: features_fused<3, ZW>(a); break;
    case 1: features_fused<1, NW>(a); break;
    case 2: features_fused<2, SW>(a); break;
    default: features_fused<0, CW>(a); break;
    }
}

// ---------------------------------------------------------------- test-time pre-processing ------------
// cbmv_generator.py:780-788 (pad top/right to a multiple of encoder_ds), :465-482 (x 1/s rescale with anti-aliasing) and
// :819-823 (zero border), one thread per output pixel of the bordered image.  The rescale is skimage.transform.resize as
// restated in oracle/ms_volume.py (rescale_explicit): gaussian_filter(sigma=(s-1)/2, mode constant) = two separable
// passes in scipy's correlate1d order with double accumulation and float32 between the passes, then the mean of the two
// central source pixels per axis (even s) or the centre pixel (odd s), clip to [0, max], x255, truncate.
struct PrepArgs {
    const uint8_t* img; uint8_t* out; const unsigned* vmax_u8;
    int h, w, pad_h, Hp, Wp, s, r, border, Ho, Wo;       // Ho, Wo: rescaled size without the border
    double wt[32];                                       // gaussian taps, centre at wt[r]
};

__global__ void image_max_kernel(const uint8_t* __restrict__ img, size_t n, unsigned* __restrict__ out) {
    unsigned m = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        m = max(m, (unsigned)img[i]);
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

__device__ __forceinline__ float prep_pixel(const PrepArgs& a, int y, int x) {       // padded image / 255, zero outside
    if ((unsigned)y >= (unsigned)a.Hp || (unsigned)x >= (unsigned)a.w || y < a.pad_h) return 0.f;
    return (float)a.img[(size_t)(y - a.pad_h) * a.w + x] / 255.0f;
}
__device__ __forceinline__ float prep_vertical(const PrepArgs& a, int y, int x) {    // first pass (axis 0), float32 result
    if ((unsigned)x >= (unsigned)a.Wp) return 0.f;                                   // second pass pads with zeros
    double t = (double)prep_pixel(a, y, x) * a.wt[a.r];
    for (int j = -a.r; j < 0; ++j) {
        const double sum = (double)prep_pixel(a, y + j, x) + (double)prep_pixel(a, y - j, x);
        t = t + sum * a.wt[a.r + j];
    }
    return (float)t;
}
__device__ __forceinline__ float prep_filtered(const PrepArgs& a, int y, int x) {    // second pass (axis 1)
    double t = (double)prep_vertical(a, y, x) * a.wt[a.r];
    for (int j = -a.r; j < 0; ++j) {
        const double sum = (double)prep_vertical(a, y, x + j) + (double)prep_vertical(a, y, x - j);
        t = t + sum * a.wt[a.r + j];
    }
    return (float)t;
}

__global__ void preprocess_kernel(PrepArgs a) {
    const int xo = blockIdx.x * blockDim.x + threadIdx.x, yo = blockIdx.y;
    const int Wb = a.Wo + 2 * a.border;
    if (xo >= Wb) return;
    const int oy = yo - a.border, ox = xo - a.border;
    uint8_t v = 0;
    if ((unsigned)oy < (unsigned)a.Ho && (unsigned)ox < (unsigned)a.Wo) {
        if (a.s == 1) {
            v = (oy >= a.pad_h && ox < a.w) ? a.img[(size_t)(oy - a.pad_h) * a.w + ox] : 0;
        } else {
            const int lo = (a.s - 1) / 2, y0 = oy * a.s + lo, x0 = ox * a.s + lo;
            float o;
            if (a.s % 2 == 0) {
                double acc = 0.0;
                for (int dy = 0; dy < 2; ++dy) {
                    const double t = 0.25 * (double)prep_filtered(a, y0 + dy, x0) + 0.25 * (double)prep_filtered(a, y0 + dy, x0 + 1);
                    acc = acc + t;
                }
                o = (float)acc;
            } else {
                o = prep_filtered(a, y0, x0);
            }
            const float vmax = (float)(*a.vmax_u8) / 255.0f;
            o = fminf(fmaxf(o, 0.f), vmax);
            v = (uint8_t)(o * 255.0f);
        }
    }
    a.out[(size_t)yo * Wb + xo] = v;
}

// The five independent per-pixel preparations in one launch (blockIdx.z): census bits L / R, NCC + ZSAD tables, Sobel L / R.
__device__ __forceinline__ void census_transform_px(const uint8_t* __restrict__ img, uint32_t* __restrict__ bits, int H, int W,
                                                    int ws, int nwords, int x, int y) {
    const int wc = ws / 2;
    const int i = y - wc, j = x - wc;
    uint32_t wd[kCensusWords];
#pragma unroll
    for (int k = 0; k < kCensusWords; ++k) wd[k] = 0;
    if (i >= 0 && j >= 0 && i < H - ws && j < W - ws) {
        const int c = img[y * W + x];
        int b = 0;
        for (int wh = 0; wh < ws; ++wh)
            for (int ww = 0; ww < ws; ++ww, ++b)
                if (c < (int)img[(i + wh) * W + j + ww]) {
#pragma unroll
                    for (int k = 0; k < kCensusWords; ++k)
                        if (k == (b >> 5)) wd[k] |= 1u << (b & 31);
                }
    }
    for (int k = 0; k < nwords; ++k) bits[((size_t)y * W + x) * nwords + k] = wd[k];
}
__device__ __forceinline__ void sobel_px(const uint8_t* __restrict__ img, float* __restrict__ out, int H, int W, int x, int y) {
    const int i = y - 1, j = x - 1;
    float v = 0.f;
    if (i >= 0 && j >= 0 && i < H - 3 && j < W - 3) {
        const uint8_t* p = img + i * W + j;
        const int sv = -(int)p[0] + (int)p[2] - 2 * (int)p[W] + 2 * (int)p[W + 2] - (int)p[2 * W] + (int)p[2 * W + 2];
        v = (float)sv;
    }
    out[y * W + x] = v;
}
__global__ __launch_bounds__(64) void volume_prep_kernel(FusedArgs a, float* __restrict__ sobl, float* __restrict__ sobr) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y;
    if (x >= a.Wb) return;
    switch (blockIdx.z) {
    case 0: census_transform_px(a.l, const_cast<uint32_t*>(a.lb), a.Hb, a.Wb, a.censw, a.nwords, x, y); break;
    case 1: census_transform_px(a.r, const_cast<uint32_t*>(a.rb), a.Hb, a.Wb, a.censw, a.nwords, x, y); break;
    case 2: pixel_tables_px(a, x, y); break;
    case 3: sobel_px(a.l, sobl, a.Hb, a.Wb, x, y); break;
    default: sobel_px(a.r, sobr, a.Hb, a.Wb, x, y); break;
    }
}

// featextract.cpp:136-172 get_right_cost: res[i][j][d] = cost[i][j+d][d] for j < W-d, else cost[0][0][0] (the fill value the
// reference takes from the first element); cost / res are [H][W][D] float32.
__global__ void right_cost_kernel(const float* __restrict__ cost, float* __restrict__ res, int H, int W, int D) {
    const size_t total = (size_t)H * W * D;
    const float fill = cost[0];
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        const int d = (int)(o % D);
        const int j = (int)((o / D) % W);
        res[o] = (j < W - d) ? cost[o + (size_t)d * D] : fill;
    }
}

static inline int grid1d(size_t total, int cap = 16384) {
    const size_t b = (total + 255) / 256;
    return (int)(b < (size_t)cap ? (b ? b : 1) : cap);
}

static int check_img(const char* fn, const void* a, const void* b, const void* c, int H, int W, int nd, int ws) {
    if (!a || !b || !c) return fail("%s: null pointer", fn);
    if (H <= 0 || W <= 0) return fail("%s: empty image %dx%d", fn, H, W);
    if (nd <= 0) return fail("%s: ndisp=%d must be positive", fn, nd);
    if (ws <= 0 || ws > 15 || (ws & 1) == 0) return fail("%s: wsize=%d must be odd and <= 15", fn, ws);
    return 0;
}

static int census_impl(const uint8_t* l, const uint8_t* r, float* out, uint32_t* bits, int H, int W, int nd, int ws,
                       bool dmajor, hipStream_t s) {
    const int nwords = (ws * ws + 31) / 32;
    uint32_t* lb = bits;
    uint32_t* rb = bits + (size_t)H * W * nwords;
    dim3 g2(cdiv(W, 64), H);
    {
        LaunchScope ls("census_transform", s, 0, 2.0 * H * W * (1 + 4.0 * nwords));
        hipLaunchKernelGGL(census_transform_kernel, g2, dim3(64), 0, s, l, lb, H, W, ws, nwords);
        hipLaunchKernelGGL(census_transform_kernel, g2, dim3(64), 0, s, r, rb, H, W, ws, nwords);
    }
    const size_t total = (size_t)H * W * nd;
    LaunchScope ls("census_cost", s, 0, 4.0 * total);
    if (dmajor) hipLaunchKernelGGL(census_cost_kernel<true>, dim3(grid1d(total)), dim3(256), 0, s, lb, rb, out, H, W, nd, ws, nwords);
    else        hipLaunchKernelGGL(census_cost_kernel<false>, dim3(grid1d(total)), dim3(256), 0, s, lb, rb, out, H, W, nd, ws, nwords);
    return check_launch("census");
}

static int sadsob_impl(const float* sl, const float* sr, float* out, float* ws, int H, int W, int nd, int wsz, hipStream_t s) {
    {
        LaunchScope ls("sadsob_vertical", s, 0, 4.0 * nd * (double)(H + 1) * (W + 1));
        hipLaunchKernelGGL(sadsob_vertical_kernel, dim3(cdiv(W + 1, 64), nd), dim3(64), 0, s, sl, sr, ws, H, W, nd);
    }
    {
        LaunchScope ls("sadsob_horizontal", s, 0, 8.0 * nd * (double)(H + 1) * (W + 1));
        hipLaunchKernelGGL(sadsob_horizontal_kernel, dim3(cdiv(H + 1, 64), nd), dim3(64), 0, s, ws, H, W, nd);
    }
    const size_t total = (size_t)nd * H * W;
    LaunchScope ls("sadsob_box", s, 0, 8.0 * total);
    hipLaunchKernelGGL(sadsob_box_kernel, dim3(grid1d(total)), dim3(256), 0, s, ws, out, H, W, nd, wsz);
    return check_launch("sadsob");
}

// fast path for the reference's default windows (volume_fused.hip)
bool volume_fast_supported(const msnet_volume_params& p, int Hb, int Wb, int nd);
size_t volume_fast_workspace_bytes(int Hb, int Wb, int nd);
int volume_fast_launch(const uint8_t* l, const uint8_t* r, int Hb, int Wb, int nd, const msnet_volume_params& p, void* workspace,
                       float* out, hipStream_t s, bool channels_last);

}  // namespace msnet

using namespace msnet;

extern "C" size_t msnet_census_workspace_bytes(int H, int W, int wsize) {
    if (H <= 0 || W <= 0 || wsize <= 0) return 0;
    return (size_t)2 * H * W * ((wsize * wsize + 31) / 32) * sizeof(uint32_t);
}

extern "C" int msnet_census(const uint8_t* l, const uint8_t* r, float* out, void* workspace, int H, int W, int ndisp,
                            int wsize, msnet_stream_t stream) {
    if (int e = check_img("msnet_census", l, r, out, H, W, ndisp, wsize)) return e;
    if (!workspace) return fail("msnet_census: null workspace");
    return census_impl(l, r, out, (uint32_t*)workspace, H, W, ndisp, wsize, false, (hipStream_t)stream);
}

extern "C" int msnet_ncc(const uint8_t* l, const uint8_t* r, float* out, int H, int W, int ndisp, int wsize,
                         msnet_stream_t stream) {
    if (int e = check_img("msnet_ncc", l, r, out, H, W, ndisp, wsize)) return e;
    hipStream_t s = (hipStream_t)stream;
    const size_t total = (size_t)ndisp * H * W;
    LaunchScope ls("ncc", s, 0, 4.0 * total);
    hipLaunchKernelGGL(ncc_kernel, dim3(grid1d(total)), dim3(256), 0, s, l, r, out, H, W, ndisp, wsize);
    return check_launch("msnet_ncc");
}

extern "C" int msnet_zsad(const uint8_t* l, const uint8_t* r, float* out, int H, int W, int ndisp, int wsize,
                          msnet_stream_t stream) {
    if (int e = check_img("msnet_zsad", l, r, out, H, W, ndisp, wsize)) return e;
    hipStream_t s = (hipStream_t)stream;
    const size_t total = (size_t)ndisp * H * W;
    LaunchScope ls("zsad", s, 0, 4.0 * total);
    hipLaunchKernelGGL(zsad_kernel, dim3(grid1d(total)), dim3(256), 0, s, l, r, out, H, W, ndisp, wsize);
    return check_launch("msnet_zsad");
}

extern "C" int msnet_sobel(const uint8_t* img, float* out, int H, int W, msnet_stream_t stream) {
    if (!img || !out) return fail("msnet_sobel: null pointer");
    if (H <= 0 || W <= 0) return fail("msnet_sobel: empty image");
    hipStream_t s = (hipStream_t)stream;
    LaunchScope ls("sobel", s, 0, 5.0 * H * W);
    hipLaunchKernelGGL(sobel_kernel, dim3(cdiv(W, 64), H), dim3(64), 0, s, img, out, H, W);
    return check_launch("msnet_sobel");
}

extern "C" size_t msnet_sadsob_workspace_bytes(int H, int W, int ndisp) {
    if (H <= 0 || W <= 0 || ndisp <= 0) return 0;
    return (size_t)ndisp * (H + 1) * (W + 1) * sizeof(float);
}

extern "C" int msnet_sadsob(const float* sl, const float* sr, float* out, void* workspace, int H, int W, int ndisp,
                            int wsize, msnet_stream_t stream) {
    if (int e = check_img("msnet_sadsob", sl, sr, out, H, W, ndisp, wsize)) return e;
    if (!workspace) return fail("msnet_sadsob: null workspace");
    return sadsob_impl(sl, sr, out, (float*)workspace, H, W, ndisp, wsize, (hipStream_t)stream);
}

extern "C" int msnet_swap_axes(const float* in, float* out, int D, int H, int W, msnet_stream_t stream) {
    if (!in || !out) return fail("msnet_swap_axes: null pointer");
    if (D <= 0 || H <= 0 || W <= 0) return fail("msnet_swap_axes: empty tensor");
    const long S = (long)H * W;
    hipStream_t s = (hipStream_t)stream;
    LaunchScope ls("swap_axes", s, 0, 8.0 * D * (double)S);
    hipLaunchKernelGGL(swap_axes_kernel, dim3((unsigned)((S + 63) / 64), cdiv(D, 64)), dim3(256), 0, s, in, out, D, S);
    return check_launch("msnet_swap_axes");
}

extern "C" int msnet_get_right_cost(const float* cost, float* out, int H, int W, int D, msnet_stream_t stream) {
    if (!cost || !out) return fail("msnet_get_right_cost: null pointer");
    if (cost == out) return fail("msnet_get_right_cost: in-place is not supported");
    if (D <= 0 || H <= 0 || W <= 0) return fail("msnet_get_right_cost: empty tensor");
    hipStream_t s = (hipStream_t)stream;
    const size_t total = (size_t)H * W * D;
    LaunchScope ls("get_right_cost", s, 0, 8.0 * total);
    hipLaunchKernelGGL(right_cost_kernel, dim3(grid1d(total)), dim3(256), 0, s, cost, out, H, W, D);
    return check_launch("msnet_get_right_cost");
}

extern "C" int msnet_extract_likelihood(const float* vol, float* out, long P, int D, float sigma, msnet_stream_t stream) {
    if (!vol || !out) return fail("msnet_extract_likelihood: null pointer");
    if (P <= 0 || D <= 0) return fail("msnet_extract_likelihood: empty volume");
    const size_t lds = (size_t)64 * (D + 1) * sizeof(float);
    if (lds > 160 * 1024) return fail("msnet_extract_likelihood: D=%d too large for one LDS tile", D);
    hipStream_t s = (hipStream_t)stream;
    if (lds > 65536) (void)hipFuncSetAttribute((const void*)aml_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    LaunchScope ls("extract_likelihood", s, 0, 8.0 * P * (double)D);
    hipLaunchKernelGGL(aml_rows_kernel, dim3((unsigned)((P + 63) / 64)), dim3(64), lds, s, vol, out, P, D, sigma);
    return check_launch("msnet_extract_likelihood");
}

extern "C" int msnet_extract_features_left(const float* census, const float* ncc, const float* sobel, const float* sad,
                                           float* out, long P, int D, float cens_sigma, float ncc_sigma, float sad_sigma,
                                           msnet_stream_t stream) {
    if (!census || !ncc || !sobel || !sad || !out) return fail("msnet_extract_features_left: null pointer");
    if (P <= 0 || D <= 0) return fail("msnet_extract_features_left: empty volume");
    const size_t lds = (size_t)64 * (D + 1) * sizeof(float);
    if (lds > 160 * 1024) return fail("msnet_extract_features_left: D=%d too large for one LDS tile", D);
    hipStream_t s = (hipStream_t)stream;
    if (lds > 65536) (void)hipFuncSetAttribute((const void*)features_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const float* in[4] = {census, ncc, sobel, sad};
    const float sg[4] = {cens_sigma, ncc_sigma, sad_sigma, sad_sigma};
    LaunchScope ls("features_rows", s, 0, 4.0 * 12.0 * P * (double)D);
    for (int ch = 0; ch < 4; ++ch)
        hipLaunchKernelGGL(features_rows_kernel, dim3((unsigned)((P + 63) / 64)), dim3(64), lds, s, in[ch],
                           out + (size_t)ch * D * P, out + (size_t)(4 + ch) * D * P, P, D, sg[ch], ch);
    return check_launch("msnet_extract_features_left");
}

extern "C" void msnet_volume_default_params(msnet_volume_params* p) {
    if (!p) return;
    p->censw = 11; p->nccw = 3; p->sadw = 5; p->sobelw = 5;
    p->cens_sigma = 128.f; p->ncc_sigma = 0.02f; p->sad_sigma = 20000.f;
    p->border_h = 10; p->border_w = 10;
}

// workspace carve: sadsob integral nd*(Hb+1)*(Wb+1) f32 | sobel L,R 2*Hb*Wb f32 | ZSAD means 2*Hb*Wb f32 |
// NCC tables 4*Hb*Wb f64 | census bit images 2*Hb*Wb*8 u32
extern "C" size_t msnet_build_volume_workspace_bytes(int Hb, int Wb, int ndisp) {
    if (Hb <= 0 || Wb <= 0 || ndisp <= 0) return 0;
    const size_t img = (size_t)Hb * Wb;
    const size_t bytes = ((size_t)ndisp * (Hb + 1) * (Wb + 1) + 4 * img) * sizeof(float) + 4 * img * sizeof(double) +
                         16 * img * sizeof(uint32_t) + 64;
    const size_t fast = volume_fast_workspace_bytes(Hb, Wb, ndisp);
    return ((bytes > fast ? bytes : fast) + 255) & ~(size_t)255;
}

extern "C" int msnet_build_volume(const uint8_t* l, const uint8_t* r, int Hb, int Wb, int ndisp,
                                  const msnet_volume_params* pp, void* workspace, float* out, msnet_stream_t stream) {
    msnet_volume_params p;
    if (pp) p = *pp; else msnet_volume_default_params(&p);
    if (!l || !r || !workspace || !out) return fail("msnet_build_volume: null pointer");
    if (ndisp <= 0) return fail("msnet_build_volume: ndisp=%d", ndisp);
    const int Hc = Hb - 2 * p.border_h, Wc = Wb - 2 * p.border_w;
    if (p.border_h < 0 || p.border_w < 0 || Hc <= 0 || Wc <= 0)
        return fail("msnet_build_volume: %dx%d image with border %dx%d leaves nothing", Hb, Wb, p.border_h, p.border_w);
    if (int e = check_img("msnet_build_volume", l, r, out, Hb, Wb, ndisp, p.censw)) return e;
    if (int e = check_img("msnet_build_volume", l, r, out, Hb, Wb, ndisp, p.nccw)) return e;
    if (int e = check_img("msnet_build_volume", l, r, out, Hb, Wb, ndisp, p.sadw)) return e;
    if (int e = check_img("msnet_build_volume", l, r, out, Hb, Wb, ndisp, p.sobelw)) return e;
    hipStream_t s = (hipStream_t)stream;
    // MSNET_VOLUME_GENERIC=1 forces the run-time-window kernels below (A/B tests of the two paths)
    const char* generic = getenv("MSNET_VOLUME_GENERIC");
    if (!(generic && generic[0] == '1') && volume_fast_supported(p, Hb, Wb, ndisp))
        return volume_fast_launch(l, r, Hb, Wb, ndisp, p, workspace, out, s, false);
    const size_t img = (size_t)Hb * Wb;
    LaunchScope whole("vol_build", s, 0, 4.0 * 8.0 * ndisp * (double)Hc * Wc + 2.0 * img);     // the whole build (bench.py's roofline_volume)
    float* integ = (float*)workspace;
    float* sobl = integ + (size_t)ndisp * (Hb + 1) * (Wb + 1);
    float* sobr = sobl + img;
    float* ml = sobr + img;
    float* mr = ml + img;
    double* tab = (double*)(((uintptr_t)(mr + img) + 15) & ~(uintptr_t)15);
    uint32_t* bits = (uint32_t*)(tab + 4 * img);
    const int nwords = (p.censw * p.censw + 31) / 32;

    FusedArgs a{};
    a.l = l; a.r = r; a.lb = bits; a.rb = bits + img * nwords;
    a.nal = tab; a.ncl = tab + img; a.nar = tab + 2 * img; a.ncr = tab + 3 * img;
    a.ml = ml; a.mr = mr; a.integ = integ; a.out = out;
    a.sigma[0] = p.cens_sigma; a.sigma[1] = p.ncc_sigma; a.sigma[2] = p.sad_sigma; a.sigma[3] = p.sad_sigma;
    a.Hb = Hb; a.Wb = Wb; a.nd = ndisp; a.bh = p.border_h; a.bw = p.border_w; a.Hc = Hc; a.Wc = Wc;
    a.censw = p.censw; a.nccw = p.nccw; a.sadw = p.sadw; a.sobelw = p.sobelw; a.nwords = nwords;

    {
        LaunchScope ls("volk_prep", s, 0, 2.0 * img * (1 + 4.0 * nwords) + 48.0 * img);
        hipLaunchKernelGGL(volume_prep_kernel, dim3(cdiv(Wb, 64), Hb, 5), dim3(64), 0, s, a, sobl, sobr);
    }
    {
        LaunchScope ls("volk_sadsob_v", s, 0, 4.0 * ndisp * (double)(Hb + 1) * (Wb + 1));
        hipLaunchKernelGGL(sadsob_vertical_kernel, dim3(cdiv(Wb + 1, 64), ndisp), dim3(64), 0, s, sobl, sobr, integ, Hb, Wb, ndisp);
    }
    {
        LaunchScope ls("volk_sadsob_h", s, 0, 8.0 * ndisp * (double)(Hb + 1) * (Wb + 1));
        hipLaunchKernelGGL(sadsob_horizontal_kernel, dim3(cdiv(Hb + 1, 64), ndisp), dim3(64), 0, s, integ, Hb, Wb, ndisp);
    }
    {
        LaunchScope ls("volk_features", s, 0, 4.0 * 8.0 * ndisp * (double)Hc * Wc);
        const dim3 gf(cdiv(Wc, 64), cdiv(Hc, 4), 4);
        if (p.censw == 11 && p.nccw == 3 && p.sobelw == 5 && p.sadw == 5)
            hipLaunchKernelGGL((features_all_kernel<11, 3, 5, 5>), gf, dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL((features_all_kernel<0, 0, 0, 0>), gf, dim3(256), 0, s, a);
    }
    return check_launch("msnet_build_volume");
}

extern "C" int msnet_build_volume_ndhwc_supported(int Hb, int Wb, int ndisp, const msnet_volume_params* pp) {
    msnet_volume_params p;
    if (pp) p = *pp; else msnet_volume_default_params(&p);
    if (Hb <= 0 || Wb <= 0 || ndisp <= 0) return 0;
    const int Hc = Hb - 2 * p.border_h, Wc = Wb - 2 * p.border_w;
    if (p.border_h < 0 || p.border_w < 0 || Hc <= 0 || Wc <= 0) return 0;
    return volume_fast_supported(p, Hb, Wb, ndisp) && (size_t)ndisp * Hc * Wc * 32 <= 0xfffffff0u ? 1 : 0;
}

extern "C" int msnet_build_volume_ndhwc(const uint8_t* l, const uint8_t* r, int Hb, int Wb, int ndisp,
                                        const msnet_volume_params* pp, void* workspace, float* out, msnet_stream_t stream) {
    msnet_volume_params p;
    if (pp) p = *pp; else msnet_volume_default_params(&p);
    if (!l || !r || !workspace || !out) return fail("msnet_build_volume_ndhwc: null pointer");
    if (!msnet_build_volume_ndhwc_supported(Hb, Wb, ndisp, &p))
        return fail("msnet_build_volume_ndhwc: only the reference's own windows (11/3/5/5), borders >= 6 and D' a multiple of 8 up "
                    "to 96 are built channels-last; use msnet_build_volume + msnet_ncdhw_to_ndhwc");
    if (int e = check_img("msnet_build_volume_ndhwc", l, r, out, Hb, Wb, ndisp, p.censw)) return e;
    return volume_fast_launch(l, r, Hb, Wb, ndisp, p, workspace, out, (hipStream_t)stream, true);
}

extern "C" int msnet_preprocess_out_shape(int h, int w, int encoder_ds, int ds, int border, int* Hb, int* Wb) {
    if (h <= 0 || w <= 0 || encoder_ds <= 0 || ds <= 0 || border < 0 || encoder_ds % ds != 0)
        return fail("msnet_preprocess_out_shape: bad arguments (h=%d w=%d encoder_ds=%d ds=%d border=%d)", h, w, encoder_ds, ds, border);
    const int Hp = h + (encoder_ds - h % encoder_ds) % encoder_ds, Wp = w + (encoder_ds - w % encoder_ds) % encoder_ds;
    if (Hb) *Hb = Hp / ds + 2 * border;
    if (Wb) *Wb = Wp / ds + 2 * border;
    return 0;
}

extern "C" int msnet_preprocess_image(const uint8_t* img, int h, int w, int encoder_ds, int ds, int border,
                                      const double* taps_host, uint8_t* out, void* workspace, msnet_stream_t stream) {
    if (!img || !out || !workspace) return fail("msnet_preprocess_image: null pointer");
    int Hb = 0, Wb = 0;
    if (int e = msnet_preprocess_out_shape(h, w, encoder_ds, ds, border, &Hb, &Wb)) return e;
    if (ds > 8) return fail("msnet_preprocess_image: ds=%d (at most 8: the gaussian has 4*(ds-1)+1 <= 29 taps)", ds);
    PrepArgs a{};
    a.img = img; a.out = out; a.vmax_u8 = (const unsigned*)workspace;
    a.h = h; a.w = w;
    a.Hp = h + (encoder_ds - h % encoder_ds) % encoder_ds; a.Wp = w + (encoder_ds - w % encoder_ds) % encoder_ds;
    a.pad_h = a.Hp - h; a.s = ds; a.border = border; a.Ho = a.Hp / ds; a.Wo = a.Wp / ds;
    const double sigma = (ds - 1) / 2.0;
    a.r = (int)(4.0 * sigma + 0.5);
    if (ds == 1) { a.r = 0; a.wt[0] = 1.0; }
    else if (taps_host) {                               // the caller's 2r+1 taps (the Python mirror passes scipy's, bit for bit)
        for (int k = 0; k <= 2 * a.r; ++k) a.wt[k] = taps_host[k];
    } else {                                            // same formula in libm; may differ from NumPy's in the last ulp
        double sum = 0.0;
        for (int k = -a.r; k <= a.r; ++k) a.wt[a.r + k] = exp(-0.5 / (sigma * sigma) * (double)(k * k));
        for (int k = 0; k <= 2 * a.r; ++k) sum += a.wt[k];
        for (int k = 0; k <= 2 * a.r; ++k) a.wt[k] /= sum;
    }
    hipStream_t s = (hipStream_t)stream;
    LaunchScope ls("preprocess", s, 0, (double)h * w + (double)Hb * Wb);
    if (hipMemsetAsync(workspace, 0, 4, s) != hipSuccess) return fail("msnet_preprocess_image: memset failed");
    hipLaunchKernelGGL(image_max_kernel, dim3(64), dim3(256), 0, s, img, (size_t)h * w, (unsigned*)workspace);
    hipLaunchKernelGGL(preprocess_kernel, dim3(cdiv(Wb, 64), Hb), dim3(64), 0, s, a);
    return check_launch("msnet_preprocess_image");
}

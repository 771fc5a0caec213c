// Shared device helpers of the 3x3x3 conv kernels (conv3d.hip: exact fp32 MFMA; conv3d_f16s.hip: split-fp16 MFMA).
#pragma once
#include <stdlib.h>
#include "common.h"


namespace msnet {

// CUs of the CURRENT device (cached per device ordinal: one process may drive several devices in turn)
static inline int num_cus() {
    static int cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cache[dev]) {
        int v = 0;
        cache[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    return cache[dev];
}

struct ConvArgs {
    const float* x; const f32x4* wpk; const float* scale; const float* shift; const float* res; float* y;
    int N, D, H, W;        // input spatial dims
    int OD, OH, OW;        // output spatial dims
    int Ci, Co;
    int relu;
    int ntd, nth, ntw;     // tiles per dim (conv: over output dims; deconv: over input dims)
    int ngroups;           // Co / (32 * WN * NB)
    int nbtot;             // Co / 32
    int nseg, seglen;      // sliding-window kernels: depth segments per tile column, tiles (depth steps) per segment
    unsigned* oflag;       // device word that receives 1 when an output leaves the fp16 range (msnet_set_overflow_flag), or null
};

// Range guard of the split-fp16 kernels (conv3d_f16s.hip: every operand's `hi` half is an fp16).  The limit is HALF the largest
// finite fp16: the Winograd-depth loaders split sums and differences of two activations (q1 = p1 + p2, ...), and those must
// stay finite as fp16 too; below 32752 every kernel of the file is safe.  Every epilogue folds the magnitudes it stores into one
// compare and raises the caller's flag; the Python modules then re-run the forward on the exact fp32 kernels (hipops.py).
// 16 v_max + 1 compare per 16 outputs.  (v_max drops a NaN: a NaN never shows in `amax`.  With in-range operands none can arise
// -- they come from a non-finite module INPUT, which the input checks below report through the magnitude BITS.)
constexpr float kSplitMax = 32752.f;
constexpr unsigned kSplitMaxBits = 0x46ffe000u;             // bits of 32752.f: (bits & 0x7fffffff) >= this <=> |x| >= 32752, inf or NaN
__device__ __forceinline__ void flag_overflow(unsigned* oflag, float amax) {
    if (oflag && !(amax < kSplitMax)) atomicOr(oflag, 1u);    // (inf maxima included)
}
__device__ __forceinline__ unsigned magnitude_bits(float v) { return __builtin_bit_cast(unsigned, v) & 0x7fffffffu; }

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// Stage NPOS voxels x CC channels (global NDHWC, channel offset c0 of Ci) into LDS [pos][CC+4].
// Voxels outside the input are the convolution's zero padding.
template <int CC, int ID, int IH, int IW, int NT = 256>
__device__ __forceinline__ void stage_tile(float* lds, const float* __restrict__ x, int n, int D, int H,
                                           int W, int Ci, int c0, int id0, int ih0, int iw0, int tid) {
    constexpr int PS = CC + 4;
    constexpr int V = CC / 4;                 // float4 per voxel
    constexpr int NSLOT = ID * IH * IW * V;
    constexpr int U = 8;                      // loads in flight per thread
    for (int base = 0; base < NSLOT; base += NT * U) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int slot = base + u * NT + tid;
            const int pos = slot / V, c4 = slot % V;
            const int iw = pos % IW, ih = (pos / IW) % IH, id = pos / (IW * IH);
            const int gd = id0 + id, gh = ih0 + ih, gw = iw0 + iw;
            v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (slot < NSLOT && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H &&
                (unsigned)gw < (unsigned)W) {
                const size_t vox = (((size_t)n * D + gd) * H + gh) * W + gw;
                v[u] = *reinterpret_cast<const f32x4*>(x + vox * Ci + c0 + c4 * 4);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int slot = base + u * NT + tid;
            if (slot < NSLOT) {
                const int pos = slot / V, c4 = slot % V;
                *reinterpret_cast<f32x4*>(lds + pos * PS + c4 * 4) = v[u];
            }
        }
    }
}

template <int NB>
__device__ __forceinline__ void load_b(f32x4 (&b)[NB], const f32x4* __restrict__ p) {
#pragma unroll
    for (int j = 0; j < NB; ++j) b[j] = p[j * 64];
}

// Epilogue of one 32x32 accumulator block: y = act(acc*scale + shift (+ residual)).  For accumulator register e
// the 32 lanes of a half hold the 32 channels of ONE voxel (two full 128-byte lines per store instruction).
// Voxel of (e, half hh): local row = c_e + 4*hh with c_e = (e&3) + 8*(e>>2); because c_e % BW is in 0..3 (+8k) the
// 4*hh never carries into the h index, so the offset splits into a lane-dependent base (folded into `base` by the
// caller) plus compile-time multiples of two uniform strides.  FULL tiles take the branch-free path: all residual
// loads are issued before the first use (the naive per-element form serialised 16 dependent HBM round trips per
// block and ran the transposed convs at 30 TFLOP/s, profiles/r01b).
template <int BW, class Valid>
__device__ __forceinline__ void epilogue_block(const f32x16& acc, float sc, float sh, const float* __restrict__ res,
                                               float* __restrict__ y, size_t base, int stride_h, int stride_w, int relu,
                                               bool full, Valid valid, unsigned* oflag = nullptr) {
    float amax = 0.f;
    if (full) {
        float rv[16];
        if (res) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int c = (e & 3) + 8 * (e >> 2);
                rv[e] = res[base + (size_t)((c / BW) * stride_h + (c % BW) * stride_w)];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) rv[e] = 0.f;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int c = (e & 3) + 8 * (e >> 2);
            float v = acc[e] * sc + sh + rv[e];
            if (relu) v = fmaxf(v, 0.f);
            amax = fmaxf(amax, fabsf(v));
            y[base + (size_t)((c / BW) * stride_h + (c % BW) * stride_w)] = v;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int c = (e & 3) + 8 * (e >> 2);
            if (valid(c / BW, c % BW)) {
                const size_t idx = base + (size_t)((c / BW) * stride_h + (c % BW) * stride_w);
                float v = acc[e] * sc + sh;
                if (res) v += res[idx];
                if (relu) v = fmaxf(v, 0.f);
                amax = fmaxf(amax, fabsf(v));
                y[idx] = v;
            }
        }
    }
    flag_overflow(oflag, amax);
}

// The same epilogue in two halves over buffer descriptors, for kernels that can issue the residual loads long before the
// accumulators are final (the loads' HBM latency then hides behind the MFMA groups instead of stalling each batch: with
// one resident block per CU the batched form above keeps only ~16 KB in flight per CU, a quarter of what HBM needs).
// Both halves are branch-free: an element outside the tensor gets byte offset 0xffffffff, which the buffer bounds check
// turns into "load 0" / "store dropped".  (With per-element `if`s the compiler's waitcnt pass put s_waitcnt vmcnt(0) in
// front of every store, i.e. each store waited for the previous store's acknowledgement: 9000 cycles per 32 stores.)
// `off` = byte offset of the lane's (row 0, col 0) element inside the buffer; strides in bytes.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(bytes > 0xfffffffcu ? 0xfffffffcu : bytes), 0x00020000);
}

template <int BW, class Valid>
__device__ __forceinline__ void residual_prefetch(f32x16& rv, __amdgpu_buffer_rsrc_t res, unsigned off, int stride_h,
                                                  int stride_w, Valid valid) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int c = (e & 3) + 8 * (e >> 2);
        const unsigned o = valid(c / BW, c % BW) ? off + (unsigned)((c / BW) * stride_h + (c % BW) * stride_w) : 0xffffffffu;
        rv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(res, o, 0, 0));
    }
}

template <int BW, class Valid>
__device__ __forceinline__ void epilogue_store(const f32x16& acc, const f32x16& rv, float sc, float sh,
                                               __amdgpu_buffer_rsrc_t y, unsigned off, int stride_h, int stride_w, int relu,
                                               Valid valid, unsigned* oflag = nullptr) {
    float amax = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int c = (e & 3) + 8 * (e >> 2);
        const bool ok = valid(c / BW, c % BW);
        const unsigned o = ok ? off + (unsigned)((c / BW) * stride_h + (c % BW) * stride_w) : 0xffffffffu;
        float v = acc[e] * sc + sh + rv[e];
        if (relu) v = fmaxf(v, 0.f);
        amax = fmaxf(amax, ok ? fabsf(v) : 0.f);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), y, o, 0, 0);
    }
    flag_overflow(oflag, amax);
}

// (A voxel-lane variant of this epilogue -- weights as MFMA operand A, so that a lane holds four consecutive channels of
// one voxel and moves 16 bytes per buffer op -- was built and measured: 4x fewer instructions but every instruction then
// touches 32 cache lines instead of 2, and a CU's store path drains ~16 B/clk either way; +-1 %, dropped.  Draining a
// finished block one piece per K-step under the next block's MFMAs did not help either: the bursts that cost time are the
// residual loads and the tile hand-over)

// LDS-only workgroup barrier: orders LDS traffic across the s_barrier without draining the vector-memory
// counter (a plain __syncthreads() may add s_waitcnt vmcnt(0), which would stall the compute waves on their
// in-flight weight loads and the loader waves on nothing useful).
#define MSNET_LDS_BARRIER()                                              \
    do {                                                                 \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");  \
        __builtin_amdgcn_s_barrier();                                    \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");  \
    } while (0)

// Barrier for a wave that only READS LDS between two barriers (the MFMA waves at a weight-group barrier): no release fence,
// i.e. no s_waitcnt lgkmcnt(0) -- its fragment prefetches stay in flight across the barrier.  Safe where nothing another wave
// writes after this barrier can be the target of a read this wave issued before it (the callers state why); the partner waves
// publish their LDS writes with the full MSNET_LDS_BARRIER.  The memory clobber keeps the compiler from moving LDS accesses
// across it; the wave itself executes in order.
#define MSNET_READER_BARRIER()                                           \
    do {                                                                 \
        __builtin_amdgcn_sched_barrier(0);                               \
        asm volatile("s_barrier" ::: "memory");                          \
        __builtin_amdgcn_sched_barrier(0);                               \
    } while (0)

}  // namespace msnet

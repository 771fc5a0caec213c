// Measured-attainable peaks of the device the library runs on ("confirm on the box ... a STREAM-like
// copy and an MFMA micro-bench and use measured-attainable peaks alongside vendor peaks").  Diagnostics for bench.py's roofline
// denominators; nothing on the product path calls these.
//   msnet_peak_copy      : float4 copy src -> dst (bytes read + bytes written per call = 2 * bytes)
//   msnet_peak_mfma_f16  : every wave of a full-chip grid issues `iters` x 8 independent v_mfma_f32_32x32x16_f16 on register
//                          operands (no memory traffic): the dense fp16 MFMA rate at the clock the chip sustains under that load
#include "common.h"

namespace msnet {

typedef _Float16 half8_p __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void peak_copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n4) {
    // one contiguous 32 KB chunk per workgroup and iteration, eight independent 16-byte loads per thread in flight before
    // the first store (the shape that reached the highest rate of the variants tried on this part)
    constexpr int U = 8;
    const size_t chunk = (size_t)256 * U;
    for (size_t base = (size_t)blockIdx.x * chunk; base < n4; base += (size_t)gridDim.x * chunk) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const size_t i = base + (size_t)u * 256 + threadIdx.x; v[u] = i < n4 ? src[i] : f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int u = 0; u < U; ++u) { const size_t i = base + (size_t)u * 256 + threadIdx.x; if (i < n4) dst[i] = v[u]; }
    }
}

__global__ __launch_bounds__(256) void peak_mfma_f16_kernel(float* __restrict__ out, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    half8_p a, b;
#pragma unroll
    for (int k = 0; k < 8; ++k) {                          // non-trivial operands: zeros would let the chip clock higher
        a[k] = (_Float16)(seed * (float)((lane * 7 + k * 3) % 17 - 8) * 0.0625f);
        b[k] = (_Float16)(seed * (float)((lane * 5 + k * 11) % 13 - 6) * 0.125f);
    }
    f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[j][e];
    if (s == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = s;     // keeps the MFMAs live, practically never stores
}

// the same loop on the 16x16x32 shape (16 accumulators of 4 registers: equal FLOPs per iteration, half the cycles per
// instruction): MI355X_MICROARCH.md reports a higher sustained clock for this shape on bf16
typedef float f32x4_p __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void peak_mfma_f16_16x16_kernel(float* __restrict__ out, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    half8_p a, b;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        a[k] = (_Float16)(seed * (float)((lane * 7 + k * 3) % 17 - 8) * 0.0625f);
        b[k] = (_Float16)(seed * (float)((lane * 5 + k * 11) % 13 - 6) * 0.125f);
    }
    f32x4_p acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) s += acc[j][e];
    if (s == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// The loops above multiply the SAME two operand registers over and over: nothing toggles on the operand paths, which
// understates what the chip's power management does to a real kernel (the conv kernels run 1.5x faster on all-zero data than
// on random data with the identical instruction stream, tools/tools_power_probe.py).  The *_rand loops cycle through eight
// different pseudo-random A fragments and four B fragments (all in registers, every MFMA sees operands that differ from the
// previous one's in most bits): the dense fp16 MFMA rate the chip SUSTAINS on changing data -- the fair ceiling for a kernel
// whose operands are fresh for every instruction.
__device__ __forceinline__ half8_p rand_frag(unsigned& st) {
    half8_p v;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        st = st * 1664525u + 1013904223u;
        v[k] = (_Float16)(((float)(st >> 8) * (1.0f / 16777216.0f) - 0.5f) * 4.0f);
    }
    return v;
}

template <bool S16>
__global__ __launch_bounds__(256) void peak_mfma_f16_rand_kernel(float* __restrict__ out, int iters) {
    unsigned st = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    half8_p a[8], b[4];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = rand_frag(st);
#pragma unroll
    for (int k = 0; k < 4; ++k) b[k] = rand_frag(st);
    float s = 0.f;
    if constexpr (!S16) {
        f32x16 acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(j * 3 + 1) & 7], b[j & 3], acc[j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += acc[j][e];
    } else {
        f32x4_p acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(j * 3 + 1) & 7], b[j & 3], acc[j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) s += acc[j][e];
    }
    if (s == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

}  // namespace msnet

using namespace msnet;

/* fp16 MFMA rate on CHANGING operands (shape16 = 0: v_mfma_f32_32x32x16_f16, 1: v_mfma_f32_16x16x32_f16); returns the FLOPs of the call */
extern "C" double msnet_peak_mfma_f16_rand(void* scratch, int iters, int shape16, msnet_stream_t stream) {
    if (!scratch || iters <= 0) { fail("msnet_peak_mfma_f16_rand: bad arguments"); return 0.0; }
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int blocks = cus * 2;
    if (shape16) hipLaunchKernelGGL(peak_mfma_f16_rand_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float*)scratch, iters);
    else hipLaunchKernelGGL(peak_mfma_f16_rand_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float*)scratch, iters);
    if (check_launch("msnet_peak_mfma_f16_rand")) return 0.0;
    return (double)blocks * 4.0 * iters * 8.0 * 2.0 * 32 * 32 * 16;       // both shapes: 8 x 32x32x16 = 16 x 16x16x32 FLOPs per iteration
}

// Per XCD: one thread stores {shader-clock counter (s_memtime: ticks at the clock the power manager currently grants that XCD),
// constant 100 MHz counter (s_memrealtime)} into slot XCC_ID of device_u64x16 (8 XCDs x 2).  The shader-clock counters of the
// XCDs are not one counter, so a probe is a grid of single-thread workgroups -- round-robin dispatch puts some on every XCD --
// and each writes its own XCD's slot.  Several workgroups land on one XCD: the pair is read back to back by ONE thread and
// leaves as ONE 16-byte store, so a slot always holds a {ticks, realtime} pair from the same workgroup (whichever wrote last;
// they are nanoseconds apart).  Two probes on one stream bracket a region: granted clock of XCD i = d(ticks_i) / d(real_i).
typedef unsigned long long u64x2_p __attribute__((ext_vector_type(2)));
__global__ void clock_probe_kernel(unsigned long long* out) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;          // hwreg(HW_REG_XCC_ID), bits 3:0
    u64x2_p v;
    v.x = __builtin_readcyclecounter();
    v.y = __builtin_amdgcn_s_memrealtime();
    __builtin_nontemporal_store(v, reinterpret_cast<u64x2_p*>(out + 2 * xcc));   // global_store_dwordx4: one transaction
}

extern "C" int msnet_clock_probe(void* device_u64x16, msnet_stream_t stream) {
    if (!device_u64x16) return msnet::fail("msnet_clock_probe: null pointer");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(64), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)device_u64x16);
    return msnet::check_launch("msnet_clock_probe");
}

extern "C" int msnet_peak_copy(const void* src, void* dst, size_t bytes, msnet_stream_t stream) {
    if (!src || !dst || bytes < 16) return fail("msnet_peak_copy: bad arguments");
    hipLaunchKernelGGL(peak_copy_kernel, dim3(256 * 16), dim3(256), 0, (hipStream_t)stream, (const f32x4*)src, (f32x4*)dst, bytes / 16);
    return check_launch("msnet_peak_copy");
}

/* returns the FLOPs one call executes (0 on error): waves * iters * 8 MFMAs * 2*32*32*16 */
extern "C" double msnet_peak_mfma_f16(void* scratch, int iters, msnet_stream_t stream) {
    if (!scratch || iters <= 0) { fail("msnet_peak_mfma_f16: bad arguments"); return 0.0; }
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int blocks = cus * 2;                            // 8 waves per CU = 2 per SIMD
    hipLaunchKernelGGL(peak_mfma_f16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float*)scratch, iters, 1.0f);
    if (check_launch("msnet_peak_mfma_f16")) return 0.0;
    return (double)blocks * 4.0 * iters * 8.0 * 2.0 * 32 * 32 * 16;
}

/* the same measurement on v_mfma_f32_16x16x32_f16 (16 per iteration); returns the FLOPs of the call */
extern "C" double msnet_peak_mfma_f16_16x16(void* scratch, int iters, msnet_stream_t stream) {
    if (!scratch || iters <= 0) { fail("msnet_peak_mfma_f16_16x16: bad arguments"); return 0.0; }
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int blocks = cus * 2;
    hipLaunchKernelGGL(peak_mfma_f16_16x16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float*)scratch, iters, 1.0f);
    if (check_launch("msnet_peak_mfma_f16_16x16")) return 0.0;
    return (double)blocks * 4.0 * iters * 16.0 * 2.0 * 16 * 16 * 32;
}

// Does the ORDER in which operand fragments are reused change the fp16 MFMA rate the chip sustains (power management)?
// Register-only loops, 2 waves per SIMD on every CU, eight random A and four random B fragments:
//   mode 0: every MFMA changes both operands          (a[j], b[j & 3])
//   mode 1: A fixed for 4 consecutive MFMAs, B cycles  (a[j >> 2], b[j & 3])
//   mode 2: B fixed for 4 consecutive MFMAs, A cycles  (a[j & 7], b[j >> 2 & 3])
//   mode 3: both fixed (constant operands)
//   mode 4: zeros
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_toggle_test.hip -o tools/bin/mfma_toggle_test
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ half8 rnd(unsigned& st, float sc) {
    half8 v;
    for (int k = 0; k < 8; ++k) { st = st * 1664525u + 1013904223u; v[k] = (_Float16)(((float)(st >> 8) * (1.0f / 16777216.0f) - 0.5f) * sc); }
    return v;
}
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    unsigned st = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    half8 a[8], b[4];
    const float sc = MODE == 4 ? 0.f : 4.f;
    for (int i = 0; i < 8; ++i) a[i] = rnd(st, sc);
    for (int i = 0; i < 4; ++i) b[i] = rnd(st, sc);
    f32x16 acc[8];
    for (int j = 0; j < 8; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ia = MODE == 0 ? j : MODE == 1 ? (j >> 2) : MODE == 2 ? j : 0;
            const int ib = MODE == 0 ? (j & 3) : MODE == 1 ? (j & 3) : MODE == 2 ? (j >> 2) : 0;
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ia], b[ib], acc[j], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int j = 0; j < 8; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
    if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(float* d, const char* name) {
    const int iters = 40000, blocks = 512;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int r = 0; r < 4; ++r) {
        hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (r && ms < best) best = ms;
    }
    printf("%-40s %8.3f ms  %7.1f TFLOP/s\n", name, best, (double)blocks * 4 * iters * 8 * 2.0 * 32 * 32 * 16 / best / 1e9);
}
int main() {
    float* d; hipMalloc(&d, 512 * 256 * 4);
    run<0>(d, "both operands change every MFMA");
    run<1>(d, "A fixed for 4 MFMAs, B cycles");
    run<2>(d, "B fixed for 4 MFMAs, A cycles");
    run<3>(d, "constant operands");
    run<4>(d, "zeros");
    return 0;
}

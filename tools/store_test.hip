// Micro-benchmark: how fast can the four MFMA waves of a workgroup get a finished 32-voxel x 64-channel fp32 block out?
//   mode 0: lane = channel (MFMA D layout with activations as operand A): 32 dword stores per block, 2 voxels x 128 B each
//   mode 1: lane = voxel   (operand roles swapped): 8 dwordx4 stores per block, 64 lanes x 16 B at a 256-byte pitch
// One workgroup per CU, every wave writes `reps` blocks back to back (or with `gap` cycles of s_sleep between blocks); prints the
// average cycles a wave spends ISSUING a block's stores and the kernel's GB/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void store_kernel(float* out, long nblocks, int reps, int gap, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0x7ffffff0, 0x00020000);
    unsigned long long acc = 0;
    for (int it = 0; it < reps; ++it) {
        const long blk = ((long)(blockIdx.x * 4 + wave) + (long)it * gridDim.x * 4) % nblocks;     // 32 voxels x 64 channels = 8 KB
        const unsigned base = (unsigned)(blk * 8192);
        const float v = (float)it;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int vox = (e & 3) + 8 * (e >> 2) + 4 * hh;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v + e), rs, base + (vox * 64 + j * 32 + r) * 4, 0, 0);
                }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ch = j * 32 + 8 * q + 4 * hh;
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    const f32x4 d = {v, v + 1, v + 2, v + 3};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, d), rs, base + (r * 64 + ch) * 4, 0, 0);
                }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        acc += t1 - t0;
        for (int g = 0; g < gap; g += 64) __builtin_amdgcn_s_sleep(1);
    }
    if (lane == 0) atomicAdd(cyc, acc);
}

int main(int argc, char** argv) {
    const long nblocks = 200000;                         // 1.6 GB
    float* out; unsigned long long* cyc;
    hipMalloc(&out, nblocks * 8192); hipMalloc(&cyc, 8);
    for (int gap : {0, 2000, 8000})
        for (int mode = 0; mode < 2; ++mode) {
            const int reps = 400;
            hipMemset(cyc, 0, 8);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int w = 0; w < 2; ++w) {
                if (w == 1) { hipMemset(cyc, 0, 8); hipEventRecord(e0); }
                if (mode == 0) hipLaunchKernelGGL(store_kernel<0>, dim3(256), dim3(256), 0, 0, out, nblocks, reps, gap, cyc);
                else hipLaunchKernelGGL(store_kernel<1>, dim3(256), dim3(256), 0, 0, out, nblocks, reps, gap, cyc);
            }
            hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            printf("gap %5d mode %d: %8.1f cycles to issue one 8 KB block's stores (per wave), kernel %.3f ms = %.0f GB/s\n", gap, mode,
                   (double)h / (256.0 * 4 * reps), ms, 256.0 * 4 * reps * 8192 / ms / 1e6);
        }
    return 0;
}

#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split4(const f32x4 v, half4& hi, half4& lo) {
    unsigned h01, h23, l01, l23;
    float t0, t1, t2, t3;
    const float k = 2048.f;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h01) : "v"(v[0]), "v"(v[1]));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h23) : "v"(v[2]), "v"(v[3]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(t0) : "v"(h01), "v"(v[0]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(t1) : "v"(h01), "v"(v[1]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(t2) : "v"(h23), "v"(v[2]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(t3) : "v"(h23), "v"(v[3]));
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(l01) : "v"(t0), "v"(k));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(l01) : "v"(t1), "v"(k));
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(l23) : "v"(t2), "v"(k));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(l23) : "v"(t3), "v"(k));
    struct U2 { unsigned a, b; };
    hi = __builtin_bit_cast(half4, U2{h01, h23});
    lo = __builtin_bit_cast(half4, U2{l01, l23});
}
__device__ __forceinline__ void split4_ref(const f32x4 v, half4& hi, half4& lo) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const _Float16 h = (_Float16)v[k];
        hi[k] = h;
        lo[k] = (_Float16)((v[k] - (float)h) * 2048.f);
    }
}
__global__ void k(const f32x4* __restrict__ src, half4* __restrict__ dst, half4* __restrict__ ref, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    half4 hi, lo;
    split4(src[i], hi, lo);
    dst[2 * i] = hi; dst[2 * i + 1] = lo;
    split4_ref(src[i], hi, lo);
    ref[2 * i] = hi; ref[2 * i + 1] = lo;
}
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
int main() {
    const int n = 1 << 20;
    float* h = (float*)malloc(n * 16);
    srand(1);
    for (int i = 0; i < 4 * n; ++i) {
        const int m = rand() % 8;
        float v = (float)rand() / RAND_MAX;
        if (m == 0) v = 0.f; else if (m == 1) v *= 6e4f; else if (m == 2) v *= 1e-3f; else if (m == 3) v *= 1e-7f; else if (m == 4) v = -v * 100.f;
        else if (m == 5) { unsigned u = (unsigned)rand() * 2654435761u; u = (u & 0x807fffffu) | ((unsigned)(100 + rand() % 40) << 23); memcpy(&v, &u, 4); }
        h[i] = v;
    }
    float* s; unsigned short *d, *r;
    (void)hipMalloc(&s, n * 16); (void)hipMalloc(&d, n * 16); (void)hipMalloc(&r, n * 16);
    (void)hipMemcpy(s, h, n * 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, (const f32x4*)s, (half4*)d, (half4*)r, n);
    unsigned short *hd = (unsigned short*)malloc(n * 16), *hr = (unsigned short*)malloc(n * 16);
    (void)hipMemcpy(hd, d, n * 16, hipMemcpyDeviceToHost); (void)hipMemcpy(hr, r, n * 16, hipMemcpyDeviceToHost);
    long bad = 0; for (long i = 0; i < 8L * n; ++i) bad += hd[i] != hr[i];
    printf("split4 asm vs reference: %ld of %ld halves differ\n", bad, 8L * n);
    return 0;
}

// Weight repacking into MFMA lane order, and NCDHW <-> NDHWC layout changes at the module boundary.
#include "common.h"

namespace msnet {

// packed[((tap*Ci/8 + ci8)*Co/32 + nb)*64 + lane][t] = W[co = nb*32 + (lane&31)][ci = ci8*8 + 4*(lane>>5) + t][tap]
// (see the operand-map comment at the top of conv3d.hip).  TRANSPOSED selects the ConvTranspose3d
// weight layout [Ci][Co][27] instead of Conv3d's [Co][Ci][27].
template <bool TRANSPOSED>
__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int Ci, int Co) {
    const size_t total = (size_t)27 * Ci * Co;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        size_t i = o;
        const int t = i & 3; i >>= 2;
        const int lane = i & 63; i >>= 6;
        const int nbtot = Co >> 5, nci8 = Ci >> 3;
        const int nb = i % nbtot; i /= nbtot;
        const int ci8 = i % nci8;
        const int tap = (int)(i / nci8);
        const int co = nb * 32 + (lane & 31);
        const int ci = ci8 * 8 + 4 * (lane >> 5) + t;
        const size_t src = TRANSPOSED ? ((size_t)ci * Co + co) * 27 + tap : ((size_t)co * Ci + ci) * 27 + tap;
        out[o] = w[src];
    }
}

// [N][C][S] -> [N][S][C] (S = D*H*W).  One workgroup moves 256 voxels x C channels through LDS so both the
// reads (along S) and the writes (C-contiguous runs) are coalesced.
__global__ __launch_bounds__(256) void ncs_to_nsc_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                         int C, long S, unsigned* oflag) {
    extern __shared__ __attribute__((aligned(16))) float tile[];   // [C][LS], LS chosen bank-conflict-free
    const int LS = 256 + (C < 32 ? 32 / C : 1);
    const long s0 = (long)blockIdx.x * 256;
    const int n = blockIdx.y;
    const int tid = threadIdx.x;
    const int cnt = (int)((S - s0 < 256) ? (S - s0) : 256);
    const float* sp = src + (size_t)n * C * S + s0;
    unsigned amax = 0u;                                    // the module input feeds a split-fp16 layer: range guard on the magnitude BITS (NaN-aware)
    int c = 0;
    for (; c + 8 <= C; c += 8) {                           // eight independent plane loads in flight per thread
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = tid < cnt ? sp[(size_t)(c + u) * S + tid] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) { amax = max(amax, __builtin_bit_cast(unsigned, v[u]) & 0x7fffffffu); tile[(c + u) * LS + tid] = v[u]; }
    }
    for (; c < C; ++c)
        if (tid < cnt) { const float v = sp[(size_t)c * S + tid]; amax = max(amax, __builtin_bit_cast(unsigned, v) & 0x7fffffffu); tile[c * LS + tid] = v; }
    // bit 1: the module INPUT left the range of the split-fp16 kernels (|x| >= 32752, conv_common.h), or is inf / NaN -- a per-call
    // condition (hipops.py)
    if (oflag && amax >= 0x46ffe000u) atomicOr(oflag, 2u);
    __syncthreads();
    float* dp = dst + ((size_t)n * S + s0) * C;
    const int total = cnt * C;
    if ((C & (C - 1)) == 0 && C >= 4) {                    // power-of-two channel counts: shifts instead of divisions, 16-byte stores
        const int sh = __builtin_ctz(C), q = C >> 2;       // q float4 per voxel
        for (int k4 = tid; k4 < (total >> 2); k4 += 256) {
            const int vox = k4 >> (sh - 2), c4 = (k4 & (q - 1)) << 2;
            const f32x4 o = {tile[c4 * LS + vox], tile[(c4 + 1) * LS + vox], tile[(c4 + 2) * LS + vox], tile[(c4 + 3) * LS + vox]};
            reinterpret_cast<f32x4*>(dp)[k4] = o;
        }
    } else {
        for (int k = tid; k < total; k += 256) dp[k] = tile[(k % C) * LS + k / C];
    }
}

__global__ __launch_bounds__(256) void nsc_to_ncs_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                         int C, long S) {
    extern __shared__ __attribute__((aligned(16))) float tile[];   // [C][LS], LS chosen bank-conflict-free
    const int LS = 256 + (C < 32 ? 32 / C : 1);
    const long s0 = (long)blockIdx.x * 256;
    const int n = blockIdx.y;
    const int tid = threadIdx.x;
    const int cnt = (int)((S - s0 < 256) ? (S - s0) : 256);
    const float* sp = src + ((size_t)n * S + s0) * C;
    const int total = cnt * C;
    for (int k = tid; k < total; k += 256) tile[(k % C) * LS + k / C] = sp[k];
    __syncthreads();
    float* dp = dst + (size_t)n * C * S + s0;
    for (int c = 0; c < C; ++c)
        if (tid < cnt) dp[(size_t)c * S + tid] = tile[c * LS + tid];
}

// Read-only range check of a channels-last module input (what ncs_to_nsc_kernel does on the way for an NCDHW input): raises
// bit 1 of the overflow word if any |x| >= 32752 or any x is inf / NaN.  n4 = number of float4.
typedef unsigned u32x4_p __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned mag4(u32x4_p v) {
    v &= 0x7fffffffu;
    return max(max(v.x, v.y), max(v.z, v.w));
}
// (edge: up to 3 floats in front of the first 16-byte boundary and up to 3 behind the last whole float4, scalar reads by block 0)
__global__ __launch_bounds__(256) void input_range_kernel(const u32x4_p* __restrict__ x, size_t n4, unsigned* oflag,
                                                          const unsigned* __restrict__ head, int nhead,
                                                          const unsigned* __restrict__ tail, int ntail) {
    unsigned amax = 0u;
    if (blockIdx.x == 0) {
        if ((int)threadIdx.x < nhead) amax = head[threadIdx.x] & 0x7fffffffu;
        if ((int)threadIdx.x < ntail) amax = max(amax, tail[threadIdx.x] & 0x7fffffffu);
    }
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {         // four independent 16-byte loads in flight per thread
        const u32x4_p v0 = __builtin_nontemporal_load(x + i), v1 = __builtin_nontemporal_load(x + i + stride);
        const u32x4_p v2 = __builtin_nontemporal_load(x + i + 2 * stride), v3 = __builtin_nontemporal_load(x + i + 3 * stride);
        amax = max(max(amax, mag4(v0)), max(max(mag4(v1), mag4(v2)), mag4(v3)));
    }
    for (; i < n4; i += stride) amax = max(amax, mag4(x[i]));
    if (oflag && amax >= 0x46ffe000u) atomicOr(oflag, 2u);
}

}  // namespace msnet

using namespace msnet;

extern "C" int msnet_check_input_range(const float* x, size_t count, msnet_stream_t stream) {
    if (!x) return fail("msnet_check_input_range: null pointer");
    if (count == 0 || ((uintptr_t)x & 3)) return fail("msnet_check_input_range: %zu floats at %p (needs a non-empty float array)", count, (const void*)x);
    unsigned* of = overflow_flag();
    if (!of) return 0;                                     // no guard registered on this thread: nothing to report to
    hipStream_t s = (hipStream_t)stream;
    // any length, any float alignment: scalar reads up to the first 16-byte boundary and behind the last whole float4
    size_t nhead = ((16 - ((uintptr_t)x & 15)) & 15) / 4;
    if (nhead > count) nhead = count;
    const size_t n4 = (count - nhead) / 4, ntail = count - nhead - 4 * n4;
    const float* body = x + nhead;
    const unsigned blocks = (unsigned)((n4 + 1023) / 1024 < 4096 ? (n4 + 1023) / 1024 : 4096);
    LaunchScope ls("input_range_check", s, 0, 4.0 * count);
    hipLaunchKernelGGL(input_range_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s, reinterpret_cast<const u32x4_p*>(body), n4, of,
                       reinterpret_cast<const unsigned*>(x), (int)nhead, reinterpret_cast<const unsigned*>(body + 4 * n4), (int)ntail);
    return check_launch("msnet_check_input_range");
}

extern "C" size_t msnet_packed_weight_floats(int Ci, int Co) { return (size_t)27 * Ci * Co; }

static int pack_common(bool transposed, const float* w, float* packed, int Ci, int Co, msnet_stream_t stream) {
    if (!w || !packed) return fail("msnet_pack_*_weight: null pointer");
    if (Ci <= 0 || Ci % 8 != 0) return fail("msnet_pack_*_weight: Ci=%d must be a positive multiple of 8", Ci);
    if (Co <= 0 || Co % 32 != 0) return fail("msnet_pack_*_weight: Co=%d must be a positive multiple of 32", Co);
    const size_t total = (size_t)27 * Ci * Co;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipStream_t s = (hipStream_t)stream;
    LaunchScope ls("pack_weight", s, 0, 8.0 * total);
    if (transposed) hipLaunchKernelGGL(pack_weight_kernel<true>, dim3(blocks), dim3(256), 0, s, w, packed, Ci, Co);
    else            hipLaunchKernelGGL(pack_weight_kernel<false>, dim3(blocks), dim3(256), 0, s, w, packed, Ci, Co);
    return check_launch("pack_weight");
}

extern "C" int msnet_pack_conv_weight(const float* w, float* packed, int Ci, int Co, msnet_stream_t stream) {
    return pack_common(false, w, packed, Ci, Co, stream);
}
extern "C" int msnet_pack_deconv_weight(const float* w, float* packed, int Ci, int Co, msnet_stream_t stream) {
    return pack_common(true, w, packed, Ci, Co, stream);
}

static int layout_common(bool to_cl, const float* src, float* dst, int N, int C, int D, int H, int W,
                         msnet_stream_t stream) {
    if (!src || !dst) return fail("msnet layout: null pointer");
    if (N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0) return fail("msnet layout: empty tensor");
    if (C > 128) return fail("msnet layout: C=%d > 128 unsupported", C);
    if (N > 65535) return fail("msnet layout: N=%d > 65535", N);
    const long S = (long)D * H * W;
    const size_t lds = (size_t)C * (256 + (C < 32 ? 32 / C : 1)) * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)((S + 255) / 256), (unsigned)N);
    LaunchScope ls(to_cl ? "ncdhw_to_ndhwc" : "ndhwc_to_ncdhw", s, 0, 8.0 * N * C * (double)S);
    if (to_cl) {
        if (lds > 65536) (void)hipFuncSetAttribute((const void*)ncs_to_nsc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(ncs_to_nsc_kernel, grid, dim3(256), lds, s, src, dst, C, S, overflow_flag());
    } else {
        if (lds > 65536) (void)hipFuncSetAttribute((const void*)nsc_to_ncs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(nsc_to_ncs_kernel, grid, dim3(256), lds, s, src, dst, C, S);
    }
    return check_launch("msnet layout");
}

extern "C" int msnet_ncdhw_to_ndhwc(const float* src, float* dst, int N, int C, int D, int H, int W, msnet_stream_t stream) {
    return layout_common(true, src, dst, N, C, D, H, W, stream);
}
extern "C" int msnet_ndhwc_to_ncdhw(const float* src, float* dst, int N, int C, int D, int H, int W, msnet_stream_t stream) {
    return layout_common(false, src, dst, N, C, D, H, W, stream);
}

// Shared host-side plumbing for libmsnet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/msnet_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace msnet {

// RAND_MAX as the reference's matchers use it for "never written" entries
// (matchers.cpp:65,251,377,462): std::fill_n(float*, n, RAND_MAX) stores (float)2147483647 = 2^31.
constexpr float kSentinel = 2147483648.0f;

void set_error(const char* fmt, ...);
bool exact_tails();               // msnet_set_exact_tails(): tails on fp32 VALU arithmetic instead of the split-fp16 MFMA
unsigned* overflow_flag();       // the calling thread's msnet_set_overflow_flag() pointer (device memory) or null
int  fail(const char* fmt, ...);   // set_error + return 1

// Profiling hooks (api.cpp).  Each kernel launch goes through LaunchScope so that, when profiling
// is enabled, a start/stop hipEvent pair on the launch stream brackets exactly that launch.
struct LaunchScope {
    // attach = true: the scope brackets exactly ONE kernel, launched through MSNET_LAUNCH below: its start / stop events ride on that
    // kernel's own dispatch packet (hipExtLaunchKernelGGL) instead of being recorded as two barrier packets around it, so a timed
    // launch no longer keeps the next kernel from starting under its drain (the ten recorded pairs of a default bench step cost
    // 1.2 % of the step) and the pair measures the kernel itself, as rocprofv3 does.
    LaunchScope(const char* name, hipStream_t s, double flops, double bytes, bool attach = false);
    ~LaunchScope();
    bool events(hipEvent_t* start, hipEvent_t* stop) const;       // attach mode, profiling on: the pair to hand to the launch
    const char* name; hipStream_t stream; void* rec; bool attach;
    mutable bool launched = false;      // attach mode: events() handed the pair to a launch (else the pair was never recorded)
};
// one kernel under an attach-mode scope: hipExtLaunchKernelGGL with the scope's events when this launch is being timed
#define MSNET_LAUNCH(ls, kernel, grid, block, lds, stream, ...)                                                        \
    do {                                                                                                               \
        hipEvent_t ea_ = nullptr, eb_ = nullptr;                                                                       \
        if ((ls).events(&ea_, &eb_)) hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, ea_, eb_, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                        \
    } while (0)

// The likelihood numerator exp(-(c - m)^2 / sigma) (featextract.cpp:444-449) as v_exp_f32((c - m)^2 * k), k = RN(-log2(e) / sigma)
// formed once per launch in double.  The library route -- IEEE division by sigma (or its three-operation equivalent), then
// expf's compensated product, rndne / ldexp range reduction and two range selects -- is 18 VALU instructions per evaluation, two
// evaluations per likelihood element in the channels-last build; this is 4.  What it costs: three roundings on the exponent
// instead of none, i.e. a RELATIVE error of about |x| * 2e-7 on e^x with x <= 0 -- at most 7e-8 ABSOLUTE (at x = -1) -- plus
// v_exp_f32's own ulp, against the 2e-6 the likelihood channels are held to (GPU and glibc expf are different <= 1 ulp
// implementations to begin with; tests/test_gpu_volume.py prints the measured maximum).  x <= 0 cannot overflow, a result below
// 2^-126 is 0 either way, exp(-0) = 1 exactly (so den >= 1 still holds), and a sentinel cost gives exp2(-3e20) = 0 as before.
// Every kernel that forms a likelihood uses these two functions, so the fast, generic and per-op paths agree bit for bit.
__host__ __device__ inline float aml_scale(float sigma) { return (float)(-1.44269504088896340736 / (double)sigma); }
__device__ __forceinline__ float aml_numerator(float c, float m, float k) {
    const float n = c - m;
    return __builtin_amdgcn_exp2f((n * n) * k);
}

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("%s: launch failed: %s", what, hipGetErrorString(e));
    return 0;
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Bijective XCD remap (cdna_hip_programming.md T1): blocks b and b+8 share an XCD under round-robin
// dispatch; give each XCD a contiguous run of logical tiles so halo re-reads of neighbouring tiles
// hit that XCD's L2.  Speed only -- any placement is correct.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
    const unsigned q = nwg >> 3, r = nwg & 7u, xcd = bid & 7u, k = bid >> 3;
    const unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

}  // namespace msnet

// Shared host-side plumbing for libmsnet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
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
    LaunchScope(const char* name, hipStream_t s, double flops, double bytes);
    ~LaunchScope();
    const char* name; hipStream_t stream; void* rec;
};

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

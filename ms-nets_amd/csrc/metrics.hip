// Driver-side metric on the device: the end-point error and bad-x rate the reference's test loop
// computes on the host after copying every disparity map back (main_msnet.py:597-618, get_epe_rate :708-713):
//     mask = (gt >= 0.001) & (gt <= max_disp);  err = |pred - gt|[mask];  epe = mean(err);  rate = count(err > thr) / count(mask)
// One pass over the two maps, HBM-bound (8 bytes per pixel); the three sums leave the device as doubles.
#include "common.h"

namespace msnet {

__global__ __launch_bounds__(256) void epe_badx_kernel(const float* __restrict__ gt, const float* __restrict__ pred, size_t n,
                                                       float max_disp, float thr, double* __restrict__ out3) {
    double s = 0.0;
    unsigned bad = 0, valid = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float g = gt[i];
        if (g >= 0.001f && g <= max_disp) {
            const float e = fabsf(pred[i] - g);
            s += (double)e;
            bad += e > thr;
            ++valid;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o);
        bad += __shfl_xor(bad, o);
        valid += __shfl_xor(valid, o);
    }
    if ((threadIdx.x & 63) == 0 && valid) {
        atomicAdd(out3, s);
        atomicAdd(out3 + 1, (double)bad);
        atomicAdd(out3 + 2, (double)valid);
    }
}

}  // namespace msnet

using namespace msnet;

extern "C" int msnet_epe_badx(const float* gt, const float* pred, size_t n, float max_disp, float threshold, double* out3,
                              msnet_stream_t stream) {
    if (!gt || !pred || !out3) return fail("msnet_epe_badx: null pointer");
    if (n == 0) return fail("msnet_epe_badx: empty maps");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(out3, 0, 3 * sizeof(double), s) != hipSuccess) return fail("msnet_epe_badx: memset failed");
    const size_t want = (n + 255) / 256;
    const int blocks = (int)(want < 2048 ? want : 2048);
    LaunchScope ls("epe_badx", s, 0, 8.0 * n);
    hipLaunchKernelGGL(epe_badx_kernel, dim3(blocks), dim3(256), 0, s, gt, pred, n, max_disp, threshold, out3);
    return check_launch("msnet_epe_badx");
}

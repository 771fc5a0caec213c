/*
 * msnet_hip.h -- C ABI of libmsnet_hip.so: the MI355X (gfx950) implementation of the MS-Nets
 * cost-volume forward pass (matching-space volume build -> 3D conv aggregator -> soft-argmin).
 *
 * The reference (ccj5351/MS-Nets) has no C ABI for this path: its boundary is two Python surfaces,
 * (1) the Boost.Python extension modules libmatchers / libfeatextract and (2) the two aggregator
 * nn.Modules.  Every entry point below names the reference interface it replaces (paths relative to
 * the reference root).  The Python mirror of those surfaces lives in ms-nets_amd/ and calls this
 * library through ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); nothing synchronises;
 *   - no entry point allocates: callers own all buffers, sizes are given by the *_bytes / *_floats helpers;
 *   - return value: 0 on success, non-zero on error; msnet_last_error() returns a static thread-local
 *     message for the last failure on the calling thread;
 *   - activations inside the aggregator are channels-last fp32 ("NDHWC": [N][D][H][W][C]);
 *     the reference's NCDHW tensors are converted at the module boundary by msnet_ncdhw_to_ndhwc.
 */
#ifndef MSNET_HIP_H
#define MSNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* msnet_stream_t;

/* ---- library -------------------------------------------------------------------------------- */
int         msnet_version(void);                 /* ABI version, currently 1 */
const char* msnet_last_error(void);
/* Per-kernel timing with HIP events on the launch stream (bench.py roofline).  enable(1) makes
 * every launch record start/stop events; collect() synchronises those events and writes one line
 * "name calls total_ms flops bytes\n" per kernel into buf (returns bytes written, <0 on error) and
 * clears the log. */
int         msnet_prof_enable(int on);
/* Restrict the timing to launches whose kernel-family name starts with one of the comma-separated prefixes in
 * `name_prefix` (NULL or "" = all): two event
 * records per launch cost ~2 us of host time each, 0.18 ms per forward when all ~45 launches are timed. */
int         msnet_prof_select(const char* name_prefix);
long        msnet_prof_collect(char* buf_host, size_t buf_bytes);

/* Range guard of the split-fp16 kernels.  Their operands' `hi` halves are fp16, and the Winograd-depth kernel splits sums and
 * differences of two activations, so an activation must stay below 32752 (half the largest finite fp16; BN-folded weights:
 * below 65504, checked by the caller when it packs them).  While a device word is registered (per calling thread; NULL
 * unregisters) every conv / transposed-conv epilogue ORs bit 0 into it when it stores a magnitude >= 32752 (or inf), and the
 * NCDHW -> NDHWC conversion of the module input and the NCDHW first-layer loader OR bit 1 into it when they read a magnitude
 * >= 32752, an inf or a NaN:
 * bit 0 = an activation (conv / transposed-conv epilogues), bit 1 = the module input (layout conversion, first-layer loader).
 * The caller zeroes the word, runs the layers, reads it back; ms-nets_amd/hipops.py re-runs the forward on the exact
 * fp32 kernels when it is set.  No reference counterpart: torch's fp32 conv has no such range limit. */
int msnet_set_overflow_flag(void* device_u32);
/* 1: msnet_deconv5_softargmin and msnet_conv3d_k3_cout1 contract the channels with fp32 FMAs (exact fp32 products, no fp16
 * range limit) instead of the split-fp16 MFMA; per calling thread; default 0. */
int msnet_set_exact_tails(int on);

/* Measured-attainable peaks of this device for bench.py's roofline denominators; diagnostics, not on
 * the product path.  msnet_peak_copy: one float4 copy of `bytes` bytes (2 * bytes of HBM traffic).  msnet_peak_mfma_f16:
 * a full-chip grid of waves issuing `iters` x 8 v_mfma_f32_32x32x16_f16 on registers; returns the FLOPs of the call
 * (0 on error); `scratch` is any device buffer of >= 1 MiB.  Time both with events on `stream`. */
int    msnet_peak_copy(const void* src, void* dst, size_t bytes, msnet_stream_t stream);
double msnet_peak_mfma_f16(void* scratch, int iters, msnet_stream_t stream);
double msnet_peak_mfma_f16_16x16(void* scratch, int iters, msnet_stream_t stream);
/* The same rate on CHANGING operands (eight pseudo-random A and four B fragments cycled in registers): what the chip's power
 * management sustains when every MFMA sees fresh data, as in a real kernel.  shape16: 0 = 32x32x16, 1 = 16x16x32. */
double msnet_peak_mfma_f16_rand(void* scratch, int iters, int shape16, msnet_stream_t stream);   /* same, v_mfma_f32_16x16x32_f16 */
/* Measurement aid: a grid of one-thread workgroups stores, per XCD i, {shader-clock ticks (s_memtime), constant 100 MHz ticks
 * (s_memrealtime)} into device_u64x16[2i], [2i+1] (zero-initialise it; a slot stays zero if no workgroup landed on that XCD).
 * Two probes on one stream around a region give the clock the power manager GRANTED each XCD over it:
 * d(ticks) / (d(real) / 1e8) Hz -- bench.py's `power.sclk_mhz` (the sysfs clock files of this pool read idle values under load). */
int    msnet_clock_probe(void* device_u64x16, msnet_stream_t stream);

/* ---- matchers: replaces src/cpp/matchers/matchers.cpp:565-580 (libmatchers) ----------------- */
/* census(left,right,ndisp,wsize) matchers.cpp:232-353.  l,r: u8[H][W]; out: f32[H][W][ndisp];
 * workspace: msnet_census_workspace_bytes(H,W,wsize) bytes (the two census bit images). */
int    msnet_census(const uint8_t* l, const uint8_t* r, float* out, void* workspace, int H, int W, int ndisp,
                    int wsize, msnet_stream_t stream);
size_t msnet_census_workspace_bytes(int H, int W, int wsize);
/* nccNister(left,right,ndisp,wsize) matchers.cpp:47-228.  out: f32[ndisp][H][W]. */
int msnet_ncc(const uint8_t* l, const uint8_t* r, float* out, int H, int W, int ndisp, int wsize,
              msnet_stream_t stream);
/* zsad(left,right,ndisp,wsize) matchers.cpp:442-512.  out: f32[ndisp][H][W]. */
int msnet_zsad(const uint8_t* l, const uint8_t* r, float* out, int H, int W, int ndisp, int wsize,
               msnet_stream_t stream);
/* sobel(img) matchers.cpp:515-554.  img: u8[H][W]; out: f32[H][W]. */
int msnet_sobel(const uint8_t* img, float* out, int H, int W, msnet_stream_t stream);
/* sadsob(sobl,sobr,ndisp,wsize) matchers.cpp:356-438.  sl,sr: f32[H][W]; out: f32[ndisp][H][W];
 * workspace: msnet_sadsob_workspace_bytes(H,W,ndisp) bytes. */
int    msnet_sadsob(const float* sl, const float* sr, float* out, void* workspace, int H, int W,
                    int ndisp, int wsize, msnet_stream_t stream);
size_t msnet_sadsob_workspace_bytes(int H, int W, int ndisp);

/* ---- featextract: replaces src/cpp/featextract/featextract.cpp:529-553 (libfeatextract) ----- */
/* swap_axes(cost) featextract.cpp:49-76.  in: f32[D][H][W] -> out: f32[H][W][D]. */
int msnet_swap_axes(const float* in, float* out, int D, int H, int W, msnet_stream_t stream);
/* featextract.cpp:136-172 get_right_cost (only reached by extract_features_lr, cbmv_generator.py:84-254, i.e. with
 * is_left_only=False): out[i][j][d] = cost[i][j+d][d] for j < W-d, else cost[0][0][0].  cost, out: f32[H][W][D], distinct. */
int msnet_get_right_cost(const float* cost, float* out, int H, int W, int D, msnet_stream_t stream);
/* extract_likelihood(vol,sigma) = extract_aml_testing, featextract.cpp:415-462.  vol,out: f32[P][D]. */
int msnet_extract_likelihood(const float* vol, float* out, long P, int D, float sigma,
                             msnet_stream_t stream);

/* extract_features_left(census,ncc,sobel,sad,...) cbmv_generator.py:258-308 on the reference's row layout:
 * four raw costs f32[P][D] (P = H'*W') -> out f32[8][D][P] = [8][D][H'][W']: channels 0-3 the clipped/normalised
 * costs (:283-287), 4-7 their likelihoods (:301-304; the Sobel channel uses sad_sigma, as the reference does). */
int msnet_extract_features_left(const float* census, const float* ncc, const float* sobel, const float* sad,
                                float* out, long P, int D, float cens_sigma, float ncc_sigma, float sad_sigma,
                                msnet_stream_t stream);

/* ---- fused volume build: replaces get_costs + extract_features_left,
 *      src/dataloader/cbmv_generator.py:27-79 and :258-308 ------------------------------------- */
typedef struct msnet_volume_params {
    int   censw, nccw, sadw, sobelw;       /* 11, 3, 5, 5  (cbmv_generator.py:434-462) */
    float cens_sigma, ncc_sigma, sad_sigma; /* 128, 0.02, 20000 (sobel channel uses sad_sigma, :298,303) */
    int   border_h, border_w;              /* 10, 10 (cbmv_generator.py:819-823) */
} msnet_volume_params;
void   msnet_volume_default_params(msnet_volume_params* p_host);
/* l,r: u8[Hb][Wb] bordered images (Hb = H'+2*border_h, Wb = W'+2*border_w); ndisp = D';
 * out: f32[8][D'][H'][W'] -- the reference's feature layout (cbmv_generator.py:307-308). */
int    msnet_build_volume(const uint8_t* l, const uint8_t* r, int Hb, int Wb, int ndisp,
                          const msnet_volume_params* p_host, void* workspace, float* out,
                          msnet_stream_t stream);
size_t msnet_build_volume_workspace_bytes(int Hb, int Wb, int ndisp);
/* The same volume written channels-last, out: f32[D'][H'][W'][8] (channel order unchanged: cbmv_generator.py:283-287,
 * 301-304) -- the aggregator kernels' own input layout, so GCNet_CostVolumeAggre.forward_ndhwc needs no layout pass between
 * the build and its first conv (msnet_conv3d_k3_c8_in_f16s).  Bit-identical to msnet_build_volume followed by
 * msnet_ncdhw_to_ndhwc.  Built for the reference's own parameters only (windows 11/3/5/5, cbmv_generator.py:434-462;
 * borders >= 6; D' % 8 == 0, D' <= 96): msnet_build_volume_ndhwc_supported() says whether a shape is taken.  Same workspace. */
int    msnet_build_volume_ndhwc_supported(int Hb, int Wb, int ndisp, const msnet_volume_params* p_host);
int    msnet_build_volume_ndhwc(const uint8_t* l, const uint8_t* r, int Hb, int Wb, int ndisp,
                                const msnet_volume_params* p_host, void* workspace, float* out,
                                msnet_stream_t stream);

/* ---- test-time pre-processing: src/dataloader/cbmv_generator.py:780-788 (pad top/right to a multiple of
 *      encoder_ds), :465-482 (down_sampling_input = skimage rescale by 1/ds with anti-aliasing, x255 -> uint8), :819-823
 *      (`border`-pixel zero border).  img: u8[h][w] grayscale; out: u8[Hb][Wb] with the shape from
 *      msnet_preprocess_out_shape; workspace: >= 4 bytes of device memory; taps_host: the 2*int(4*sigma+0.5)+1 gaussian
 *      taps for sigma = (ds-1)/2 as scipy.ndimage computes them (host pointer), or NULL to compute them with libm.
 *      ds = 1 copies.  The rescale restates skimage.transform.resize (parity unpinned: scikit-image is not available). */
int msnet_preprocess_out_shape(int h, int w, int encoder_ds, int ds, int border, int* Hb, int* Wb);
int msnet_preprocess_image(const uint8_t* img, int h, int w, int encoder_ds, int ds, int border,
                           const double* taps_host, uint8_t* out, void* workspace, msnet_stream_t stream);

/* ---- aggregator building blocks: replace the torch.nn calls inside
 *      src/models/gcnet_3dcnn.py:20-27,97-141 and src/models/psmnet_3dcnn.py:22-25,69-89,126-179 */
int msnet_ncdhw_to_ndhwc(const float* src, float* dst, int N, int C, int D, int H, int W,
                         msnet_stream_t stream);
int msnet_ndhwc_to_ncdhw(const float* src, float* dst, int N, int C, int D, int H, int W,
                         msnet_stream_t stream);
/* Range check of a module input that already IS channels-last (no conversion pass to carry it): one read of `count` floats
 * (any count >= 1, any float alignment: whole float4 reads between scalar edges); raises bit 1 of the calling thread's overflow word (msnet_set_overflow_flag) when a value
 * is outside the split-fp16 kernels' range (|x| >= 32752) or not finite.  A no-op without a registered word.  Entry of
 * PSMNet_CostVolumeAggre.forward_ndhwc for the reference's 64-plane volume (psmnet_3dcnn.py:126-131). */
int msnet_check_input_range(const float* x, size_t count, msnet_stream_t stream);

/* Weight repacking into the MFMA operand order used by the conv kernels.
 * conv:   w f32[Co][Ci][3][3][3] (nn.Conv3d.weight)           -> packed, msnet_packed_weight_floats(Ci,Co) floats
 * deconv: w f32[Ci][Co][3][3][3] (nn.ConvTranspose3d.weight)  -> same packed size */
size_t msnet_packed_weight_floats(int Ci, int Co);
int    msnet_pack_conv_weight(const float* w, float* packed, int Ci, int Co, msnet_stream_t stream);
int    msnet_pack_deconv_weight(const float* w, float* packed, int Ci, int Co, msnet_stream_t stream);

/* convbn_3d (+ReLU, + residual): y = act( conv3d_k3_p1_stride(x) * scale[co] + shift[co] (+ residual) ).
 * gcnet_3dcnn.py:20-22, psmnet_3dcnn.py:22-25.  x: NDHWC [N][D][H][W][Ci]; y/residual: [N][OD][OH][OW][Co],
 * OD = (D-1)/stride+1 etc.  scale/shift: the eval-mode BatchNorm3d affine (NULL = identity / zero).
 * Ci in {8} or a multiple of 16; Co a multiple of 32; stride 1 or 2. */
int msnet_conv3d_k3(const float* x, const float* wpk, const float* scale, const float* shift,
                    const float* residual, float* y, int N, int D, int H, int W, int Ci, int Co,
                    int stride, int relu, msnet_stream_t stream);
/* Split-fp16 fast path of msnet_conv3d_k3 (same contract, fp32 tensors in HBM): every operand is used as
 * hi + lo*2^-11 (two fp16 halves, 22 significand bits) and each product costs three fp16 MFMAs
 * (ms-nets_amd/csrc/conv3d_f16s.hip).  wpk_f16s comes from msnet_pack_conv_weight_f16s (same byte count as the
 * fp32 packing: msnet_packed_weight_floats(max(Ci,16),Co) floats) for the same (Ci, Co, stride) it will be used with.
 * msnet_conv3d_k3_f16s_supported() tells whether a (Ci, Co, stride) has a split-fp16 kernel (else use msnet_conv3d_k3). */
int msnet_pack_conv_weight_f16s(const float* w, void* packed, int Ci, int Co, int stride, msnet_stream_t stream);
int msnet_conv3d_k3_f16s_supported(int Ci, int Co, int stride);
int msnet_conv3d_k3_f16s(const float* x, const void* wpk_f16s, const float* scale, const float* shift,
                         const float* residual, float* y, int N, int D, int H, int W, int Ci, int Co,
                         int stride, int relu, msnet_stream_t stream);
/* Split-fp16 fast path of msnet_deconv3d_k3s2 (same contract); supported for Ci = 64, Co in {32, 64}. */
int msnet_deconv3d_k3s2_f16s_supported(int Ci, int Co);
int msnet_pack_deconv_weight_f16s(const float* w, void* packed, int Ci, int Co, msnet_stream_t stream);
int msnet_deconv3d_k3s2_f16s(const float* x, const void* wpk_f16s, const float* scale, const float* shift,
                             const float* residual, float* y, int N, int D, int H, int W, int Ci, int Co,
                             int relu, msnet_stream_t stream);
/* deconvbn_3d: ConvTranspose3d(k3,s2,p1,op1) + BN (+ residual) (+ReLU).  gcnet_3dcnn.py:24-27,
 * psmnet_3dcnn.py:41-44,63-67.  x: [N][D][H][W][Ci] -> y: [N][2D][2H][2W][Co]. */
int msnet_deconv3d_k3s2(const float* x, const float* wpk, const float* scale, const float* shift,
                        const float* residual, float* y, int N, int D, int H, int W, int Ci, int Co,
                        int relu, msnet_stream_t stream);
/* Winograd F(2,3) along depth for the 32 -> 32 stride-1 convbn_3d layers (gcnet_3dcnn.py:101 conv3dbn_2; psmnet_3dcnn.py:94-101
 * dres0 / dres1 and the hourglass 32 -> 32 layers): two thirds of the MFMAs of msnet_conv3d_k3_f16s for the same result
 * (F(2,3) is exact in real arithmetic; with 22-bit split operands a layer stays within 5e-6 of the fp64 conv).
 * g36: f32 [32][32][36], the transformed BN-folded weights -- tap T = k*9 + kh*3 + kw, g0 = w[kd=0], g1 = (w0+w1+w2)/2,
 * g2 = (w0-w1+w2)/2, g3 = w[kd=2] (the caller computes them in fp64); packed: 36*32*32*2 fp16 = 73,728 bytes.
 * _supported: 1 when the shape is taken (stride 1, 32 -> 32, not a small layer). */
int msnet_pack_conv_weight_wd_f16s(const float* g36, void* packed, msnet_stream_t stream);
int msnet_conv3d_k3_wd_f16s_supported(int D, int H, int W, int Ci, int Co, int stride);
int msnet_conv3d_k3_wd_f16s(const float* x, const void* wpk_wd, const float* scale, const float* shift,
                            const float* residual, float* y, int N, int D, int H, int W, int relu,
                            msnet_stream_t stream);
/* Experiment entry: the same kernel on 32-channel slices of wider channels-last tensors -- x / y /
 * residual point at the slice's first channel inside records of x_channels / y_channels floats.  With it a
 * 64 -> 64 layer (gcnet_3dcnn.py:30-44, convs 2 and 3 of a Conv3DBlock) runs as four Winograd-depth launches, the partial sum of
 * the first input half handed to the second through `residual`.  Measured slower than the direct kernel; not on the default path. */
int msnet_conv3d_k3_wd_f16s_strided(const float* x, const void* wpk_wd, const float* scale, const float* shift,
                                    const float* residual, float* y, int N, int D, int H, int W, int x_channels,
                                    int y_channels, int relu, msnet_stream_t stream);
/* The first layer (cbmv_in_planes = 8, gcnet_3dcnn.py:99-101) read straight from the module's NCDHW volume
 * x: f32[N][8][D][H][W] (the layout cbmv_generator.py:307-308 produces) -> y: NDHWC f32[N][D][H][W][Co], Co = 32 or 64, stride 1,
 * no residual; split-fp16 MFMA.  Replaces msnet_ncdhw_to_ndhwc + msnet_conv3d_k3_f16s for that layer: the volume is not
 * copied.  wpk_f16s as for msnet_conv3d_k3_f16s (Ci = 8).  Out-of-range INPUT values raise bit 1 of the overflow word. */
int msnet_conv3d_k3_c8_ncdhw_f16s(const float* x_ncdhw, const void* wpk_f16s, const float* scale, const float* shift,
                                  float* y, int N, int D, int H, int W, int Co, int relu, msnet_stream_t stream);
/* The same first layer on a channels-last module input x: f32[N][D][H][W][8] (what msnet_build_volume_ndhwc writes) ->
 * y: NDHWC f32[N][D][H][W][Co].  msnet_conv3d_k3_f16s with Ci = 8 plus the range check of the module INPUT (bit 1 of the
 * overflow word), which the layout-conversion pass carries on the NCDHW route.  Entry of GCNet_CostVolumeAggre.forward_ndhwc. */
int msnet_conv3d_k3_c8_in_f16s(const float* x_ndhwc, const void* wpk_f16s, const float* scale, const float* shift,
                               float* y, int N, int D, int H, int W, int Co, int relu, msnet_stream_t stream);
/* Stride-1 conv (no residual) on a channels-last module input of ANY supported width, x: f32[N][D][H][W][Ci] ->
 * y: NDHWC f32[N][D][H][W][Co], with the range check of the module INPUT (bit 1 of the overflow word).  Entry of
 * PSMNet_CostVolumeAggre.forward_ndhwc for dres0.0 on the reference's 64-plane volume (psmnet_3dcnn.py:96-99,126-131): the
 * tiled Co = 32 kernel checks the values its loaders stage, so the volume is read once; shapes that kernel does not take run
 * msnet_check_input_range + msnet_conv3d_k3_f16s.  Same bits as msnet_conv3d_k3_f16s. */
int msnet_conv3d_k3_in_f16s(const float* x_ndhwc, const void* wpk_f16s, const float* scale, const float* shift, float* y,
                            int N, int D, int H, int W, int Ci, int Co, int relu, msnet_stream_t stream);
/* Conv3d(Ci->1, k3, p1, bias=False) head (psmnet_3dcnn.py:112-122 classif*.2), optional "+ add"
 * (cost2 = classif2(out2) + cost1, :146-147).  x: NDHWC; w: f32[1][Ci][3][3][3]; y/add: f32[N][D][H][W].
 * y = wscale * conv(x, w) (+ add): the caller may hand over the weights multiplied by a power of two (so that their fp16
 * `hi` halves are normal numbers on the split-fp16 MFMA) and undo it with wscale; 1 for plain weights. */
int msnet_conv3d_k3_cout1(const float* x, const float* w, float wscale, const float* add, float* y, int N, int D,
                          int H, int W, int Ci, msnet_stream_t stream);

/* ---- tails ---------------------------------------------------------------------------------- */
/* softmax over D + sum_d d*p_d.  gcnet_3dcnn.py:126-141.  logits f32[N][D][H][W] -> disp f32[N][H][W]. */
int msnet_softargmin(const float* logits, float* disp, int N, int D, int H, int W, msnet_stream_t stream);
/* deconv5 (ConvTranspose3d Ci->1, k3, s2, p1, op1, bias) fused with the soft-argmin: the [N][2D][2H][2W]
 * logit volume is never written.  gcnet_3dcnn.py:124-141.  x: NDHWC [N][D][H][W][Ci];
 * w: f32[Ci][1][3][3][3]; bias_host: the scalar bias; disp: f32[N][2H][2W].
 * logits = wscale * deconv(x, w) + bias: wscale undoes a power-of-two pre-scale of w (see msnet_conv3d_k3_cout1); 1 for plain weights. */
int msnet_deconv5_softargmin(const float* x, const float* w, float bias_host, float wscale, float* disp, int N, int D,
                             int H, int W, int Ci, msnet_stream_t stream);
/* The same tail with a caller-owned workspace (msnet_deconv5_softargmin_workspace_bytes(N, D, H, W) bytes; 0 = not needed): a
 * tile's D slices are then cut into depth segments, one workgroup each, whose online-softmax states (max, sum e, sum d*e per
 * output pixel) meet in a small merge pass -- at the benchmark shapes the single chain per tile leaves the CUs 2.4 workgroups
 * each and is latency-bound.  The segmentation depends on (D, H, W) only -- never on N (a batch returns its samples' single-forward
 * bits) and never on the CU count the runtime reports (fixed 256-CU sizing: the same bits on every gfx950 box, partition mode and
 * CU-masked stream).  Test hook, not an interface: the environment variable MSNET_TAIL_SEGS=<n> forces n segments
 * (tests/test_gpu_aggregators.py::test_fused_tail_depth_segments); leave it unset anywhere results must reproduce.  Same logits, same order of pushes inside a segment; against the plain entry the result differs by the
 * rounding of where the segments' fp32 sums are joined (parity tests hold both against the oracle at the same tolerance). */
size_t msnet_deconv5_softargmin_workspace_bytes(int N, int D, int H, int W);
int msnet_deconv5_softargmin_ws(const float* x, const float* w, float bias_host, float wscale, float* disp, int N, int D,
                                int H, int W, int Ci, void* workspace, size_t workspace_bytes, msnet_stream_t stream);
/* Un-fused deconv5 (stride 2 or the is_quarter_input_size stride-4/op-3 variant, gcnet_3dcnn.py:88-92).
 * logits: f32[N][s*D][s*H][s*W]. */
int msnet_deconv3d_cout1(const float* x, const float* w, float bias_host, float* logits, int N, int D,
                         int H, int W, int Ci, int stride, msnet_stream_t stream);
/* F.interpolate(trilinear, align_corners=True) to [D][H][W] fused with softmax + regression.
 * psmnet_3dcnn.py:167-174.  cost f32[N][d][h][w] -> disp f32[N][H][W]. */
int msnet_trilinear_softargmin(const float* cost, float* disp, int N, int d, int h, int w, int D, int H,
                               int W, msnet_stream_t stream);

/* ---- driver-side metric ------------------------------------------------------------------------ */
/* get_epe_rate(disp, prediction, max_disp, threshold), main_msnet.py:708-713, on the device: over the pixels with
 * 0.001 <= gt <= max_disp, out3 = {sum |pred - gt|, count(|pred - gt| > threshold), count(valid)} as doubles (the caller
 * divides: epe = out3[0] / out3[2], rate = out3[1] / out3[2]).  gt, pred: f32[n]; out3: device double[3]. */
int msnet_epe_badx(const float* gt, const float* pred, size_t n, float max_disp, float threshold, double* out3,
                   msnet_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MSNET_HIP_H */

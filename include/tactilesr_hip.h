/* tactilesr_hip.h -- C ABI of libtactilesr_hip.so (MI355X / gfx950 only).
 *
 * The reference (wmtlab/tactileSR) has no FFI of its own: its hot path is the set of
 * torch.nn / torch.nn.functional calls issued by model/tactileSR_model.py and
 * model/tPSFNet.py.  Each entry point below replaces one group of those calls; the
 * reference line(s) it stands in for are cited per function.  INTEGRATION.md shows the
 * ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (PyTorch): the library never
 *    allocates, frees or retains memory; `stream` is a hipStream_t (0 = null stream);
 *    all work is enqueued asynchronously on it;
 *  - return value: 0 = TSR_OK, 1 = bad argument, 2 = launch failure; no exceptions;
 *  - "CB16" = the internal channel-blocked activation layout float[B][C/16][H*W][16];
 *    `ctot` is the channel count of the whole buffer, `coff` the first channel touched
 *    (both multiples of 16) -- producers write straight into channel slices of the
 *    consumer's buffer, which is how every torch.cat of the reference is elided
 *    (model/tactileSR_model.py:74,81,200,203);
 *  - boundary tensors are NCHW fp32 contiguous, as the reference's callers pass them
 *    (train/tactileSR_train.py:43-47).
 */
#ifndef TACTILESR_HIP_H
#define TACTILESR_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

#define TSR_OK 0
#define TSR_ERR_ARG 1
#define TSR_ERR_LAUNCH 2

/* Library / ABI version (bumped on any signature change). */
int tsr_abi_version(void);

/* Re-order an nn.Conv2d weight (OIHW fp32; Cout in {64,128}, Cin % 16 == 0, k in {1,3,5})
 * into the [Cin/16][k*k][4][Cout][4] stream order tsr_conv2d_fwd consumes.
 * w_packed holds Cout*Cin*k*k floats. */
int tsr_pack_conv_weight(const float* w_oihw, float* w_packed, int cout, int cin, int ks, void* stream);

/* out[:, coff:coff+cout] = act( conv2d(in[:, in_coff:in_coff+cin], W, stride 1, pad k/2) * scale + shift
 *                               (+ res[:, res_coff:res_coff+cout]) ),  act = ReLU if relu else identity.
 * Replaces nn.Conv2d (+bias) + nn.BatchNorm2d(eval) + nn.ReLU / residual add of
 *   MSRB   model/tactileSR_model.py:167-191,196-206   (conv_3_1, conv_5_1, conv_3_2, conv_5_2, confusion,
 *                                                      `output += x`, relu)
 *   ResBlock                     :219-225
 *   stem conv2 / inputContact_layer / output_layer[0]  :41-43,47-49,53-54
 * scale/shift are per-output-channel (NULL = 1 / 0): eval-mode BN folds to
 *   scale = gamma/sqrt(running_var+eps), shift = (bias-running_mean)*scale+beta.
 * in/out/res are CB16; fp32 MFMA (exact fp32 fma chain). */
int tsr_conv2d_fwd(const float* in, int in_ctot, int in_coff, int cin,
                   const float* w_packed, int cout, int ks,
                   const float* scale, const float* shift,
                   const float* res, int res_ctot, int res_coff,
                   float* out, int out_ctot, int out_coff, int relu,
                   int B, int H, int W, void* stream);

/* nn.Upsample(scale_factor=sf, bilinear, align_corners=False) + Conv2d(3->64, 3x3, pad 1, no bias)
 * + scale/shift + optional ReLU: the pattern stem's first conv (model/tactileSR_model.py:34-39)
 * and the force stem (:59-63).  lr is the NCHW taxel tensor (B, lr_ctot, hin, win); channels
 * [lr_coff, lr_coff+3) are read (the x[:, 3t:3t+3] slices of :71-78).  Output CB16, H = hin*sf. */
int tsr_stem_fwd(const float* lr, int lr_ctot, int lr_coff, int axis_cnt, int hin, int win, int sf,
                 const float* w_oihw, const float* scale, const float* shift,
                 float* out, int out_ctot, int out_coff, int relu, int B, void* stream);

/* Conv2d(cin->1, 3x3, pad 1, no bias) + ReLU, CB16 in, NCHW (B,1,H,W) out: output_layer[2:]
 * (model/tactileSR_model.py:55-56).  The trailing same-size F.interpolate (:83) is an identity. */
int tsr_head_fwd(const float* in, int in_ctot, int cin, const float* w_oihw, float* out_nchw,
                 int relu, int B, int H, int W, void* stream);

/* Layout plumbing (tests, stage probes): NCHW (B,C,HW) <-> a channel slice of a CB16 buffer. */
int tsr_nchw_to_cb16(const float* src, float* dst, int B, int C, int HW, int dst_ctot, int dst_coff, void* stream);
int tsr_cb16_to_nchw(const float* src, float* dst, int B, int C, int HW, int src_ctot, int src_coff, void* stream);

#ifdef __cplusplus
}
#endif
#endif

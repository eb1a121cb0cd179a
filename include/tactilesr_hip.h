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

/* 0 for the shipped library; non-zero for an experimental variant built by tools/build_variant.py (extra -D flags).  The
 * Python binding refuses a non-zero library unless TSR_ALLOW_VARIANT=1 is set, and bench.py reports the value. */
int tsr_build_flags(void);

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

/* Split-bf16 variant of tsr_conv2d_fwd on the bf16 matrix cores (16x the fp32-MFMA rate on gfx950):
 * fp32 operands are split into nsplit bf16 planes and the significant cross products are accumulated in
 * fp32.  nsplit = 3 ("bf16x6": 6 products, 24 significand bits -> fp32-equivalent results, error <= the fp32
 * MFMA path's); nsplit = 2 ("bf16x3": 3 products, ~4e-6 per layer); nsplit = 1 (plain bf16 operands).
 * Same arguments and semantics as tsr_conv2d_fwd; weights come from tsr_pack_conv_weight_bf16s
 * (tsr_conv_weight_bf16s_elems() bf16 values: taps padded to the kernel's step size). */
long long tsr_conv_weight_bf16s_elems(int cout, int cin, int ks, int nsplit);   /* bf16 elements of w_packed */
int tsr_pack_conv_weight_bf16s(const float* w_oihw, void* w_packed, int cout, int cin, int ks, int nsplit,
                               void* stream);
int tsr_conv2d_fwd_bf16s(const float* in, int in_ctot, int in_coff, int cin,
                         const void* w_packed, int cout, int ks, int nsplit,
                         const float* scale, const float* shift,
                         const float* res, int res_ctot, int res_coff,
                         float* out, int out_ctot, int out_coff, int relu,
                         int B, int H, int W, void* stream);

/* The two stage-1 convolutions of an MSRB (3x3 64->64 and 5x5 64->64 on the same input,
 * model/tactileSR_model.py:167-175,198-200) as ONE launch on one staged halo (inference, fp16x3 arithmetic): a 5x5 conv
 * to 128 channels whose 3x3 half skips the 16 outer taps.  Output channels, scale and shift are in the kernel's channel
 * order: tsr_pair_channel_perm fills perm[k] = channel of torch.cat([conv3, conv5], 1) that kernel channel k holds; the
 * consumers' weights are permuted along C_in accordingly.  w_packed: tsr_conv_weight_pair_elems(cin) fp16 elements. */
int tsr_pair_channel_perm(int* perm128);      /* host array of 128 ints */
long long tsr_conv_weight_pair_elems(int cin);
int tsr_pack_conv_weight_pair_f16s(const float* w3_oihw, const float* w5_oihw, void* w_packed, int cin, float wscale,
                                   const float* w_amax, void* stream);
int tsr_conv2d_fwd_f16s_pair(const float* in, int in_ctot, int in_coff, int cin, const void* w_packed, float w_inv_scale,
                             const float* in_amax, float* out_amax, const float* scale, const float* shift,
                             float* out, int out_ctot, int out_coff, int relu, int B, int H, int W, void* stream);

/* fp16 two-plane variant ("fp16x3"): operands are scaled by powers of two into fp16's range -- weights at pack
 * time (wscale chosen by the caller: max|w|*wscale in [2^13,2^14); pass w_inv_scale = 1/wscale), activations in
 * the kernel from the device scalar in_amax = max|x| that the producer wrote through its out_amax -- then split
 * x*sx = h1+h2, w*sw = g1+g2 (22+ significand bits) and h1g1 + h1g2 + h2g1 are accumulated in fp32; the scales
 * are undone exactly in the epilogue.  Half the MFMA work of bf16x6; network-level error equal to the reference's
 * own fp32 CPU run against fp64 (tests).  Same remaining arguments as tsr_conv2d_fwd. */
int tsr_pack_conv_weight_f16s(const float* w_oihw, void* w_packed, int cout, int cin, int ks, float wscale,
                              void* stream);
/* Same with the scale derived ON THE DEVICE from w_amax[0] = max|w| (a device scalar the caller computed without a host
 * round trip): wscale = 2^(13 - floor(log2 max|w|)).  The matching convolution takes the same scalar through
 * tsr_conv_desc.w_amax. */
int tsr_pack_conv_weight_f16s_dev(const float* w_oihw, void* w_packed, int cout, int cin, int ks, const float* w_amax,
                                  void* stream);
int tsr_conv2d_fwd_f16s(const float* in, int in_ctot, int in_coff, int cin,
                        const void* w_packed, int cout, int ks, float w_inv_scale,
                        const float* in_amax, float* out_amax,
                        const float* scale, const float* shift,
                        const float* res, int res_ctot, int res_coff,
                        float* out, int out_ctot, int out_coff, int relu,
                        int B, int H, int W, void* stream);
/* MSRB stage-2 convolution with its half of the 1x1 `confusion` conv fused (model/tactileSR_model.py:196-206; no
 * `cat2` tensor in HBM).  W_c = [W_a | W_b]:  P = W_a.relu(bn(conv3(x1))) + b_c + x  (3x3 launch: w2 = W_a, shift2 = b_c,
 * res = x, relu2 = 0), out = relu(W_b.relu(bn(conv5(x1))) + P)  (5x5 launch: w2 = W_b, res = P, relu2 = 1).
 * fp16x3 arithmetic; the conv has C_out = 128; w2_packed = tsr_pack_conv_weight_f16s of the 64x128x1x1 half;
 * `out` / `res` / `out_amax` describe the 64-channel result. */
int tsr_conv2d_fwd_f16s_fuse1x1(const float* in, int in_ctot, int in_coff, int cin,
                                const void* w_packed, int ks, float w_inv_scale,
                                const float* in_amax, float* out_amax,
                                const float* scale, const float* shift, int relu,
                                const void* w2_packed, float w2_inv_scale, const float* shift2,
                                const float* res, int res_ctot, int res_coff,
                                float* out, int out_ctot, int out_coff, int relu2,
                                int B, int H, int W, void* stream);
/* bf16 ACTIVATION STORAGE (BASELINE's "bf16" configurations: reduced precision, tolerance 2e-2, never the parity path).
 * Activations are bf16 CB16 tensors -- `in` / `res` / `out` address bf16 elements, same [B][C/16][H*W][16] order --
 * weights the one-plane pack of tsr_pack_conv_weight_bf16s(nsplit = 1), fp32 accumulation and epilogue arithmetic
 * (same y = relu(conv*scale + shift + res) as tsr_conv2d_fwd).  tsr_stem_fwd_b16 writes such a tensor, tsr_head_fwd_b16
 * reads one (both otherwise identical to tsr_stem_fwd / tsr_head_fwd). */
int tsr_conv2d_fwd_b16(const void* in, int in_ctot, int in_coff, int cin, const void* w_packed, int cout, int ks,
                       const float* scale, const float* shift, const void* res, int res_ctot, int res_coff,
                       void* out, int out_ctot, int out_coff, int relu, int B, int H, int W, void* stream);
/* The bf16-storage INFERENCE convolutions proper (csrc/conv_b16k.hip; 3x3 / 5x5, C_out 64 / 128, C_in a multiple of 32):
 * v_mfma_f32_16x16x32_bf16 with channels as rows -- the accumulators are in CB16 order (8-B bf16 stores, no transpose)
 * and, for the fused form, already the operand layout of the 1x1 product -- LDS-DMA halo rows into a circular row buffer
 * and an LDS-DMA weight ring.  tsr_conv2d_fwd_b16k: same arguments and semantics as tsr_conv2d_fwd_b16.
 * tsr_conv2d_fwd_b16k_fuse1x1: a stage-2 convolution of an MSRB (128 -> 128, folded BatchNorm, ReLU) with its half of the
 * 1x1 `confusion` applied to the tile as it sits in the accumulator registers (see tsr_conv2d_fwd_f16s_fuse1x1; res / out
 * are 64-channel bf16 tensors).  tsr_conv2d_fwd_b16k_pair: the stage-1 pair of an MSRB, conv_3_1 || conv_5_1 (each conv +
 * BN + ReLU, model/tactileSR_model.py:167-175) and the first torch.cat (:200) as ONE launch on one staged halo: w_packed =
 * tsr_pack_conv_weight_b16k_pair(W = cat([3x3 weight zero-padded to 5x5, 5x5 weight]) along C_out, [128][cin][5][5];
 * tsr_conv_weight_b16k_pair_elems(cin) elements: one slab per barrier step, two outer taps per step), scale / shift = the two
 * convs' folded BatchNorm vectors concatenated (128), out = 128 channels in torch.cat order; the 3x3 half's MFMAs, fragment
 * reads and weight bytes on the 16 outer taps are skipped.  The weight layouts:
 * w_packed from tsr_pack_conv_weight_b16k ([C_in/32][tap][4][C_out][8] bf16; tsr_conv_weight_b16k_elems elements),
 * w2_packed from tsr_pack_w2_b16k (the 64x128 fp32 half of `confusion`, model/tactileSR_model.py:203-206, HALVED, in the K
 * order of the fused epilogue). */
long long tsr_conv_weight_b16k_elems(int cout, int cin, int ks);
int tsr_pack_conv_weight_b16k(const float* w_oihw, void* w_packed, int cout, int cin, int ks, void* stream);
int tsr_pack_w2_b16k(const float* w2_64x128, void* w_packed, void* stream);
/* Training with bf16 activation storage: a dgrad launch of tsr_conv2d_ex whose shape tsr_conv2d_ex_dgrad_b16k(nprime, cout,
 * ks) accepts may be described with nsplit = -3 instead of -1: it then runs on the same kernel (epi_mode 2, or 0 for the
 * unmasked partial gradient; no input / residual transform), with its weight packed by tsr_pack_conv_weight_dgrad_b16k
 * (arguments as tsr_pack_conv_weight_dgrad_bf16s, nprime = 128 or 64; tsr_conv_weight_b16k_elems(nprime, cout, ks) elements).
 * Slab entries as for nsplit = -1 (tsr_conv2d_slab_entries_ex accepts -3 / -4).  The predicate also accepts (128, 64, 1): the
 * masked dgrad of a 1x1 conv with 64 output channels (epi_mode 2, no partial gradient) as a streaming kernel without LDS
 * (csrc/conv1x1_b16k.hip; one slab entry per workgroup: ask tsr_conv2d_slab_entries_ex with nsplit = -3).  The same nsplit = -3 runs the FORWARD launches of that shape class (C_out = 128, C_in a
 * multiple of 32, 3x3 / 5x5, plain input) there too: epi_mode 1 (raw output + Welford partials; weights from
 * tsr_pack_conv_weight_b16k) and epi_mode 0 with that pack; nsplit = -4 is epi_mode 1 of the stage-1 pair of an MSRB
 * (conv_3_1 || conv_5_1 as one 5x5 launch with 128 output channels, weights from tsr_pack_conv_weight_b16k_pair). */
int tsr_conv2d_ex_dgrad_b16k(int nprime, int cout, int ks);
/* 1 if tsr_conv2d_ex accepts nsplit = -3 for the FORWARD of a 1x1 conv of this shape whose input is VIRTUAL (in_scale /
 * in_shift set: relu(z * scale + shift) of the stored pre-BatchNorm tensor, formed in LDS behind the DMA): epi_mode 0 (shift =
 * bias, residual, ReLU), weights from tsr_pack_conv_weight_b16k(.., ks = 1).  csrc/conv1x1_b16k.hip. */
int tsr_conv2d_ex_fwd1x1_b16k(int cout, int cin);
int tsr_pack_conv_weight_dgrad_b16k(const float* w_oihw, void* w_packed, int cout, int cin, int ks, int ci0, int nprime,
                                    void* stream);
long long tsr_conv_weight_b16k_pair_elems(int cin);
int tsr_pack_conv_weight_b16k_pair(const float* w128_oihw5, void* w_packed, int cin, void* stream);
int tsr_conv2d_fwd_b16k(const void* in, int in_ctot, int in_coff, int cin, const void* w_packed, int cout, int ks,
                        const float* scale, const float* shift, const void* res, int res_ctot, int res_coff,
                        void* out, int out_ctot, int out_coff, int relu, int B, int H, int W, void* stream);
int tsr_conv2d_fwd_b16k_fuse1x1(const void* in, int in_ctot, int in_coff, int cin, const void* w_packed, int ks,
                                const float* scale, const float* shift, int relu,
                                const void* w2_packed, const float* shift2,
                                const void* res, int res_ctot, int res_coff,
                                void* out, int out_ctot, int out_coff, int relu2,
                                int B, int H, int W, void* stream);
int tsr_conv2d_fwd_b16k_pair(const void* in, int in_ctot, int in_coff, int cin, const void* w_packed,
                             const float* scale, const float* shift, void* out, int out_ctot, int out_coff,
                             int relu, int B, int H, int W, void* stream);
int tsr_stem_fwd_b16(const float* lr, int lr_ctot, int lr_coff, int axis_cnt, int hin, int win, int sf,
                     const float* w_oihw, const float* scale, const float* shift,
                     void* out_bf16, int out_ctot, int out_coff, int relu, int B, void* stream);
int tsr_head_fwd_b16(const void* in_bf16, int in_ctot, int cin, const float* w_oihw, float* out_nchw,
                     int relu, int B, int H, int W, void* stream);

/* nn.Upsample(scale_factor=sf, bilinear, align_corners=False) + Conv2d(3->64, 3x3, pad 1, no bias)
 * + scale/shift + optional ReLU: the pattern stem's first conv (model/tactileSR_model.py:34-39)
 * and the force stem (:59-63).  lr is the NCHW taxel tensor (B, lr_ctot, hin, win); channels
 * [lr_coff, lr_coff+3) are read (the x[:, 3t:3t+3] slices of :71-78).  Output CB16, H = hin*sf. */
int tsr_stem_fwd(const float* lr, int lr_ctot, int lr_coff, int axis_cnt, int hin, int win, int sf,
                 const float* w_oihw, const float* scale, const float* shift,
                 float* out, int out_ctot, int out_coff, int relu, int B, float* out_amax, void* stream);
/* out_amax (optional device scalar): receives max|output| by atomic max -- the fp16-split convolution that
 * consumes the tensor derives its power-of-two input scale from it. */

/* Conv2d(cin->1, 3x3, pad 1, no bias) + ReLU, CB16 in, NCHW (B,1,H,W) out: output_layer[2:]
 * (model/tactileSR_model.py:55-56).  The trailing same-size F.interpolate (:83) is an identity. */
int tsr_head_fwd(const float* in, int in_ctot, int cin, const float* w_oihw, float* out_nchw,
                 int relu, int B, int H, int W, void* stream);

/* ---------------------------------------------------------------------------------------
 * Training path (train/tactileSR_train.py:41-51 -> cpu/trainer.py:346-362): train-mode
 * BatchNorm, convolution_backward (dgrad = tsr_conv2d_ex with tsr_pack_conv_weight_dgrad
 * weights, wgrad = tsr_conv2d_wgrad), MSE, Adam.
 * ------------------------------------------------------------------------------------- */

/* Extended convolution launch descriptor (plain C, device pointers; NULL/0 = unused).
 *  - in_scale/in_shift [cin]: x' = relu(x*s+t) applied to in-bounds input pixels while staging
 *    (the producer stored its raw bias-free conv output; its train-mode BN+ReLU happens here);
 *    res_scale/res_shift [cout]: same on the residual operand;
 *  - epi_mode 0: out = act(acc*scale+shift (+res));
 *    epi_mode 1: out = acc (raw); slab[(e*cout+c)*2+{0,1}] = (mean, M2) of the 64 pixels of entry
 *                e = (workgroup, image), slab_cnt[e] = valid pixel count  (BatchNorm batch statistics;
 *                entries = tsr_conv2d_slab_entries(B,H,W));
 *    epi_mode 2: out = (acc*scale (+res)) * [mask*mask_scale+mask_shift > 0]  (ReLU backward by the stored
 *                forward tensor `mask`); if bn_a: slab[(e*cout+c)*2+{0,1}] = (sum out, sum out*xhat),
 *                xhat = mask*bn_a+bn_b  (the two reductions of BatchNorm backward). */
typedef struct tsr_conv_desc {
  const float* in; int in_ctot; int in_coff; int cin;
  const float* w_packed; int cout; int ks;
  const float* scale; const float* shift;
  const float* res; int res_ctot; int res_coff;
  float* out; int out_ctot; int out_coff; int relu;
  int B; int H; int W;
  const float* in_scale; const float* in_shift;
  const float* res_scale; const float* res_shift;
  int epi_mode;
  const float* mask; int mask_ctot; int mask_coff;
  const float* mask_scale; const float* mask_shift;
  const float* bn_a; const float* bn_b;
  float* slab; float* slab_cnt;
  int nsplit;   /* -3 / -4: as -1 on csrc/conv_b16k.hip (see tsr_conv2d_ex_dgrad_b16k); -1: plain bf16 operands AND bf16 CB16 tensors (in / res / mask / out address bf16 elements; w_packed as
                   for 1): the train step with bf16 activation storage;
                   0: fp32 MFMA, w_packed from tsr_pack_conv_weight[_dgrad]; 1..3: split-bf16 MFMA (3 = fp32-equivalent),
                   w_packed from tsr_pack_conv_weight[_dgrad]_bf16s; -2: fp16 two-plane split ("fp16x3"), w_packed from
                   tsr_pack_conv_weight[_dgrad]_f16s, needs in_amax and w_inv_scale below */
  const float* in_amax;   /* device scalar max|in| (of the raw tensor; a fused input transform is bounded in-kernel) */
  float w_inv_scale;      /* 1 / wscale used at pack time */
  float* out_amax;        /* device scalar receiving max|out| (atomic max), any mode; NULL = off */
  const float* w_amax;    /* nsplit == -2, optional: device scalar max|w| of a weight packed by ..._f16s_dev; replaces
                             w_inv_scale (no host round trip for the weight scale) */
} tsr_conv_desc;

int tsr_conv2d_ex(const tsr_conv_desc* desc, void* stream);
int tsr_conv2d_slab_entries(int B, int H, int W);
/* Entries the launch described by (cout, ks, nsplit) writes -- the count to hand to tsr_bn_stats_finalize /
 * tsr_bn_bwd_finalize (the fp16-split 3x3/5x5 kernels with 64 output channels use 4 images per workgroup, all other
 * forms 2: tsr_conv2d_slab_entries is the 2-image count).  Size slabs for the larger of the two. */
int tsr_conv2d_slab_entries_ex(int B, int H, int W, int cout, int ks, int nsplit);

/* Weights of the data-gradient convolution: W'[n][co][kh][kw] = W[co][ci0+n][K-1-kh][K-1-kw],
 * n in [0,nprime), nprime in {64,128}; packed for tsr_conv2d_* with cin := cout, cout := nprime. */
int tsr_pack_conv_weight_dgrad(const float* w_oihw, float* w_packed, int cout, int cin, int ks,
                               int ci0, int nprime, void* stream);

int tsr_pack_conv_weight_dgrad_bf16s(const float* w_oihw, void* w_packed, int cout, int cin, int ks,
                                     int ci0, int nprime, int nsplit, void* stream);

int tsr_pack_conv_weight_dgrad_f16s(const float* w_oihw, void* w_packed, int cout, int cin, int ks,
                                    int ci0, int nprime, float wscale, void* stream);
int tsr_pack_conv_weight_dgrad_f16s_dev(const float* w_oihw, void* w_packed, int cout, int cin, int ks,
                                        int ci0, int nprime, const float* w_amax, void* stream);

/* Weight (and bias) gradient partials: slab[s][cout][cin][k][k] (s < nsplit, OIHW) with
 * dW = sum_s slab[s] (tsr_reduce_splits), from a = conv input (CB16, optional relu(a*scale+shift)
 * transform) and dz = gradient w.r.t. the conv output (CB16).  cin, cout multiples of 64.
 * bias_slab[s][cout] (optional) receives sum over pixels of dz. */
int tsr_conv2d_wgrad(const float* a, int a_ctot, int a_coff, int cin,
                     const float* a_scale, const float* a_shift,
                     const float* dz, int dz_ctot, int dz_coff, int cout, int ks,
                     float* slab, float* bias_slab, int nsplit, int B, int H, int W, void* stream);
/* Same on the 16-bit matrix cores with split operands: planes = 3 -> three bf16 planes, six products
 * (fp32-equivalent; a_amax/dz_amax unused); planes = 1 -> plain bf16 operands (reduced precision: the "bf16"
 * configurations, never the parity path); planes = -2 -> two power-of-two-scaled fp16 planes, three products,
 * scales derived from the device scalars a_amax = max|a| (raw tensor) and dz_amax = max|dz|;
 * planes = -1 -> plain bf16 operands read from bf16 CB16 TENSORS (`a`, `dz` then address bf16 elements): the train step
 * with bf16 activation storage (BASELINE configs[2] / [4], "bf16"; reference switch: cpu/trainer.py:96,203,346-362). */
int tsr_conv2d_wgrad_bf16s(const float* a, int a_ctot, int a_coff, int cin,
                           const float* a_scale, const float* a_shift,
                           const float* dz, int dz_ctot, int dz_coff, int cout, int ks, int planes,
                           const float* a_amax, const float* dz_amax,
                           float* slab, float* bias_slab, int nsplit, int B, int H, int W, void* stream);
/* Batch splits to launch tsr_conv2d_wgrad_bf16s with (slab / bias_slab hold that many partials): one resident round
 * of workgroups for this layer shape, never more than there are (image, 4x8 patch) work items; and the number of
 * workgroups one split launches. */
int tsr_conv2d_wgrad_splits(int cout, int cin, int ks, int planes, int B, int H, int W);
int tsr_conv2d_wgrad_wgs_per_split(int cout, int cin, int ks, int planes);
int tsr_reduce_splits(const float* slab, float* out, long long n, int nsplit, float alpha, void* stream);

/* nn.BatchNorm2d train mode (model/tactileSR_model.py:38,42,48,169,175,181,187), from the
 * epi_mode-1 slabs: batch mean / biased variance -> scale = gamma*invstd, shift = beta-mean*scale
 * (to apply on the bias-free conv output), xhat_a = invstd, xhat_b = -mean*invstd; running_mean/var
 * updated in place (momentum, unbiased variance, +bias on the mean).  work: 512*C*3 doubles. */
int tsr_bn_stats_finalize(const float* slab, const float* slab_cnt, int entries, int C,
                          const float* bias, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, float momentum, float eps,
                          float* scale, float* shift, float* xhat_a, float* xhat_b,
                          double* work, void* stream);
/* Same statistics slabs for a 64-channel CB16 slice produced by a kernel without a stats epilogue
 * (the VALU stem): entries = tsr_cb16_stats_entries(B, HW). */
int tsr_cb16_stats_entries(int B, int HW);
int tsr_cb16_stats(const float* z, int z_ctot, int z_coff, int B, int HW, float* slab, float* slab_cnt,
                   void* stream);
/* BatchNorm backward reductions from the epi_mode-2 slabs: dgamma, dbeta and the coefficients of
 * dz = c1*g + c2*z + c3 (N = B*H*W). */
int tsr_bn_bwd_finalize(const float* slab, int entries, int C, double N, const float* scale,
                        const float* xhat_a, const float* xhat_b, float* dgamma, float* dbeta,
                        float* c1, float* c2, float* c3, double* work, void* stream);
int tsr_bn_bwd_apply(float* g, int g_ctot, int g_coff, const float* z, int z_ctot, int z_coff,
                     const float* c1, const float* c2, const float* c3, int C, int B, int HW,
                     float* out_amax /* optional: max|g| after the update */, void* stream);

/* Backward of tsr_stem_fwd's conv weight: slab[s][64][3][3][3] partials (taxels carry no gradient). */
int tsr_stem_wgrad(const float* lr, int lr_ctot, int lr_coff, int hin, int win, int sf,
                   const float* dz, int dz_ctot, int dz_coff, float* slab, int nsplit, int B, void* stream);
/* Backward of tsr_head_fwd: dz_h0 = dgrad(dout*[out>0]) * [h0>0] (CB16) and weight partials
 * wslab[s][cin][3][3]. */
int tsr_head_bwd(const float* dout, const float* out, const float* h0, int h_ctot, int cin,
                 const float* w_oihw, float* dz_h0, int dz_ctot, float* wslab, int nsplit,
                 int B, int H, int W, float* dz_amax /* optional: max|dz_h0| */, void* stream);

/* The same four kernels on bf16 CB16 tensors -- the train step with bf16 ACTIVATION STORAGE (tsr_conv_desc.nsplit = -1,
 * tsr_conv2d_wgrad_bf16s planes = -1): every stored activation / gradient tensor is bf16, arithmetic and the
 * statistics / weight-gradient slabs stay fp32.  Pointers typed void* address bf16 elements. */
int tsr_cb16_stats_b16(const void* z, int z_ctot, int z_coff, int B, int HW, float* slab, float* slab_cnt,
                       void* stream);
int tsr_bn_bwd_apply_b16(void* g, int g_ctot, int g_coff, const void* z, int z_ctot, int z_coff,
                         const float* c1, const float* c2, const float* c3, int C, int B, int HW, void* stream);
/* out[B][C/16][HW][16] (bf16) = bf16(relu(fp32(z) * scale + shift)) of channels z_coff .. z_coff + C of the stored
 * pre-BatchNorm tensor: the MATERIALISED form of a virtual activation.  tsr_conv2d_wgrad_bf16s (planes = -1) runs the 3x3 /
 * 5x5 launches for which tsr_conv2d_wgrad_b16k(cout, cin, ks) returns 1 on csrc/wgrad_b16k.hip (operands straight from HBM
 * into LDS) when they carry NO input transform: a caller with a virtual input materialises it once and passes it plain. */
int tsr_bn_relu_b16(const void* z, int z_ctot, int z_coff, int C, const float* scale, const float* shift, void* out,
                    int B, int HW, void* stream);
int tsr_conv2d_wgrad_b16k(int cout, int cin, int ks);
int tsr_stem_wgrad_b16(const float* lr, int lr_ctot, int lr_coff, int hin, int win, int sf,
                       const void* dz, int dz_ctot, int dz_coff, float* slab, int nsplit, int B, void* stream);
int tsr_head_bwd_b16(const float* dout, const float* out, const void* h0, int h_ctot, int cin,
                     const float* w_oihw, void* dz_h0, int dz_ctot, float* wslab, int nsplit,
                     int B, int H, int W, void* stream);

/* HR.float()/HR_scale_num + F.interpolate(size=(H,W), bilinear) (train/tactileSR_train.py:44-45). */
int tsr_target_prep(const float* hr_raw, float* out, float inv_scale, int B, int hin, int win, int H, int W,
                    void* stream);
/* nn.MSELoss() forward (loss[0]) and backward (dy = grad_scale*2(y-t)/n; dy may be NULL).  work: 256 doubles. */
int tsr_mse_fwd_bwd(const float* y, const float* target, float* dy, float* loss, long long n,
                    float grad_scale, double* work, void* stream);
/* torch.optim.Adam step with L2-in-gradient weight decay (train/tactileSR_train.py:212); step is 1-based. */
int tsr_adam_l2_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n,
                     float lr, float beta1, float beta2, float eps, float weight_decay, int step, void* stream);
/* The same step for MANY tensors in one launch (the reference's optim.Adam over all 124 parameter tensors,
 * train/tactileSR_train.py:212; cpu/trainer.py:361): `chunks` is a DEVICE array of n_chunks records, one per
 * <= 4096-element piece of a tensor, pointers already offset to the piece.  Same arithmetic as tsr_adam_l2_step. */
typedef struct tsr_adam_chunk {
  float* param; const float* grad; float* exp_avg; float* exp_avg_sq;
  int n; int reserved;
} tsr_adam_chunk;
int tsr_adam_l2_multi(const tsr_adam_chunk* chunks, int n_chunks, float lr, double beta1, double beta2, float eps,
                      float weight_decay, int step, void* stream);

/* Per-sample PSNR / SSIM of eval_func (train/tactileSR_train.py:87-94, utility/tools.py:49-81) for B samples
 * of n elements: PSNR = 10log10(max^2/(sum(a-b)^2/psnr_div)) with psnr_div = shape[0]*shape[1] of what the
 * reference passes ((1,H,W) -> H, its /40 quirk; (H,W) -> H*W); SSIM = one global window. */
int tsr_psnr_ssim(const float* a, const float* b, int B, int n, double psnr_div, double max_value,
                  double C1, double C2, float* psnr, float* ssim, void* stream);

/* ---------------------------------------------------------------------------------------
 * tPSFNet (model/tPSFNet.py): batched, one workgroup per sample (the reference loops over the
 * batch in python, :118-125).  size is fixed at 100x100 depth / 99x99 PSF / 4x4 taxels, as the
 * reference's geometry constants are (:40-55).
 * ------------------------------------------------------------------------------------- */
/* tactilePSF + depth2tactile + degradation_process (:78-100,129-141) from alpha_beta (B,3) =
 * (alpha, beta, gamma): HR (B,1,100,100), LR_deg (B,1,4,4), psf (B,1,99,99).  Separable form. */
int tpsf_forward(const float* depth, const float* alpha_beta, float* HR, float* LR_deg, float* psf,
                 int B, void* stream);
/* d loss / d (alpha, beta, gamma) (B,3) given d loss / d LR_deg (B,16); plateau pixels carry no
 * gradient, depth carries none (autograd of :118-125 as used by train/tPSFNet_train.py:180-190).  HR = the forward
 * output of the same (depth, alpha_beta) (tpsf_forward: the reductions over it are not recomputed); work: B*100*100
 * floats of scratch (the per-pixel dL/dHR between the two kernels). */
int tpsf_backward(const float* depth, const float* alpha_beta, const float* HR, const float* dLR_deg,
                  float* d_alpha_beta, float* work, int B, void* stream);
/* C[i][j] = act(sum_k A(i,k)B(k,j) + bias[j]), A(i,k)=A[i*sa0+k*sa1], B(k,j)=B[k*sb0+j*sb1]; act 0 none,
 * 1 ReLU, 2 Softplus: the nn.Linear layers of MLP_layer (:26-36) and their backward GEMMs. */
int tsr_sgemm(const float* A, long long sa0, long long sa1, const float* B, long long sb0, long long sb1,
              const float* bias, float* C, int M, int N, int K, int act, void* stream);
/* Split-K form for the reductions over the batch (dW = dy^T x, db = 1^T dy of the same layers): slab[s][M][N]
 * receives the partial product of K range s (no bias / activation); add the slabs with tsr_reduce_splits. */
int tsr_sgemm_splitk(const float* A, long long sa0, long long sa1, const float* B, long long sb0, long long sb1,
                     float* slab, int M, int N, int K, int nsplit, void* stream);
/* dx GEMM of a layer with the ReLU backward of the layer below fused: C[i][j] = (sum_k A(i,k)B(k,j)) if mask_ref[i][j] > 0
 * else 0 (mask_ref = the stored ReLU output the gradient flows back through, row-major [M][N]). */
int tsr_sgemm_masked(const float* A, long long sa0, long long sa1, const float* B, long long sb0, long long sb1,
                     const float* mask_ref, float* C, int M, int N, int K, void* stream);
/* tsr_sgemm_splitk with a caller-chosen distance between the partial results (split s lands at slab + s*split_stride,
 * split_stride >= M*N): the partials of several GEMMs share one [nsplit][total] buffer that ONE tsr_reduce_splits call
 * adds into a flat gradient buffer. */
int tsr_sgemm_splitk_strided(const float* A, long long sa0, long long sa1, const float* B, long long sb0, long long sb1,
                             float* slab, long long split_stride, int M, int N, int K, int nsplit, void* stream);
/* Column sums of Y[M][N] per row range (db = 1^T dy over the same K ranges as the split-K GEMM with K = M):
 * slab[s*split_stride + j] = sum of rows of range s. */
int tsr_colsum_splitk(const float* Y, float* slab, long long split_stride, int M, int N, int nsplit, void* stream);
/* dy *= act'(.) in place from the stored activation output (1 ReLU, 2 Softplus). */
int tsr_act_bwd(float* dy, const float* y, long long n, int mode, void* stream);

/* Layout plumbing (tests, stage probes): NCHW (B,C,HW) <-> a channel slice of a CB16 buffer. */
int tsr_nchw_to_cb16(const float* src, float* dst, int B, int C, int HW, int dst_ctot, int dst_coff, void* stream);
int tsr_cb16_to_nchw(const float* src, float* dst, int B, int C, int HW, int src_ctot, int src_coff, void* stream);

#ifdef __cplusplus
}
#endif
#endif

// Kernel-side argument block shared by the convolution kernels (fp32 MFMA and split-bf16 MFMA).
#pragma once
struct ConvArgs {
  const float* in;  int in_ctot;  int in_coff;  int cin;
  const float* wp;
  const float* scale; const float* shift;
  const float* res; int res_ctot; int res_coff;
  float* out; int out_ctot; int out_coff;
  int relu;
  int B, H, W;
  int tiles_x, tiles_y;
  // ---- training-path extensions (all optional; NULL/0 = inference behaviour) ----
  // input transform applied while staging in-bounds halo pixels: x' = relu(x*in_scale[c]+in_shift[c])
  // (the producer stored the raw, bias-free conv output; its train-mode BN+ReLU is applied here)
  const float* in_scale; const float* in_shift;
  // same transform on the residual operand
  const float* res_scale; const float* res_shift;
  // epi_mode 1: store the raw accumulator and emit per-(workgroup, image) Welford partials
  //             (mean, M2) per channel to slab[(wg*2+wm)*COUT*2 ...], counts to slab_cnt
  // epi_mode 2: v = (acc + res) * [mask*mask_scale+mask_shift > 0]  (ReLU backward by the stored
  //             activation); if bn_a: also emit sum(v), sum(v*xhat), xhat = mask*bn_a+bn_b
  int epi_mode;
  const float* mask; int mask_ctot; int mask_coff;
  const float* mask_scale; const float* mask_shift;
  const float* bn_a; const float* bn_b;
  float* slab; float* slab_cnt;
  // ---- fp16-split path (dynamic power-of-two operand scaling; all optional) ----
  const float* in_amax;   // device scalar: max |input activation| (written by the producer) -> input scale 2^k
  float w_inv_scale;      // 1 / (power-of-two weight scale applied at pack time); 0 = unused
  float* out_amax;        // device scalar: atomic max of |output| after the epilogue (for the consumer)
  // ---- fused second GEMM (conv_fuse1x1.h; inference, fp16x3, C_out = 128): out = act2(W2 . act(bn(conv)) + shift2 + res)
  const void* w2;         // 64 x 128 x 1 x 1 weight packed by tsr_pack_conv_weight_f16s; out / res then describe 64 channels
  float w2_inv_scale;     // 1 / (its pack-time power-of-two scale)
  const float* shift2;    // optional per-channel bias of the second GEMM
  int relu2;
  const float* w_amax;    // optional device scalar max|w|: the pack-time weight scale is re-derived from it (replaces w_inv_scale)
};

// Fused second GEMM of an MSRB stage-2 convolution (inference, fp16x3 arithmetic): the 1x1 `confusion` conv never sees a
// `cat2` tensor in HBM.  The reference block computes (model/tactileSR_model.py:196-206)
//     out = relu(W_c . cat(relu(bn(conv3(x1))), relu(bn(conv5(x1)))) + b_c + x)
// and a 1x1 convolution is linear in its input channels, so with W_c = [W_a | W_b]
//     P   =      W_a . relu(bn(conv3(x1))) + b_c + x          (epilogue of the 3x3 launch, no ReLU)
//     out = relu(W_b . relu(bn(conv5(x1))) + P)               (epilogue of the 5x5 launch)
// Each stage-2 workgroup holds its 128 px x 128 ch result in accumulators; this epilogue applies BN + ReLU there,
// splits the tile into two fp16 planes with a tile-local power-of-two scale, parks it in the (now idle) LDS of the main
// loop in A-fragment order, and runs a 128 px x 64 co x K = 128 GEMM on the matrix cores (three products, the W half
// streamed from L2 in B-fragment order: the pack of tsr_pack_conv_weight_f16s for a 64x128x1x1 weight) before the usual
// residual / ReLU / 16-B-store output stage.  Saves the 1x1 launches (HBM-bound, 12 ms of 167 ms at B = 4096) and half
// of the stage-2 store traffic for +2 % (5x5) / +5.6 % (3x3) MFMA work.
#pragma once
#include "tsr_common.h"
#include "conv_args.h"
#include "conv_epilogue.h"

typedef _Float16 fz_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 fz_f16x4 __attribute__((ext_vector_type(4)));

struct Fuse1x1Geom {
  static constexpr int PIXB = 64;                  // [plane 2][16 ch] fp16 per (block, pixel)
  static constexpr int ROWB = (8 * 4 + 1) * 16;    // 8 pixels + one 16-B slot: row stride == 1 (mod 16) slots, like the halo
  static constexpr int IMGB = 8 * ROWB;
  static constexpr int BLKB = 2 * IMGB;            // two images per workgroup
  static constexpr int BYTES = 8 * BLKB;           // 128 channels = 8 blocks: 67,584 B
};

// acc: the conv accumulators of this wave (image wm, channels wn*64 + nb*32 + li); lds: >= Fuse1x1Geom::BYTES + 64
__device__ __forceinline__ void conv_fuse1x1_epilogue(const ConvArgs& a, f32x16 (&acc)[2][2], char* lds, int b0, int y0,
                                                      int x0, int wm, int wn, int h, int li, int HW, float accmul) {
  typedef Fuse1x1Geom E;
  const int b = b0 + wm;
  const bool img_ok = b < a.B;
  // ---- 1. BatchNorm fold + ReLU in place (lane = channel), zero outside the image, tile maximum
  float amax = 0.f;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int n = wn * 64 + nb * 32 + li;
    const float sc = (a.scale ? a.scale[n] : 1.f) * accmul, sh = a.shift ? a.shift[n] : 0.f;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int gy = y0 + 4 * mb + (m >> 3), gx = x0 + (m & 7);
        float v = fmaf(acc[mb][nb][r], sc, sh);
        if (a.relu) v = tsr_relu(v);
        if (!(img_ok && gy < a.H && gx < a.W)) v = 0.f;
        acc[mb][nb][r] = v;
        amax = fmaxf(amax, fabsf(v));
      }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
  float* red = (float*)(lds + E::BYTES);
  // (the main loop ended on a barrier: no wave still reads the halo / weight ring)
  if ((h | li) == 0) red[wm * 2 + wn] = amax;
  __syncthreads();
  // (a wave whose image holds an Inf does not take part in the tile scale: the other image of the tile keeps its result)
  float tmax = 0.f;
#pragma unroll
  for (int w = 0; w < 4; ++w) tmax = fmaxf(tmax, red[w] < 3.0e38f ? red[w] : 0.f);
  float s_e = 1.f;
  if (tmax > 0.f) {
    int e = (int)((__float_as_uint(tmax) >> 23) & 0xFF) - 127;
    int be = 13 - e + 127;
    be = be < 1 ? 1 : (be > 254 ? 254 : be);
    s_e = __uint_as_float((unsigned)be << 23);
  }
  // ---- 2. (lane = channel, register = pixel) -> (lane = pixel, 4 channels), two scaled fp16 planes, into LDS
  const int k4 = li >> 2, j = li & 3;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int nq = wn * 64 + nb * 32 + 4 * k4;
    char* eb = lds + (nq >> 4) * E::BLKB + wm * E::IMGB + (j + 4 * h) * E::PIXB + (nq & 15) * 2;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float t0 = acc[mb][nb][4 * g + 0], t1 = acc[mb][nb][4 * g + 1], t2 = acc[mb][nb][4 * g + 2],
              t3 = acc[mb][nb][4 * g + 3];
        quad_transpose(t0, t1, t2, t3, j);
        float v[4] = {t0 * s_e, t1 * s_e, t2 * s_e, t3 * s_e};
        fz_f16x4 p0, p1;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          p0[c] = (_Float16)v[c];
          p1[c] = (_Float16)(v[c] - (float)p0[c]);
        }
        char* dst = eb + (4 * mb + g) * E::ROWB;
        *(fz_f16x4*)(dst) = p0;
        *(fz_f16x4*)(dst + 32) = p1;
      }
  }
  __syncthreads();
  // ---- 3. out[64 px of image wm][32 co of half wn] = sum over 128 channels, K = 16 per step
  f32x16 acc2[2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[mb][r] = 0.f;
  const char* ea = lds + wm * E::IMGB + (li >> 3) * E::ROWB + (li & 7) * E::PIXB + h * 16;
  const _Float16* wb = (const _Float16*)a.w2 + ((size_t)h * 64 + wn * 32 + li) * 8;     // [chunk][plane][2][64][8]
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    fz_f16x8 fa[2][2], fb[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) fa[p][mb] = *(const fz_f16x8*)(ea + c * E::BLKB + (4 * mb) * E::ROWB + p * 32);
      fb[p] = *(const fz_f16x8*)(wb + (size_t)(c * 2 + p) * (2 * 64 * 8));
    }
    // small terms first: h2.g1, h1.g2, h1.g1
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) acc2[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1][mb], fb[0], acc2[mb], 0, 0, 0);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) acc2[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][mb], fb[1], acc2[mb], 0, 0, 0);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) acc2[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][mb], fb[0], acc2[mb], 0, 0, 0);
  }
  // ---- 4. + bias + residual, ReLU, max|out|, 16-B stores (the mode-0 output stage for C_out = 64)
  const float mul2 = a.w2_inv_scale / s_e;
  const int out_blocks = a.out_ctot >> 4, res_blocks = a.res_ctot >> 4;
  const int bsafe = img_ok ? b : 0;
  const int nq2 = wn * 32 + 4 * k4;
  const int oq = a.out_coff + nq2;
  f32x4 sh4 = {0.f, 0.f, 0.f, 0.f};
  if (a.shift2) sh4 = *(const f32x4*)(a.shift2 + nq2);
  float* ob4 = a.out + (((size_t)bsafe * out_blocks + (oq >> 4)) * HW) * 16 + (oq & 15);
  const float* rb4 = nullptr;
  if (a.res) {
    const int rq = a.res_coff + nq2;
    rb4 = a.res + (((size_t)bsafe * res_blocks + (rq >> 4)) * HW) * 16 + (rq & 15);
  }
  float omax = 0.f;
  const int gx = x0 + j + 4 * h;
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float t0 = acc2[mb][4 * g + 0], t1 = acc2[mb][4 * g + 1], t2 = acc2[mb][4 * g + 2], t3 = acc2[mb][4 * g + 3];
      quad_transpose(t0, t1, t2, t3, j);
      const int gy = y0 + 4 * mb + g;
      if (img_ok && gy < a.H && gx < a.W) {
        const size_t po = (size_t)(gy * a.W + gx) * 16;
        f32x4 v = {t0, t1, t2, t3};
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = v[c] * mul2 + sh4[c];
        if (rb4) {
          const f32x4 rv = *(const f32x4*)(rb4 + po);
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] += rv[c];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (a.relu2) v[c] = tsr_relu(v[c]);
          omax = fmaxf(omax, fabsf(v[c]));
        }
        *(f32x4*)(ob4 + po) = v;
      }
    }
  if (a.out_amax) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) omax = fmaxf(omax, __shfl_xor(omax, o));
    if ((h | li) == 0) publish_amax(a.out_amax, omax);
  }
}

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

// ------------------------------------------------------------------------------------------------------------------
// The same fusion for the bf16 ACTIVATION-STORAGE path (4 images per workgroup, wave = image x all 128 channels): one
// bf16 plane, one product, bf16 residual / output tensors.  A wave needs only its own image's channels, so the
// (lane = channel) -> (lane = pixel) turn goes through a wave-private LDS region, 64 channels at a time (12 KB per wave).
typedef __bf16 fz_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 fz_bf16x4 __attribute__((ext_vector_type(4)));

struct Fuse1x1GeomB16 {
  static constexpr int PIXB = 48;                  // 16 ch bf16 + one 16-B slot (the NS = 1 halo's conflict-free stride)
  static constexpr int ROWB = 24 * 16;             // 8 pixels x 3 slots: row stride == 8 (mod 16) slots
  static constexpr int IMGB = 8 * ROWB;            // one 16-channel block of one image: 3,072 B
  static constexpr int WAVEB = 4 * IMGB;           // 64 channels
  static constexpr int BYTES = 4 * WAVEB;          // 4 waves: 49,152 B
};

__device__ __forceinline__ void conv_fuse1x1_b16_epilogue(const ConvArgs& a, f32x16 (&acc)[2][4], char* lds, int b0,
                                                          int y0, int x0, int wm, int h, int li, int HW) {
  typedef Fuse1x1GeomB16 E;
  const int b = b0 + wm;
  const bool img_ok = b < a.B;
  const int k4 = li >> 2, j = li & 3;
  char* ew = lds + wm * E::WAVEB;
  f32x16 acc2[2][2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[mb][n2][r] = 0.f;
  const char* ea = ew + (li >> 3) * E::ROWB + (li & 7) * E::PIXB + h * 16;
  const __bf16* wb = (const __bf16*)a.w2 + ((size_t)h * 64 + li) * 8;          // [chunk][2][64][8]
  // the residual tile of the output stage, requested up front and branch-free (inside the output stage's bounds branch
  // each of the 16 loads was waited for on its own)
  const int out_blocks = a.out_ctot >> 4, res_blocks = a.res_ctot >> 4;
  const int bsafe = img_ok ? b : 0;
  const int gx = x0 + j + 4 * h;
  bool okk[8];
  unsigned pov[8];
  fz_bf16x4 rvv[2][8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int gy = y0 + t;                     // t = 4 mb + g
    okk[t] = img_ok && gy < a.H && gx < a.W;
    pov[t] = okk[t] ? (unsigned)(gy * a.W + gx) * 16u : 0u;
  }
#pragma unroll
  for (int n2 = 0; n2 < 2; ++n2) {
    const int rq = a.res_coff + n2 * 32 + 4 * k4;
    const __bf16* rb4 = a.res ? (const __bf16*)a.res + (((size_t)bsafe * res_blocks + (rq >> 4)) * HW) * 16 + (rq & 15) : nullptr;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (rb4) rvv[n2][t] = *(const fz_bf16x4*)(rb4 + pov[t]);
      else rvv[n2][t] = (fz_bf16x4){(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
    }
  }
  // Everything the epilogue reads from global memory is requested HERE, in one round trip: the BatchNorm vectors of all four
  // 32-channel blocks, the output stage's shift, and the W2 fragments of the first half (the second half's are requested
  // as soon as the first half's accumulators are parked and their registers free).  Loaded where they were used -- a
  // scale / shift pair per block, a W2 fragment pair per K step right in front of its MFMAs -- each was a full L2 round
  // trip in a workgroup that has no MFMA work to cover it: s_memtime stamps put this epilogue at 26-28 k cycles, 40 % of the
  // 3x3 launch's workgroup time (fp16x3 form, which prefetches: 18 k).
  float bsc[4], bsh[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    bsc[nb] = a.scale ? a.scale[nb * 32 + li] : 1.f;
    bsh[nb] = a.shift ? a.shift[nb * 32 + li] : 0.f;
  }
  f32x4 osh[2];
#pragma unroll
  for (int n2 = 0; n2 < 2; ++n2)
    osh[n2] = a.shift2 ? *(const f32x4*)(a.shift2 + n2 * 32 + 4 * k4) : (f32x4){0.f, 0.f, 0.f, 0.f};
  fz_bf16x8 fbw[4][2];
  auto load_w2 = [&](int hb) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int n2 = 0; n2 < 2; ++n2) fbw[c][n2] = *(const fz_bf16x8*)(wb + (size_t)(4 * hb + c) * (2 * 64 * 8) + n2 * 32 * 8);
  };
  load_w2(0);
  __builtin_amdgcn_sched_barrier(0);
  // The parked tile is WAVE-PRIVATE (ew = this wave's 12 KB: written and read by this wave only, and LDS operations of a wave
  // complete in order), and the main loop ended on a barrier: no workgroup barrier is needed in here.
#pragma unroll
  for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int nb = 2 * hb + q;
      const float sc = bsc[nb], sh = bsh[nb];
      const int nq = nb * 32 + 4 * k4;
      char* eb = ew + ((nq >> 4) - 4 * hb) * E::IMGB + (j + 4 * h) * E::PIXB + (nq & 15) * 2;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float t[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int r = 4 * g + c;
            const int m = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int gy = y0 + 4 * mb + (m >> 3), gx = x0 + (m & 7);
            float v = fmaf(acc[mb][nb][r], sc, sh);
            if (a.relu) v = tsr_relu(v);
            if (!(img_ok && gy < a.H && gx < a.W)) v = 0.f;
            t[c] = v;
          }
          quad_transpose(t[0], t[1], t[2], t[3], j);
          fz_bf16x4 p0;
#pragma unroll
          for (int c = 0; c < 4; ++c) p0[c] = (__bf16)t[c];
          *(fz_bf16x4*)(eb + (4 * mb + g) * E::ROWB) = p0;
        }
    }
    // (the wave's own ds_writes above are ordered before its ds_reads below: same wave, same LDS queue; the wave barrier
    //  only keeps the compiler from moving one across the other)
    __builtin_amdgcn_wave_barrier();
    fz_bf16x8 fa[4][2];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) fa[c][mb] = *(const fz_bf16x8*)(ea + c * E::IMGB + (4 * mb) * E::ROWB);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2)
          acc2[mb][n2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[c][mb], fbw[c][n2], acc2[mb][n2], 0, 0, 0);
    __builtin_amdgcn_wave_barrier();
    if (hb == 0) {
      __builtin_amdgcn_sched_barrier(0);
      load_w2(1);          // the first half's accumulators are dead: their registers take the second half's fragments
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // + bias + residual (bf16), ReLU, bf16 store
#pragma unroll
  for (int n2 = 0; n2 < 2; ++n2) {
    const int nq2 = n2 * 32 + 4 * k4;
    const int oq = a.out_coff + nq2;
    const f32x4 sh4 = osh[n2];
    __bf16* ob4 = (__bf16*)a.out + (((size_t)bsafe * out_blocks + (oq >> 4)) * HW) * 16 + (oq & 15);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float t0 = acc2[mb][n2][4 * g + 0], t1 = acc2[mb][n2][4 * g + 1], t2 = acc2[mb][n2][4 * g + 2],
              t3 = acc2[mb][n2][4 * g + 3];
        quad_transpose(t0, t1, t2, t3, j);
        const int t = 4 * mb + g;
        if (okk[t]) {
          f32x4 v = {t0 + sh4[0], t1 + sh4[1], t2 + sh4[2], t3 + sh4[3]};
          const fz_bf16x4 rv = rvv[n2][t];
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] += (float)rv[c];
          fz_bf16x4 o;
#pragma unroll
          for (int c = 0; c < 4; ++c) o[c] = (__bf16)(a.relu2 ? tsr_relu(v[c]) : v[c]);
          *(fz_bf16x4*)(ob4 + pov[t]) = o;
        }
      }
  }
}

// conv_fuse1x1.h's fused second GEMM for the K = 32 convolution kernel (conv_mfma_k32.hip): the MSRB stage-2 tile
// (128 px x 128 ch per workgroup, 16x16 accumulator tiles) gets BN + ReLU, is parked in LDS as two scaled fp16 planes
// and multiplied by its half of the 1x1 `confusion` weight on v_mfma_f32_16x16x32_f16 (K = 32 = two 16-channel
// blocks per instruction, 4 steps), then bias / residual / ReLU and 16-B stores.  See conv_fuse1x1.h for the algebra.
#pragma once
#include "tsr_common.h"
#include "conv_args.h"
#include "conv_epilogue.h"

typedef _Float16 fq_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 fq_f16x4 __attribute__((ext_vector_type(4)));

template <int IMG> struct Fuse1x1Geom16 {
  static constexpr int PIXB = 64;                  // [plane 2][16 ch] fp16 per (block, pixel)
  static constexpr int ROWB = (8 * 4 + 2) * 16;    // row stride == 2 (mod 4) slots: conflict-free A fragments (k32_row_slots)
  static constexpr int IMGB = 8 * ROWB;
  static constexpr int BLKB = IMG * IMGB;          // IMG images per workgroup (2: 256 threads, 4: 512 threads)
  static constexpr int BYTES = 8 * BLKB;           // 128 channels = 8 blocks: 69,632 B / 139,264 B
};

// acc[mt][nt]: lane (cl = lane & 15, ph = lane >> 4) holds channel wn*64 + 16 nt + cl, pixels (2 mt + (ph >> 1), 4 (ph & 1) + r)
template <int IMG>
__device__ __forceinline__ void conv_fuse1x1_epilogue16(const ConvArgs& a, f32x4 (&acc)[4][4], char* lds, int b0, int y0,
                                                        int x0, int wm, int wn, int lane, int HW, float accmul) {
  typedef Fuse1x1Geom16<IMG> E;
  const int cl = lane & 15, ph = lane >> 4;
  const int k4 = cl >> 2, j = cl & 3;
  const int dyl = ph >> 1, dxl = 4 * (ph & 1);
  const int b = b0 + wm;
  const bool img_ok = b < a.B;
  // W2 fragments of the second GEMM (16 x 16 B per lane, from L2): requested first, so that their latency runs under the
  // BN / park phase instead of behind its barriers
  const int khalf = ph & 1, bsel = ph >> 1;
  // 1x1 pack of tsr_pack_conv_weight_f16s: [chunk][plane][k half][64][8]
  const _Float16* wb = (const _Float16*)a.w2 + ((size_t)(bsel * 2 * 2 + khalf) * 64 + wn * 32 + cl) * 8;
  fq_f16x8 fb[4][2][2];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int n2 = 0; n2 < 2; ++n2)
        fb[kk][p][n2] = *(const fq_f16x8*)(wb + (size_t)((2 * kk) * 2 + p) * (2 * 64 * 8) + n2 * 16 * 8);
  // the residual tile of step 4 (x, or the other conv's partial sum): requested here as well, branch-free (a slot outside
  // the image reads the image's first pixel and is dropped) -- inside step 4's bounds branch every one of the eight loads
  // was waited for on its own at the very end of the workgroup
  const int out_blocks = a.out_ctot >> 4, res_blocks = a.res_ctot >> 4;
  const int bsafe = img_ok ? b : 0;
  const int gx = x0 + dxl + j;
  bool okk[4];
  unsigned pov[4];
  f32x4 rvv[2][4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int gy = y0 + 2 * mt + dyl;
    okk[mt] = img_ok && gy < a.H && gx < a.W;
    pov[mt] = okk[mt] ? (unsigned)(gy * a.W + gx) * 16u : 0u;
  }
#pragma unroll
  for (int n2 = 0; n2 < 2; ++n2) {
    const int rq = a.res_coff + wn * 32 + n2 * 16 + 4 * k4;
    const float* rb4 = a.res ? a.res + (((size_t)bsafe * res_blocks + (rq >> 4)) * HW) * 16 + (rq & 15) : nullptr;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) rvv[n2][mt] = rb4 ? *(const f32x4*)(rb4 + pov[mt]) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  // (the BN fold's and the output stage's per-channel values as well: one round trip for everything the epilogue reads)
  float bsc[4], bsh[4];
  f32x4 osh[2];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int n = wn * 64 + nt * 16 + cl;
    bsc[nt] = a.scale ? a.scale[n] : 1.f;
    bsh[nt] = a.shift ? a.shift[n] : 0.f;
  }
#pragma unroll
  for (int n2 = 0; n2 < 2; ++n2)
    osh[n2] = a.shift2 ? *(const f32x4*)(a.shift2 + wn * 32 + n2 * 16 + 4 * k4) : (f32x4){0.f, 0.f, 0.f, 0.f};
  // ---- 1. BatchNorm fold + ReLU in place (lane = channel), zero outside the image, tile maximum
  float amax = 0.f;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const float sc = bsc[nt] * accmul, sh = bsh[nt];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gy = y0 + 2 * mt + dyl, gx = x0 + dxl + r;
        float v = fmaf(acc[mt][nt][r], sc, sh);
        if (a.relu) v = tsr_relu(v);
        if (!(img_ok && gy < a.H && gx < a.W)) v = 0.f;
        acc[mt][nt][r] = v;
        amax = fmaxf(amax, fabsf(v));
      }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
  float* red = (float*)(lds + E::BYTES);
  // (the main loop ended on a barrier: no wave still reads the halo / weight ring)
  if (lane == 0) red[wm * 2 + wn] = amax;
  __syncthreads();
  float tmax = 0.f;
#pragma unroll
  // (a wave whose image holds an Inf does not take part in the tile scale: the other images of the tile keep their result)
  for (int w = 0; w < 2 * IMG; ++w) tmax = fmaxf(tmax, red[w] < 3.0e38f ? red[w] : 0.f);
  float s_e = 1.f;
  if (tmax > 0.f) {
    int e = (int)((__float_as_uint(tmax) >> 23) & 0xFF) - 127;
    int be = 13 - e + 127;
    be = be < 1 ? 1 : (be > 254 ? 254 : be);
    s_e = __uint_as_float((unsigned)be << 23);
  }
  // ---- 2. (lane = channel, register = pixel) -> (lane = pixel, 4 channels), two scaled fp16 planes, into LDS
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int nq = wn * 64 + nt * 16 + 4 * k4;
    char* eb = lds + (nq >> 4) * E::BLKB + wm * E::IMGB + dyl * E::ROWB + (dxl + j) * E::PIXB + (nq & 15) * 2;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      float t0 = acc[mt][nt][0], t1 = acc[mt][nt][1], t2 = acc[mt][nt][2], t3 = acc[mt][nt][3];
      quad_transpose(t0, t1, t2, t3, j);
      float v[4] = {t0 * s_e, t1 * s_e, t2 * s_e, t3 * s_e};
      fq_f16x4 p0, p1;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        p0[c] = (_Float16)v[c];
        p1[c] = (_Float16)(v[c] - (float)p0[c]);
      }
      char* dst = eb + (2 * mt) * E::ROWB;
      *(fq_f16x4*)(dst) = p0;
      *(fq_f16x4*)(dst + 32) = p1;
    }
  }
  __syncthreads();
  // ---- 3. out[64 px of image wm][32 co of half wn] = sum over 128 channels, K = 32 (two blocks) per step
  f32x4 acc2[4][2];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int n2 = 0; n2 < 2; ++n2) acc2[mt][n2] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const char* ea = lds + bsel * E::BLKB + wm * E::IMGB + (cl >> 3) * E::ROWB + (cl & 7) * E::PIXB + khalf * 16;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    fq_f16x8 fa[2][4];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) fa[p][mt] = *(const fq_f16x8*)(ea + (2 * kk) * E::BLKB + (2 * mt) * E::ROWB + p * 32);
    }
    // small terms first: h2.g1, h1.g2, h1.g1
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int n2 = 0; n2 < 2; ++n2)
        acc2[mt][n2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[1][mt], fb[kk][0][n2], acc2[mt][n2], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int n2 = 0; n2 < 2; ++n2)
        acc2[mt][n2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[0][mt], fb[kk][1][n2], acc2[mt][n2], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int n2 = 0; n2 < 2; ++n2)
        acc2[mt][n2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[0][mt], fb[kk][0][n2], acc2[mt][n2], 0, 0, 0);
  }
  // ---- 4. + bias + residual, ReLU, max|out|, 16-B stores
  const float mul2 = a.w2_inv_scale / s_e;
  float omax = 0.f;
#pragma unroll
  for (int n2 = 0; n2 < 2; ++n2) {
    const int nq2 = wn * 32 + n2 * 16 + 4 * k4;
    const int oq = a.out_coff + nq2;
    const f32x4 sh4 = osh[n2];
    float* ob4 = a.out + (((size_t)bsafe * out_blocks + (oq >> 4)) * HW) * 16 + (oq & 15);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      float t0 = acc2[mt][n2][0], t1 = acc2[mt][n2][1], t2 = acc2[mt][n2][2], t3 = acc2[mt][n2][3];
      quad_transpose(t0, t1, t2, t3, j);
      if (okk[mt]) {
        f32x4 v = {t0, t1, t2, t3};
        const f32x4 rv = rvv[n2][mt];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          v[c] = v[c] * mul2 + sh4[c] + rv[c];
          if (a.relu2) v[c] = tsr_relu(v[c]);
          omax = fmaxf(omax, fabsf(v[c]));
        }
        *(f32x4*)(ob4 + pov[mt]) = v;
      }
    }
  }
  if (a.out_amax) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) omax = fmaxf(omax, __shfl_xor(omax, o));
    if (lane == 0) publish_amax(a.out_amax, omax);
  }
}

// Output stage of the K = 32 convolution kernel (conv_mfma_k32.hip): the 16x16 accumulator tiles of
// v_mfma_f32_16x16x32_* -> global memory, in the same three forms as conv_epilogue.h (epi_mode 0 / 1 / 2).
//
// Accumulator geometry of one wave (64 pixels = one 8x8 patch of one image, NT 16-channel blocks):
//   acc[mt][nt][r], lane = (cl = lane & 15, ph = lane >> 4):
//     channel 16*nt + cl,  pixel row 2*mt + (ph >> 1),  pixel column 4*(ph & 1) + r.
// As in conv_epilogue.h a 4x4 transpose inside every lane quad (registers r = 4 adjacent pixels <-> lanes cl & 3 = 4
// adjacent channels) turns that into (lane = pixel, registers = 4 consecutive channels), so residual / mask reads and
// the stores are 16 B per lane and whole 64-B CB16 lines per quad of quads.
#pragma once
#include "tsr_common.h"
#include "conv_args.h"
#include "conv_epilogue.h"

template <int COUT, bool EXT, int WN = 2, bool IO16 = false>
__device__ __forceinline__ void conv_epilogue16(const ConvArgs& a, f32x4 (&acc)[4][COUT / (16 * WN)], int ebase, int b0,
                                                int y0, int x0, int wm, int wn, int lane, int HW, float accmul = 1.f) {
  constexpr int NT = COUT / (16 * WN);
  const int cl = lane & 15, ph = lane >> 4;
  const int k4 = cl >> 2, j = cl & 3;
  const int dyl = ph >> 1, dxl = 4 * (ph & 1);            // this lane's pixel row within a tile / first pixel column
  float amax = 0.f;
  const int b = b0 + wm;
  const bool img_ok = b < a.B;
  const int out_blocks = a.out_ctot >> 4;
  const int res_blocks = a.res_ctot >> 4;
  const int mask_blocks = a.mask_ctot >> 4;
  const int bsafe = img_ok ? b : 0;
  typedef __bf16 ep_bf16x4 __attribute__((ext_vector_type(4)));
  const int gx = x0 + dxl + j;                            // pixel column after the transpose
  // Per-channel vectors of ALL channel blocks, requested before the first store: behind a store to a.out the compiler may
  // not hoist them (the pointers could alias), and every block would start with an exposed L2 round trip.
  const f32x4 one4v = {1.f, 1.f, 1.f, 1.f}, zero4v = {0.f, 0.f, 0.f, 0.f};
  f32x4 pv0[NT], pv1[NT], pv2[NT], pv3[NT], pv4[NT];
  const bool m2 = EXT && a.epi_mode == 2, m0 = !EXT || a.epi_mode == 0;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int nq = wn * (COUT / WN) + nt * 16 + 4 * k4;
    pv0[nt] = one4v; pv1[nt] = zero4v; pv2[nt] = one4v; pv3[nt] = zero4v; pv4[nt] = zero4v;
    if (m0) {             // scale, shift, residual scale, residual shift
      if (a.scale) pv0[nt] = *(const f32x4*)(a.scale + nq);
      if (a.shift) pv1[nt] = *(const f32x4*)(a.shift + nq);
      if (EXT && a.res_scale) { pv2[nt] = *(const f32x4*)(a.res_scale + nq); pv3[nt] = *(const f32x4*)(a.res_shift + nq); }
    } else if (m2) {      // scale, mask scale, mask shift, bn_a, bn_b
      if (a.scale) pv0[nt] = *(const f32x4*)(a.scale + nq);
      if (a.mask_scale) { pv2[nt] = *(const f32x4*)(a.mask_scale + nq); pv1[nt] = *(const f32x4*)(a.mask_shift + nq); }
      if (a.bn_a) { pv3[nt] = *(const f32x4*)(a.bn_a + nq); pv4[nt] = *(const f32x4*)(a.bn_b + nq); }
    }
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = wn * (COUT / WN) + nt * 16 + cl;        // channel before the transpose
    const int nq = wn * (COUT / WN) + nt * 16 + 4 * k4;   // first of this lane's 4 channels after it
    const int oq = a.out_coff + nq;
    const size_t oidx4 = (((size_t)bsafe * out_blocks + (oq >> 4)) * HW) * 16 + (oq & 15);
    float* ob4 = a.out + oidx4;
    const float* rb4 = nullptr;
    size_t ridx4 = 0;
    if (a.res) {
      const int rq = a.res_coff + nq;
      ridx4 = (((size_t)bsafe * res_blocks + (rq >> 4)) * HW) * 16 + (rq & 15);
      rb4 = a.res + ridx4;
    }
    if (!EXT || a.epi_mode == 0) {
      const f32x4 sc4 = pv0[nt], sh4 = pv1[nt], rsc4 = pv2[nt], rsh4 = pv3[nt];
      // residual tile of this channel block: requested together, branch-free (dropped slots read the image's first pixel)
      f32x4 rvv[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int gy = y0 + 2 * mt + dyl;
        const unsigned pr = (img_ok && gy < a.H && gx < a.W) ? (unsigned)(gy * a.W + gx) * 16u : 0u;
        rvv[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (rb4) {
          if (IO16) {
            const ep_bf16x4 r16 = *(const ep_bf16x4*)((const __bf16*)a.res + ridx4 + pr);
#pragma unroll
            for (int c = 0; c < 4; ++c) rvv[mt][c] = (float)r16[c];
          } else {
            rvv[mt] = *(const f32x4*)(rb4 + pr);
          }
        }
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        float t0 = acc[mt][nt][0], t1 = acc[mt][nt][1], t2 = acc[mt][nt][2], t3 = acc[mt][nt][3];
        quad_transpose(t0, t1, t2, t3, j);
        const int gy = y0 + 2 * mt + dyl;
        if (img_ok && gy < a.H && gx < a.W) {
          const size_t po = (size_t)(gy * a.W + gx) * 16;
          f32x4 v = {t0, t1, t2, t3};
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] = (v[c] * accmul) * sc4[c] + sh4[c];
          if (rb4) {
            f32x4 rv = rvv[mt];
            if (EXT && a.res_scale) {
#pragma unroll
              for (int c = 0; c < 4; ++c) rv[c] = tsr_relu(fmaf(rv[c], rsc4[c], rsh4[c]));
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] += rv[c];
          }
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            if (a.relu) v[c] = tsr_relu(v[c]);
            amax = fmaxf(amax, fabsf(v[c]));
          }
          if (IO16) {
            ep_bf16x4 o16;
#pragma unroll
            for (int c = 0; c < 4; ++c) o16[c] = (__bf16)v[c];
            *(ep_bf16x4*)((__bf16*)a.out + oidx4 + po) = o16;
          } else {
            *(f32x4*)(ob4 + po) = v;
          }
        }
      }
    } else if (EXT && a.epi_mode == 1) {
      // raw accumulator out + Welford partial (mean, M2) of this wave's valid pixels of channel n
      float cnt = 0.f, sum = 0.f;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gy = y0 + 2 * mt + dyl, gxr = x0 + dxl + r;
          acc[mt][nt][r] *= accmul;
          if (img_ok && gy < a.H && gxr < a.W) {
            const float v = acc[mt][nt][r];
            cnt += 1.f;
            sum += v;
            amax = fmaxf(amax, fabsf(v));
          }
        }
      }
      float mean = cnt > 0.f ? sum / cnt : 0.f;
      float m2 = 0.f;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gy = y0 + 2 * mt + dyl, gxr = x0 + dxl + r;
          if (img_ok && gy < a.H && gxr < a.W) {
            const float d = acc[mt][nt][r] - mean;
            m2 = fmaf(d, d, m2);
          }
        }
      }
      // Chan merge over the four lanes (ph = 0..3) that hold the same channel
#pragma unroll
      for (int o = 16; o <= 32; o <<= 1) {
        const float cnt2 = __shfl_xor(cnt, o), mean2 = __shfl_xor(mean, o), m22 = __shfl_xor(m2, o);
        const float tot = cnt + cnt2;
        const float dlt = mean2 - mean;
        const float meanm = tot > 0.f ? mean + dlt * (cnt2 / tot) : 0.f;
        const float m2m = tot > 0.f ? m2 + m22 + dlt * dlt * (cnt * cnt2 / tot) : 0.f;
        cnt = tot; mean = meanm; m2 = m2m;
      }
      if (ph == 0) {
        const size_t e = (size_t)ebase + wm;      // (workgroup, image slot) entry
        float* sl = a.slab + (e * COUT + n) * 2;
        sl[0] = mean;
        sl[1] = m2;
        if (nt == 0 && cl == 0 && wn == 0) a.slab_cnt[e] = cnt;
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        float t0 = acc[mt][nt][0], t1 = acc[mt][nt][1], t2 = acc[mt][nt][2], t3 = acc[mt][nt][3];
        quad_transpose(t0, t1, t2, t3, j);
        const int gy = y0 + 2 * mt + dyl;
        if (img_ok && gy < a.H && gx < a.W) *(f32x4*)(ob4 + (size_t)(gy * a.W + gx) * 16) = (f32x4){t0, t1, t2, t3};
      }
    } else if (EXT) {
      // ReLU backward by the stored activation (+ optional BN-backward partial sums), transposed layout
      const int mq = a.mask_coff + nq;
      const float* mb4 = a.mask + (((size_t)bsafe * mask_blocks + (mq >> 4)) * HW) * 16 + (mq & 15);
      const f32x4 zero4 = zero4v;
      const f32x4 sc4 = pv0[nt], msc4 = pv2[nt], msh4 = pv1[nt], ba4 = pv3[nt], bb4 = pv4[nt];
      f32x4 s1 = zero4, s2 = zero4;
      // the stored activation and the partial gradient of the four tiles are requested together, branch-free (a slot
      // outside the image reads the image's first pixel and is dropped): one memory latency per channel block instead of
      // one per tile -- the loads used to sit inside the bounds branch, each waited for on its own
      f32x4 mvv[4], rvv[4];
      bool okk[4];
      unsigned pov[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int gy = y0 + 2 * mt + dyl;
        okk[mt] = img_ok && gy < a.H && gx < a.W;
        pov[mt] = okk[mt] ? (unsigned)(gy * a.W + gx) * 16u : 0u;
        mvv[mt] = *(const f32x4*)(mb4 + pov[mt]);
        rvv[mt] = rb4 ? *(const f32x4*)(rb4 + pov[mt]) : zero4;
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        float t0 = acc[mt][nt][0], t1 = acc[mt][nt][1], t2 = acc[mt][nt][2], t3 = acc[mt][nt][3];
        quad_transpose(t0, t1, t2, t3, j);
        f32x4 v = {t0, t1, t2, t3};
        const f32x4 mv = okk[mt] ? mvv[mt] : zero4, rv = rvv[mt];      // (a dropped slot must not feed NaN into the sums)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float x = (v[c] * accmul) * sc4[c] + rv[c];
          if (!(fmaf(mv[c], msc4[c], msh4[c]) > 0.f)) x = 0.f;
          if (!okk[mt]) x = 0.f;
          v[c] = x;
          amax = fmaxf(amax, fabsf(x));
          s1[c] += x;
          s2[c] = fmaf(x, fmaf(mv[c], ba4[c], bb4[c]), s2[c]);
        }
        if (okk[mt]) *(f32x4*)(ob4 + pov[mt]) = v;
      }
      if (a.bn_a) {
        // the 16 lanes (4 of the quad x 4 pixel groups) that hold the same 4 channels
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          s1[c] += __shfl_xor(s1[c], 1); s2[c] += __shfl_xor(s2[c], 1);
          s1[c] += __shfl_xor(s1[c], 2); s2[c] += __shfl_xor(s2[c], 2);
          s1[c] += __shfl_xor(s1[c], 16); s2[c] += __shfl_xor(s2[c], 16);
          s1[c] += __shfl_xor(s1[c], 32); s2[c] += __shfl_xor(s2[c], 32);
        }
        if (ph == 0 && j == 0) {
          const size_t e = (size_t)ebase + wm;      // (workgroup, image slot) entry
          float* sl = a.slab + (e * COUT + nq) * 2;
#pragma unroll
          for (int c = 0; c < 4; ++c) { sl[2 * c] = s1[c]; sl[2 * c + 1] = s2[c]; }
        }
      }
    }
  }
  if (a.out_amax) {     // one atomic per wave
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    if (lane == 0) publish_amax(a.out_amax, amax);
  }
}

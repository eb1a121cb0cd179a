// Shared output stage of the convolution kernels (fp32 MFMA and split-bf16 MFMA): the 32x32 accumulator
// tiles of one wave -> global memory, in one of three forms (see ConvArgs / tsr_conv_desc):
//   epi_mode 0: y = act(acc*scale + shift (+ residual))                      (inference, non-BN train layers)
//   epi_mode 1: raw accumulator + per-(workgroup,image) Welford partials     (train-mode BatchNorm statistics)
//   epi_mode 2: (acc*scale (+res)) * [mask > 0] (+ BatchNorm-backward sums)  (dgrad)
#pragma once
#include "tsr_common.h"
#include "conv_args.h"

// 4x4 transpose inside every quad of lanes: before, lane j of the quad holds (t0..t3) = its value for 4 items;
// after, lane j holds item j's values of the quad's lanes 0..3.  Two butterfly exchanges (lane^1, lane^2) on DPP.
__device__ __forceinline__ void quad_transpose(float& t0, float& t1, float& t2, float& t3, int j) {
  const float s01 = (j & 1) ? t0 : t1, s23 = (j & 1) ? t2 : t3;
  const float x01 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s01), 0xB1, 0xF, 0xF, true));
  const float x23 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s23), 0xB1, 0xF, 0xF, true));
  if (j & 1) { t0 = x01; t2 = x23; } else { t1 = x01; t3 = x23; }
  const float s02 = (j & 2) ? t0 : t2, s13 = (j & 2) ? t1 : t3;
  const float x02 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s02), 0x4E, 0xF, 0xF, true));
  const float x13 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s13), 0x4E, 0xF, 0xF, true));
  if (j & 2) { t0 = x02; t1 = x13; } else { t2 = x02; t3 = x13; }
}

// accmul: exact power-of-two factor undoing the operand scaling of the fp16-split path (1 otherwise).
// WN = waves across C_out (2: each wave owns C_out/2 of one of 2 images; 1: each wave owns all C_out of one of 4).
// IO16: output, residual and mask tensors are bf16 CB16 (the "bf16" configurations: inference and training epilogues).
template <int COUT, bool EXT, int WN = 2, bool IO16 = false>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[2][COUT / (32 * WN)], int bid, int b0,
                                              int y0, int x0, int wm, int wn, int h, int li, int HW,
                                              float accmul = 1.f) {
  constexpr int NB = COUT / (32 * WN);
  constexpr int IMG = 4 / WN;
  float amax = 0.f;
  const int b = b0 + wm;
  const bool img_ok = b < a.B;
  const int out_blocks = a.out_ctot >> 4;
  const int res_blocks = a.res_ctot >> 4;
  const int mask_blocks = a.mask_ctot >> 4;
  const int bsafe = img_ok ? b : 0;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = wn * (COUT / WN) + nb * 32 + li;
    const int oc = a.out_coff + n;
    if (!EXT || a.epi_mode == 0) {
      // y = acc*scale + shift (+ residual) (ReLU) -> channel slice of out.
      // The accumulator holds (lane = channel, register = pixel); a 4x4 transpose inside every lane quad (registers
      // r&3 = 4 adjacent pixels of a row <-> lanes li&3 = 4 adjacent channels) turns that into (lane = pixel,
      // registers = 4 consecutive channels), so the residual read and the store are 16 B per lane and one
      // instruction covers whole 64-B CB16 lines (eight pixels x two channel blocks) instead of 4-B pieces.
      const int k4 = li >> 2, j = li & 3;                       // channel quad of this nb block, pixel within the row half
      const int nq = wn * (COUT / WN) + nb * 32 + 4 * k4;       // first of this lane's 4 channels after the transpose
      const int oq = a.out_coff + nq;
      f32x4 sc4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f}, rsc4 = sc4, rsh4 = sh4;
      if (a.scale) sc4 = *(const f32x4*)(a.scale + nq);
      if (a.shift) sh4 = *(const f32x4*)(a.shift + nq);
      if (EXT && a.res_scale) { rsc4 = *(const f32x4*)(a.res_scale + nq); rsh4 = *(const f32x4*)(a.res_shift + nq); }
      const size_t oidx4 = (((size_t)bsafe * out_blocks + (oq >> 4)) * HW) * 16 + (oq & 15);
      const float* rb4 = nullptr;
      size_t ridx4 = 0;
      if (a.res) {
        const int rq = a.res_coff + nq;
        ridx4 = (((size_t)bsafe * res_blocks + (rq >> 4)) * HW) * 16 + (rq & 15);
        rb4 = a.res;
      }
      const int gx = x0 + j + 4 * h;
      // residual rows of this channel block: requested together, branch-free (dropped slots read the image's first pixel)
      f32x4 rvv[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int gy = y0 + t;
        const unsigned pr = (img_ok && gy < a.H && gx < a.W) ? (unsigned)(gy * a.W + gx) * 16u : 0u;
        rvv[t] = rb4 ? tsr_ld4<IO16>(a.res, ridx4 + pr) : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float t0 = acc[mb][nb][4 * g + 0], t1 = acc[mb][nb][4 * g + 1], t2 = acc[mb][nb][4 * g + 2],
                t3 = acc[mb][nb][4 * g + 3];
          quad_transpose(t0, t1, t2, t3, j);
          const int gy = y0 + 4 * mb + g;
          if (img_ok && gy < a.H && gx < a.W) {
            const size_t po = (size_t)(gy * a.W + gx) * 16;
            f32x4 v = {t0, t1, t2, t3};
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = (v[c] * accmul) * sc4[c] + sh4[c];
            if (rb4) {
              f32x4 rv = rvv[4 * mb + g];
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
            tsr_st4<IO16>(a.out, oidx4 + po, v);
          }
        }
      }
    } else if (EXT && a.epi_mode == 1) {
      // raw accumulator out + Welford partial (mean, M2) of this wave's valid pixels
      float cnt = 0.f, sum = 0.f;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = (r & 3) + 8 * (r >> 2) + 4 * h;
          const int gy = y0 + 4 * mb + (m >> 3), gx = x0 + (m & 7);
          acc[mb][nb][r] *= accmul;
          if (img_ok && gy < a.H && gx < a.W) {
            const float v = acc[mb][nb][r];
            cnt += 1.f;
            sum += v;
            amax = fmaxf(amax, fabsf(v));
          }
        }
      }
      float mean = cnt > 0.f ? sum / cnt : 0.f;
      float m2 = 0.f;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = (r & 3) + 8 * (r >> 2) + 4 * h;
          const int gy = y0 + 4 * mb + (m >> 3), gx = x0 + (m & 7);
          if (img_ok && gy < a.H && gx < a.W) {
            const float d = acc[mb][nb][r] - mean;
            m2 = fmaf(d, d, m2);
          }
        }
      }
      // Chan merge with the partner half-wave (same channel, other 4 pixel columns)
      const float cnt2 = __shfl_xor(cnt, 32), mean2 = __shfl_xor(mean, 32), m22 = __shfl_xor(m2, 32);
      const float nt = cnt + cnt2;
      const float dlt = mean2 - mean;
      const float meanm = nt > 0.f ? mean + dlt * (cnt2 / nt) : 0.f;
      const float m2m = nt > 0.f ? m2 + m22 + dlt * dlt * (cnt * cnt2 / nt) : 0.f;
      if (h == 0) {
        const size_t e = (size_t)bid * IMG + wm;
        float* sl = a.slab + (e * COUT + n) * 2;
        sl[0] = meanm;
        sl[1] = m2m;
        if (nb == 0 && li == 0 && wn == 0) a.slab_cnt[e] = nt;
      }
      {   // raw accumulator -> out, as 16-B stores after the quad transpose (see mode 0)
        const int k4 = li >> 2, j = li & 3;
        const int oq = a.out_coff + wn * (COUT / WN) + nb * 32 + 4 * k4;
        const size_t oidx4 = (((size_t)bsafe * out_blocks + (oq >> 4)) * HW) * 16 + (oq & 15);
        const int gx = x0 + j + 4 * h;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float t0 = acc[mb][nb][4 * g + 0], t1 = acc[mb][nb][4 * g + 1], t2 = acc[mb][nb][4 * g + 2],
                  t3 = acc[mb][nb][4 * g + 3];
            quad_transpose(t0, t1, t2, t3, j);
            const int gy = y0 + 4 * mb + g;
            if (img_ok && gy < a.H && gx < a.W)
              tsr_st4<IO16>(a.out, oidx4 + (size_t)(gy * a.W + gx) * 16, (f32x4){t0, t1, t2, t3});
          }
      }
    } else if (EXT) {
      // ReLU backward by the stored activation (+ optional BN-backward partial sums), in the transposed layout of
      // mode 0: residual, mask and output move as 16 B per lane; the per-channel sums are then reduced over the 8
      // lanes (4 of the quad x 2 wave halves) that hold the same 4 channels.
      const int k4 = li >> 2, j = li & 3;
      const int nq = wn * (COUT / WN) + nb * 32 + 4 * k4;
      const int oq = a.out_coff + nq, mq = a.mask_coff + nq;
      const size_t oidx4 = (((size_t)bsafe * out_blocks + (oq >> 4)) * HW) * 16 + (oq & 15);
      const size_t midx4 = (((size_t)bsafe * mask_blocks + (mq >> 4)) * HW) * 16 + (mq & 15);
      size_t ridx4 = 0;
      if (a.res) {
        const int rq = a.res_coff + nq;
        ridx4 = (((size_t)bsafe * res_blocks + (rq >> 4)) * HW) * 16 + (rq & 15);
      }
      f32x4 one4 = {1.f, 1.f, 1.f, 1.f}, zero4 = {0.f, 0.f, 0.f, 0.f};
      const f32x4 sc4 = a.scale ? *(const f32x4*)(a.scale + nq) : one4;
      const f32x4 msc4 = a.mask_scale ? *(const f32x4*)(a.mask_scale + nq) : one4;
      const f32x4 msh4 = a.mask_scale ? *(const f32x4*)(a.mask_shift + nq) : zero4;
      const f32x4 ba4 = a.bn_a ? *(const f32x4*)(a.bn_a + nq) : zero4;
      const f32x4 bb4 = a.bn_a ? *(const f32x4*)(a.bn_b + nq) : zero4;
      f32x4 s1 = zero4, s2 = zero4;
      const int gx = x0 + j + 4 * h;
      // the stored activation and the partial gradient of the eight tile rows are requested together, branch-free (a slot
      // outside the image reads the image's first pixel and is dropped): the loads used to sit inside the bounds branch,
      // each waited for on its own
      f32x4 mvv[8], rvv[8];
      bool okk[8];
      unsigned pov[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int gy = y0 + t;                   // t = 4 mb + g
        okk[t] = img_ok && gy < a.H && gx < a.W;
        pov[t] = okk[t] ? (unsigned)(gy * a.W + gx) * 16u : 0u;
        mvv[t] = tsr_ld4<IO16>(a.mask, midx4 + pov[t]);
        rvv[t] = a.res ? tsr_ld4<IO16>(a.res, ridx4 + pov[t]) : zero4;
      }
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float t0 = acc[mb][nb][4 * g + 0], t1 = acc[mb][nb][4 * g + 1], t2 = acc[mb][nb][4 * g + 2],
                t3 = acc[mb][nb][4 * g + 3];
          quad_transpose(t0, t1, t2, t3, j);
          const int t = 4 * mb + g;
          f32x4 v = {t0, t1, t2, t3};
          const f32x4 mv = okk[t] ? mvv[t] : zero4, rv = rvv[t];      // (a dropped slot must not feed NaN into the sums)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            float x = (v[c] * accmul) * sc4[c] + rv[c];
            if (!(fmaf(mv[c], msc4[c], msh4[c]) > 0.f)) x = 0.f;
            if (!okk[t]) x = 0.f;
            v[c] = x;
            amax = fmaxf(amax, fabsf(x));
            s1[c] += x;
            s2[c] = fmaf(x, fmaf(mv[c], ba4[c], bb4[c]), s2[c]);
          }
          if (okk[t]) tsr_st4<IO16>(a.out, oidx4 + pov[t], v);
        }
      if (a.bn_a) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          s1[c] += __shfl_xor(s1[c], 1); s2[c] += __shfl_xor(s2[c], 1);
          s1[c] += __shfl_xor(s1[c], 2); s2[c] += __shfl_xor(s2[c], 2);
          s1[c] += __shfl_xor(s1[c], 32); s2[c] += __shfl_xor(s2[c], 32);
        }
        if (h == 0 && j == 0) {
          const size_t e = (size_t)bid * IMG + wm;
          float* sl = a.slab + (e * COUT + nq) * 2;
#pragma unroll
          for (int c = 0; c < 4; ++c) { sl[2 * c] = s1[c]; sl[2 * c + 1] = s2[c]; }
        }
      }
    }
  }
  if (a.out_amax) {     // one atomic per wave: |v| >= 0, so the uint order of the bit patterns is the float order
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    if ((h | li) == 0) publish_amax(a.out_amax, amax);
  }
}

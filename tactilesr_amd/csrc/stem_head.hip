// Boundary kernels of TactileSR.forward: the taxel stem and the image head.
//
//  * tsr_stem_fwd: nn.Upsample(scale_factor, bilinear, align_corners=False) + Conv2d(
//    axisCnt -> 64, 3x3, pad 1, no bias) + [BatchNorm eval as scale/shift] + ReLU
//    (model/tactileSR_model.py:34-39 pattern stem, :59-63 force stem).  The upsampled
//    image is never written to HBM: a row band is rebuilt in LDS from the 4x4 taxels.
//    Reads NCHW taxels (48 floats / sample), writes CB16.  HBM-write bound.
//  * tsr_head_fwd: Conv2d(C -> 1, 3x3, pad 1, no bias) + ReLU (model/tactileSR_model.py:
//    55-56); the trailing F.interpolate to the same size (:83) is an exact identity and
//    is elided.  Reads CB16, writes the NCHW (B,1,H,W) result.  HBM-read bound.
//  * layout converters NCHW <-> CB16 (test / probe plumbing).
#include "tsr_common.h"
#include "conv_epilogue.h"      // quad_transpose

// ATen upsample_bilinear2d (align_corners=False) source index.
__device__ __forceinline__ void bilin_src(int dst, float scale, int n_in, int& i0, int& i1, float& lam) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  i0 = i0 < n_in - 1 ? i0 : n_in - 1;
  i1 = i0 + 1 < n_in ? i0 + 1 : n_in - 1;
  lam = src - (float)i0;
}

// One workgroup = a band of RB output rows of one image: the upsampled band (+ halo) is formed in LDS, then the 3x3 conv
// 3 -> 64 runs as a matrix product on v_mfma_f32_16x16x32_bf16 with the 27 (channel, tap) products as K (padded to 32) and
// channels as rows: D[channel][pixel].  Both operands are fp32 values, so each enters as THREE bf16 planes (x = x1 + x2 + x3,
// 24 significant bits) and a product is all nine plane pairs -- exact products, fp32 accumulation, small terms first:
// fp32-grade, like the VALU form it replaces (1728 FMAs per pixel: the kernel was VALU-bound at 0.54 / 0.75 ms for 0.84 / 1.7
// GB written at B = 4096).  The B operand of lane (pixel n, k group g) is 8 LDS reads of the upsampled band; the
// accumulator's 4 channels per lane are 16 B (8 B in bf16) of the CB16 output: a wave's store covers whole lines.
typedef __bf16 sh_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 sh_bf16x4 __attribute__((ext_vector_type(4)));

// OUT16: the output tensor is bf16 CB16 (the "bf16" activation-storage path); same arithmetic, rounded on store.
template <int CIN, bool OUT16 = false>
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ lr, int lr_ctot, int lr_coff,
                                                   int hin, int win, int sf,
                                                   const float* __restrict__ w,      // OIHW (64,CIN,3,3)
                                                   const float* __restrict__ scale,
                                                   const float* __restrict__ shift,
                                                   float* __restrict__ out, int out_ctot, int out_coff,
                                                   int relu, int B, int RB, int IPW, float* __restrict__ out_amax) {
  static_assert(CIN * 9 <= 32, "one K step");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = hin * sf, W = win * sf;
  const int WP = ((W + 3) & ~3) + 2;      // padded row of the upsampled band (zero beyond the image)
  float* tax = smem;                      // [CIN][hin*win]
  float* up = tax + ((CIN * hin * win + 3) & ~3);  // [CIN][RB+2][WP]
  const int tid = threadIdx.x;
  const int y0 = blockIdx.x * RB;
  const int rows = (H - y0) < RB ? (H - y0) : RB;

  // ---- operands (once per workgroup; it then walks IPW images).  K slot k = 8 g + j of lane (n, g) is (channel c, tap kh,
  // kw) = (k / 9, (k % 9) / 3, k % 3); slots >= 9 CIN carry a zero weight (and read band element 0).  A: row n of tile mt =
  // output channel 16 mt + n, three planes.
  const int lane = tid & 63, wv = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  int koff[8];
  sh_bf16x8 A[3][4];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 8 * g + j;
    const bool real = k < 9 * CIN;
    const int c = k / 9, r9 = k - c * 9, kh = r9 / 3, kw = r9 - kh * 3;
    koff[j] = real ? (c * (RB + 2) + kh) * WP + kw : 0;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      float v = real ? w[((mt * 16 + n) * CIN + c) * 9 + r9] : 0.f;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        const __bf16 q = (__bf16)v;
        A[pl][mt][j] = q;
        v -= (float)q;
      }
    }
  }
  f32x4 scv[4], shv[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    scv[mt] = scale ? *(const f32x4*)(scale + mt * 16 + 4 * g) : (f32x4){1.f, 1.f, 1.f, 1.f};
    shv[mt] = shift ? *(const f32x4*)(shift + mt * 16 + 4 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const float sc = 1.0f / (float)sf;
  const int nup = CIN * (RB + 2) * WP;
  const int HW = H * W;
  const int npx = rows * W;
  const int out_blocks = out_ctot >> 4;
  float amax = 0.f;

  for (int b = blockIdx.y * IPW; b < B && b < (blockIdx.y + 1) * IPW; ++b) {
    __syncthreads();                                 // (the previous image's band has been read)
    for (int i = tid; i < CIN * hin * win; i += 256)
      tax[i] = lr[((size_t)b * lr_ctot + lr_coff) * hin * win + i];
    __syncthreads();
    for (int i = tid; i < nup; i += 256) {
      const int c = i / ((RB + 2) * WP);
      const int rem = i - c * ((RB + 2) * WP);
      const int yy = rem / WP, xx = rem - yy * WP;
      const int gy = y0 - 1 + yy, gx = xx - 1;
      float v = 0.f;
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
        int ya, yb, xa, xb;
        float ly, lx;
        bilin_src(gy, sc, hin, ya, yb, ly);
        bilin_src(gx, sc, win, xa, xb, lx);
        const float* t = tax + c * hin * win;
        const float top = (1.f - lx) * t[ya * win + xa] + lx * t[ya * win + xb];
        const float bot = (1.f - lx) * t[yb * win + xa] + lx * t[yb * win + xb];
        v = (1.f - ly) * top + ly * bot;
      }
      up[i] = v;
    }
    __syncthreads();

    const size_t obase = ((size_t)b * out_blocks + (out_coff >> 4)) * HW * 16 + 4 * g;      // + mt * HW * 16 + pixel * 16
    for (int q = wv; q * 16 < npx; q += 4) {
      const int i = q * 16 + n;
      const bool live = i < npx;
      const int ic = live ? i : npx - 1;
      const int y = ic / W, x = ic - y * W;
      const float* ub = up + y * WP + x;
      sh_bf16x8 U[3];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = ub[koff[j]];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          const __bf16 qv = (__bf16)v;
          U[pl][j] = qv;
          v -= (float)qv;
        }
      }
      const size_t po = obase + (size_t)((y0 + y) * W + x) * 16;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};                           // all nine plane pairs, smallest terms first
#pragma unroll
        for (int s9 = 4; s9 >= 0; --s9)
#pragma unroll
          for (int pa = 2; pa >= 0; --pa)
            if (s9 - pa >= 0 && s9 - pa <= 2) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[pa][mt], U[s9 - pa], acc, 0, 0, 0);
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float t = acc[c] * scv[mt][c] + shv[mt][c];
          o[c] = relu ? tsr_relu(t) : t;
          if (live) amax = fmaxf(amax, fabsf(o[c]));
        }
        if (live) tsr_st4<OUT16>(out, po + (size_t)mt * HW * 16, o);
      }
    }
  }
  if (out_amax) {      // max |output| for the fp16-split consumer's power-of-two scale
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    if ((tid & 63) == 0) publish_amax(out_amax, amax);
  }
}

static int stem_fwd_impl(const float* lr, int lr_ctot, int lr_coff, int axis_cnt, int hin, int win, int sf,
                         const float* w_oihw, const float* scale, const float* shift,
                         float* out, int out_ctot, int out_coff, int relu, int B, float* out_amax, void* stream,
                         bool out16) {
  if (!lr || !w_oihw || !out || B <= 0 || axis_cnt != 3 || hin <= 0 || win <= 0 || sf <= 0) return TSR_ERR_ARG;
  if ((out_ctot & 15) || (out_coff & 15) || out_coff + 64 > out_ctot || lr_coff + axis_cnt > lr_ctot)
    return TSR_ERR_ARG;
  const int H = hin * sf, W = win * sf;
  const int WP = ((W + 3) & ~3) + 2;
  // rows per workgroup: the upsampled band (RB + 2 rows of 3 channels) within 32 KB of LDS; among the fitting sizes the one
  // whose pixel count fills its 64-pixel passes (4 waves x 16 pixels) best
  const size_t fixed = (size_t)((3 * hin * win + 3) & ~3) * 4;
  int RB = 0;
  double best = -1.0;
  for (int rb = 2; rb <= H; ++rb) {
    if (fixed + (size_t)3 * (rb + 2) * WP * 4 > 32 * 1024) break;
    long items = 0, slots = 0;
    for (int y = 0; y < H; y += rb) {
      const int r = H - y < rb ? H - y : rb;
      items += (long)r * W;
      slots += ((long)r * W + 63) / 64 * 64;
    }
    const double eff = (double)items / (double)slots;
    if (eff > best + 1e-9) { best = eff; RB = rb; }
  }
  if (RB == 0) return TSR_ERR_ARG;
  const size_t smem = fixed + (size_t)3 * (RB + 2) * WP * 4;
  // images per workgroup (the weight planes are built once per workgroup): as many as still leave >= 4 workgroups per CU
  const int bands = (H + RB - 1) / RB;
  int IPW = (int)(((long long)B * bands) / (256 * 4));
  IPW = IPW < 1 ? 1 : (IPW > 16 ? 16 : IPW);
  dim3 grid(bands, (B + IPW - 1) / IPW);
  if (out16)
    hipLaunchKernelGGL((stem_kernel<3, true>), grid, dim3(256), smem, (hipStream_t)stream, lr, lr_ctot, lr_coff, hin,
                       win, sf, w_oihw, scale, shift, out, out_ctot, out_coff, relu, B, RB, IPW, out_amax);
  else
    hipLaunchKernelGGL((stem_kernel<3, false>), grid, dim3(256), smem, (hipStream_t)stream, lr, lr_ctot, lr_coff, hin,
                       win, sf, w_oihw, scale, shift, out, out_ctot, out_coff, relu, B, RB, IPW, out_amax);
  return tsr_check_launch();
}

extern "C" int tsr_stem_fwd(const float* lr, int lr_ctot, int lr_coff, int axis_cnt, int hin, int win, int sf,
                            const float* w_oihw, const float* scale, const float* shift,
                            float* out, int out_ctot, int out_coff, int relu, int B, float* out_amax, void* stream) {
  return stem_fwd_impl(lr, lr_ctot, lr_coff, axis_cnt, hin, win, sf, w_oihw, scale, shift, out, out_ctot, out_coff, relu,
                       B, out_amax, stream, false);
}

// bf16 activation storage: the output is bf16 CB16 (`out` addresses bf16 elements)
extern "C" int tsr_stem_fwd_b16(const float* lr, int lr_ctot, int lr_coff, int axis_cnt, int hin, int win, int sf,
                                const float* w_oihw, const float* scale, const float* shift,
                                void* out_bf16, int out_ctot, int out_coff, int relu, int B, void* stream) {
  return stem_fwd_impl(lr, lr_ctot, lr_coff, axis_cnt, hin, win, sf, w_oihw, scale, shift, (float*)out_bf16, out_ctot,
                       out_coff, relu, B, nullptr, stream, true);
}

// ---------------------------------------------------------------------------------------
// Head (conv 128 -> 1, 3x3).  Thread = (pixel, channel quad): the four lanes of a pixel read the four 16-B quarters of
// each 64-B CB16 line; the partial dot products meet through two DPP adds.
// LDS-tiled form: a thread-per-pixel kernel that reads its nine taps straight from global memory asks L1 / the texture
// path for every input line nine times (72 16-B load instructions per thread; rounds 1-2 ran that at 2.1 TB/s = 0.26 of
// HBM although its HBM traffic was the algorithmic 3.36 GB).  Here a workgroup owns an 8x8 pixel patch of one image, stages the 10x10
// halo of one 16-channel block per step (1.6 global loads per thread instead of 9; double-buffered, one barrier per
// block) and takes the nine taps from LDS: thread = (pixel, channel quad) as before, so a 16-lane group reads 256
// contiguous bytes (4 pixels x 64 B): bank-conflict free without padding.
template <bool IN16>
__global__ __launch_bounds__(256) void head_lds_kernel(const float* __restrict__ in, int in_ctot, int cin,
                                                       const float* __restrict__ w, float* __restrict__ out, int relu,
                                                       int B, int H, int W, int tiles_x, int tiles_y) {
  extern __shared__ __attribute__((aligned(16))) float hl[];   // [cin/16][9][16] weights, then 2 x [100 px][16] halo
  const int tid = threadIdx.x;
  const int nblk = cin >> 4;
  for (int i = tid; i < nblk * 9 * 16; i += 256) {
    const int j = i & 15, r = i >> 4;
    const int tap = r % 9, blk = r / 9;
    hl[i] = w[(blk * 16 + j) * 9 + tap];
  }
  float* halo = hl + nblk * 9 * 16;
  const int HW = H * W, tpi = tiles_x * tiles_y;
  int b, trem;
  {   // XCD-aware order: the patches of one image (they share halo rows) run on one XCD / L2
    const int nwg = gridDim.x;
    const int qn = nwg >> 3, rn = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int logical = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx;
    b = logical / tpi;
    trem = logical - b * tpi;
  }
  const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
  const int y0 = ty * 8, x0 = tx * 8;
  const int in_blocks = in_ctot >> 4;
  // staging items (halo pixel, quad): this thread's two slots
  int soff[2], sdst[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int it = tid + 256 * k;
    soff[k] = -1;
    sdst[k] = -1;
    if (it < 400) {
      const int qd = it & 3, hp = it >> 2;
      const int hy = hp / 10, hx = hp - hy * 10;
      const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
      sdst[k] = hp * 16 + qd * 4;
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) soff[k] = (gy * W + gx) * 16 + qd * 4;
    }
  }
  const size_t ibase = (size_t)b * in_blocks * HW * 16;
  auto load = [&](int blk, f32x4* hv) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      hv[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (soff[k] >= 0) hv[k] = tsr_ld4<IN16>(in, ibase + (size_t)blk * HW * 16 + soff[k]);
    }
  };
  auto store = [&](int buf, const f32x4* hv) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (sdst[k] >= 0) *(f32x4*)(halo + buf * 1600 + sdst[k]) = hv[k];
  };
  const int q = tid & 3, p = tid >> 2, py = p >> 3, px = p & 7;
  f32x4 hv[2];
  load(0, hv);
  store(0, hv);
  if (nblk > 1) load(1, hv);
  __syncthreads();
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (int blk = 0; blk < nblk; ++blk) {
    const float* hb = halo + (blk & 1) * 1600 + (py * 10 + px) * 16 + 4 * q;
    const float* wb = hl + blk * 9 * 16 + 4 * q;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      float s = 0.f;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const f32x4 v = *(const f32x4*)(hb + (kh * 10 + kw) * 16);
        const f32x4 wv = *(const f32x4*)(wb + (kh * 3 + kw) * 16);
        s = fmaf(v[0], wv[0], fmaf(v[1], wv[1], fmaf(v[2], wv[2], fmaf(v[3], wv[3], s))));
      }
      if (kh == 0) a0 += s; else if (kh == 1) a1 += s; else a2 += s;
    }
    if (blk + 1 < nblk) {
      store((blk + 1) & 1, hv);             // the other buffer: last read before the barrier that ended step blk - 1
      if (blk + 2 < nblk) load(blk + 2, hv);
    }
    __syncthreads();
  }
  float v = (a0 + a1) + a2;
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // lane ^ 1
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // lane ^ 2
  if (relu) v = tsr_relu(v);
  const int gy = y0 + py, gx = x0 + px;
  if (q == 0 && gy < H && gx < W) out[(size_t)b * HW + gy * W + gx] = v;
}

template <bool IN16>
static int head_launch(const float* in, int in_ctot, int cin, const float* w_oihw, float* out_nchw, int relu, int B, int H,
                       int W, void* stream) {
  const int tiles_x = (W + 7) / 8, tiles_y = (H + 7) / 8;
  const size_t smem = (size_t)cin * 9 * 4 + 2 * 1600 * 4;
  hipLaunchKernelGGL(head_lds_kernel<IN16>, dim3(B * tiles_x * tiles_y), dim3(256), smem, (hipStream_t)stream, in, in_ctot,
                     cin, w_oihw, out_nchw, relu, B, H, W, tiles_x, tiles_y);
  return tsr_check_launch();
}

extern "C" int tsr_head_fwd(const float* in, int in_ctot, int cin, const float* w_oihw, float* out_nchw,
                            int relu, int B, int H, int W, void* stream) {
  if (!in || !w_oihw || !out_nchw || B <= 0 || (cin & 15) || (in_ctot & 15) || cin > in_ctot || cin <= 0)
    return TSR_ERR_ARG;
  return head_launch<false>(in, in_ctot, cin, w_oihw, out_nchw, relu, B, H, W, stream);
}

// Head on a bf16 CB16 input as a matrix product (bf16 activation storage; cin = 64 / 128).  The LDS-tiled kernel above moves
// 1.2 TB/s on bf16 tensors: 8 barriers per 8x8 patch, a 10x10 halo per patch (1.56x the bytes through L2), 8-B loads.  Here
//   D[tap][pixel] = sum_c w[c][tap] * h0[c][pixel]      (v_mfma_f32_16x16x32_bf16, channels as K, the 9 taps as rows)
// is formed for every pixel of a band of rows (+ one halo row above and below) straight from global memory -- the B operand
// of lane (pixel n, k group g) is ONE 16-B load: 8 consecutive channels of its pixel -- then the nine shifted planes are summed
// from LDS: out[y][x] = sum_(kh,kw) D[kh*3+kw][y+kh-1][x+kw-1].  The stored activation is exact in bf16; the fp32 weight enters
// as three bf16 planes w1 + w2 + w3 (24 significant bits), three MFMAs per K step, fp32 accumulation: fp32-grade like the FMA
// form, in a different summation order.
typedef __bf16 hm_bf16x8 __attribute__((ext_vector_type(8)));
template <int NK>
__global__ __launch_bounds__(256) void head_mfma_b16_kernel(const char* __restrict__ in, int in_ctot, const float* __restrict__ w,
                                                            float* __restrict__ out, int relu, int H, int W, int R, int bands) {
  extern __shared__ __attribute__((aligned(16))) float dl[];      // [9 taps][RW + pad]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / bands, band = blockIdx.x - b * bands;
  const int y0 = band * R;
  const int rows = (H - y0 < R ? H - y0 : R) + 2;
  const int RW = rows * W, RWP = (R + 2) * W + 1;
  const int HW = H * W;
  // weights: row (tap) n of the A operand, channels 32 kk + 8 g .. + 7, three bf16 planes
  hm_bf16x8 A[3][NK];
#pragma unroll
  for (int kk = 0; kk < NK; ++kk)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = n < 9 ? w[(kk * 32 + 8 * g + j) * 9 + n] : 0.f;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        const __bf16 q = (__bf16)v;
        A[pl][kk][j] = q;
        v -= (float)q;
      }
    }
  const size_t plane = (size_t)HW * 32;
  const char* ib = in + (size_t)b * (in_ctot >> 4) * plane + (size_t)(g >> 1) * plane + (g & 1) * 16;
  const int P0 = (y0 - 1) * W;                                   // image pixel index of region pixel 0
  const int ngrp = (RW + 15) >> 4;
  for (int q0 = wv; q0 < ngrp; q0 += 8) {                        // two groups in flight per wave
    hm_bf16x8 Bf[2][NK];
    int r[2];
    bool ok[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int q = q0 + 4 * u;
      r[u] = q * 16 + n;
      const int P = P0 + r[u];
      ok[u] = q < ngrp && r[u] < RW && P >= 0 && P < HW;
      const unsigned po = (unsigned)(ok[u] ? P : 0) * 32u;
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) Bf[u][kk] = *(const hm_bf16x8*)(ib + (size_t)(2 * kk) * plane + po);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int pl = 2; pl >= 0; --pl)                              // small terms first
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[pl][kk], Bf[u][kk], acc, 0, 0, 0);
      if (q0 + 4 * u < ngrp && r[u] < RW) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (4 * g + j < 9) dl[(4 * g + j) * RWP + r[u]] = ok[u] ? acc[j] : 0.f;
      }
    }
  }
  __syncthreads();
  const int npx = (rows - 2) * W;
  for (int i = threadIdx.x; i < npx; i += 256) {
    const int y = i / W, x = i - y * W;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const float* d = dl + (kh * 3) * RWP + (y + kh) * W + x;
      const float l = x > 0 ? d[-1] : 0.f, c = d[RWP], rr = x + 1 < W ? d[2 * RWP + 1] : 0.f;
      const float t = (l + c) + rr;
      if (kh == 0) s0 = t; else if (kh == 1) s1 = t; else s2 = t;
    }
    float v = (s0 + s1) + s2;
    if (relu) v = tsr_relu(v);
    out[(size_t)b * HW + (y0 + y) * W + x] = v;
  }
}

static int head_mfma_launch(const void* in, int in_ctot, int cin, const float* w, float* out, int relu, int B, int H, int W,
                            void* stream) {
  int R = 32 * 1024 / (W * 36) - 2;                               // band rows: nine fp32 planes of (R + 2) rows within 32 KB
  if (R > H) R = H;
  if (R < 1) return -1;                                           // (very wide images: the LDS-tiled kernel)
  const int bands = (H + R - 1) / R;
  const size_t smem = (size_t)9 * ((R + 2) * W + 1) * 4;
  if ((long long)B * bands > 0x7fffffffLL) return -1;
  if (cin == 128)
    hipLaunchKernelGGL(head_mfma_b16_kernel<4>, dim3(B * bands), dim3(256), smem, (hipStream_t)stream, (const char*)in, in_ctot, w,
                       out, relu, H, W, R, bands);
  else
    hipLaunchKernelGGL(head_mfma_b16_kernel<2>, dim3(B * bands), dim3(256), smem, (hipStream_t)stream, (const char*)in, in_ctot, w,
                       out, relu, H, W, R, bands);
  return tsr_check_launch();
}

// bf16 activation storage: `in` is bf16 CB16; the image comes out fp32 NCHW as always
extern "C" int tsr_head_fwd_b16(const void* in_bf16, int in_ctot, int cin, const float* w_oihw, float* out_nchw,
                                int relu, int B, int H, int W, void* stream) {
  if (!in_bf16 || !w_oihw || !out_nchw || B <= 0 || (cin & 15) || (in_ctot & 15) || cin > in_ctot || cin <= 0)
    return TSR_ERR_ARG;
  if (cin == 128 || cin == 64) {
    const int st = head_mfma_launch(in_bf16, in_ctot, cin, w_oihw, out_nchw, relu, B, H, W, stream);
    if (st >= 0) return st;
  }
  return head_launch<true>((const float*)in_bf16, in_ctot, cin, w_oihw, out_nchw, relu, B, H, W, stream);
}

// ---------------------------------------------------------------------------------------
__global__ void nchw_to_cb16_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int HW,
                                    int dst_ctot, int dst_coff, size_t total) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int pix = i % HW;
    size_t r = i / HW;
    const int c = r % C;
    const int b = r / C;
    dst[cb16_index(b, dst_coff + c, pix, dst_ctot, HW)] = src[i];
  }
}

__global__ void cb16_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int HW,
                                    int src_ctot, int src_coff, size_t total) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int pix = i % HW;
    size_t r = i / HW;
    const int c = r % C;
    const int b = r / C;
    dst[i] = src[cb16_index(b, src_coff + c, pix, src_ctot, HW)];
  }
}

extern "C" int tsr_nchw_to_cb16(const float* src, float* dst, int B, int C, int HW, int dst_ctot,
                                int dst_coff, void* stream) {
  if (!src || !dst || (dst_ctot & 15) || dst_coff + C > dst_ctot) return TSR_ERR_ARG;
  const size_t total = (size_t)B * C * HW;
  const size_t g = (total + 255) / 256;
  hipLaunchKernelGGL(nchw_to_cb16_kernel, dim3(g > 65535 ? 65535 : (int)g), dim3(256), 0,
                     (hipStream_t)stream, src, dst, C, HW, dst_ctot, dst_coff, total);
  return tsr_check_launch();
}

extern "C" int tsr_cb16_to_nchw(const float* src, float* dst, int B, int C, int HW, int src_ctot,
                                int src_coff, void* stream) {
  if (!src || !dst || (src_ctot & 15) || src_coff + C > src_ctot) return TSR_ERR_ARG;
  const size_t total = (size_t)B * C * HW;
  const size_t g = (total + 255) / 256;
  hipLaunchKernelGGL(cb16_to_nchw_kernel, dim3(g > 65535 ? 65535 : (int)g), dim3(256), 0,
                     (hipStream_t)stream, src, dst, C, HW, src_ctot, src_coff, total);
  return tsr_check_launch();
}

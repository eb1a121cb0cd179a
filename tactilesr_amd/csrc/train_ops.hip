// Training-path kernels around the MFMA convolutions (all HBM-bound or tiny):
//   BatchNorm2d train mode (batch statistics + running-stat update, backward coefficients),
//   the 3->64 stem and 128->1 head backward, target preparation, MSE loss fwd/bwd, Adam(L2).
// Reference semantics: nn.BatchNorm2d(eps=1e-5, momentum=0.1) inside model/tactileSR_model.py:
// 38-49,167-189; nn.MSELoss + HR/10 + F.interpolate of train/tactileSR_train.py:39-49;
// optim.Adam(lr, weight_decay) (L2-in-grad) of train/tactileSR_train.py:212; the step order of
// cpu/trainer.py:346-362.
#include "tsr_common.h"
#include "tactilesr_hip.h"

#define RED_BLOCKS 512     // partial-sum blocks of the slab reductions (work buffers hold RED_BLOCKS*C*3 doubles)

// ------------------------------------------------------------------------------------------
// per-channel fp64 reduction of the conv epilogue slabs: slab[entry][C][2]
//   mode 0 (Welford partials mean,M2 with counts): S0 = sum n, S1 = sum n*mean, S2 = sum (M2 + n*mean^2)
//   mode 1 (plain sums s1,s2):                      S0 = entries, S1 = sum s1, S2 = sum s2
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slab,
                                                          const float* __restrict__ cnt, int entries, int C,
                                                          int mode, double* __restrict__ part) {
  __shared__ double sh[256 * 3];
  const int tid = threadIdx.x;
  const int per = 256 / C;           // entries handled concurrently by one block (C in {64,128})
  const int c = tid % C, el = tid / C;
  double s0 = 0, s1 = 0, s2 = 0, t0 = 0, t1 = 0, t2 = 0;     // two independent chains: two loads in flight
  const int stride = gridDim.x * per;
  int e = blockIdx.x * per + el;
  for (; e + stride < entries; e += 2 * stride) {
    const float2 v = *(const float2*)(slab + ((size_t)e * C + c) * 2);
    const float2 u = *(const float2*)(slab + ((size_t)(e + stride) * C + c) * 2);
    if (mode == 0) {
      const double n = (double)cnt[e], m = (double)v.x, n2 = (double)cnt[e + stride], m2 = (double)u.x;
      s0 += n; s1 += n * m; s2 += (double)v.y + n * m * m;
      t0 += n2; t1 += n2 * m2; t2 += (double)u.y + n2 * m2 * m2;
    } else {
      s0 += 1.0; s1 += (double)v.x; s2 += (double)v.y;
      t0 += 1.0; t1 += (double)u.x; t2 += (double)u.y;
    }
  }
  if (e < entries) {
    const float2 v = *(const float2*)(slab + ((size_t)e * C + c) * 2);
    if (mode == 0) {
      const double n = (double)cnt[e], m = (double)v.x;
      s0 += n; s1 += n * m; s2 += (double)v.y + n * m * m;
    } else {
      s0 += 1.0; s1 += (double)v.x; s2 += (double)v.y;
    }
  }
  s0 += t0; s1 += t1; s2 += t2;
  sh[tid * 3 + 0] = s0; sh[tid * 3 + 1] = s1; sh[tid * 3 + 2] = s2;
  __syncthreads();
  if (el == 0) {
    for (int k = 1; k < per; ++k) {
      s0 += sh[(k * C + c) * 3 + 0]; s1 += sh[(k * C + c) * 3 + 1]; s2 += sh[(k * C + c) * 3 + 2];
    }
    double* p = part + ((size_t)blockIdx.x * C + c) * 3;
    p[0] = s0; p[1] = s1; p[2] = s2;
  }
}

__global__ void bn_stats_final_kernel(const double* __restrict__ part, int nblocks, int C,
                                      const float* __restrict__ bias, const float* __restrict__ gamma,
                                      const float* __restrict__ beta, float* __restrict__ running_mean,
                                      float* __restrict__ running_var, float momentum, float eps,
                                      float* __restrict__ scale, float* __restrict__ shift,
                                      float* __restrict__ xa, float* __restrict__ xb) {
  // 16 channels per block; 16 lanes per channel walk the partial blocks, then one lane finishes
  __shared__ double sh[256 * 3];
  const int tid = threadIdx.x, cl = tid & 15, kl = tid >> 4;
  const int c = blockIdx.x * 16 + cl;
  double n = 0, s1 = 0, s2 = 0;
  for (int k = kl; k < nblocks; k += 16) {
    const double* p = part + ((size_t)k * C + c) * 3;
    n += p[0]; s1 += p[1]; s2 += p[2];
  }
  sh[tid * 3 + 0] = n; sh[tid * 3 + 1] = s1; sh[tid * 3 + 2] = s2;
  __syncthreads();
  if (kl != 0) return;
  for (int k = 1; k < 16; ++k) {
    n += sh[(k * 16 + cl) * 3 + 0]; s1 += sh[(k * 16 + cl) * 3 + 1]; s2 += sh[(k * 16 + cl) * 3 + 2];
  }
  const double mean = s1 / n;
  double var = s2 / n - mean * mean;          // biased (normalisation)
  var = var < 0 ? 0 : var;
  const double invstd = 1.0 / sqrt(var + (double)eps);
  const double sc = (double)gamma[c] * invstd;
  scale[c] = (float)sc;
  shift[c] = (float)((double)beta[c] - mean * sc);
  xa[c] = (float)invstd;
  xb[c] = (float)(-mean * invstd);
  if (running_mean) {
    const double mz = mean + (bias ? (double)bias[c] : 0.0);   // stats were taken on the bias-free accumulator
    const double unbiased = n > 1 ? var * n / (n - 1) : var;
    running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mz);
    running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
  }
}

extern "C" int tsr_bn_stats_finalize(const float* slab, const float* slab_cnt, int entries, int C,
                                     const float* bias, const float* gamma, const float* beta,
                                     float* running_mean, float* running_var, float momentum, float eps,
                                     float* scale, float* shift, float* xhat_a, float* xhat_b,
                                     double* work, void* stream) {
  if (!slab || !slab_cnt || !gamma || !beta || !scale || !shift || !xhat_a || !xhat_b || !work || entries <= 0 ||
      (C != 64 && C != 128))
    return TSR_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(RED_BLOCKS), dim3(256), 0, st, slab, slab_cnt, entries, C, 0, work);
  hipLaunchKernelGGL(bn_stats_final_kernel, dim3(C / 16), dim3(256), 0, st, work, RED_BLOCKS, C, bias, gamma, beta,
                     running_mean, running_var, momentum, eps, scale, shift, xhat_a, xhat_b);
  return tsr_check_launch();
}

__global__ void bn_bwd_final_kernel(const double* __restrict__ part, int nblocks, int C, double N,
                                    const float* __restrict__ scale, const float* __restrict__ xa,
                                    const float* __restrict__ xb, float* __restrict__ dgamma,
                                    float* __restrict__ dbeta, float* __restrict__ c1, float* __restrict__ c2,
                                    float* __restrict__ c3) {
  __shared__ double sh[256 * 2];
  const int tid = threadIdx.x, cl = tid & 15, kl = tid >> 4;
  const int c = blockIdx.x * 16 + cl;
  double s1 = 0, s2 = 0;
  for (int k = kl; k < nblocks; k += 16) {
    const double* p = part + ((size_t)k * C + c) * 3;
    s1 += p[1]; s2 += p[2];
  }
  sh[tid * 2 + 0] = s1; sh[tid * 2 + 1] = s2;
  __syncthreads();
  if (kl != 0) return;
  for (int k = 1; k < 16; ++k) { s1 += sh[(k * 16 + cl) * 2 + 0]; s2 += sh[(k * 16 + cl) * 2 + 1]; }
  dbeta[c] = (float)s1;
  dgamma[c] = (float)s2;
  // dz = scale*(g - dbeta/N - xhat*dgamma/N), xhat = z*xa + xb  ->  dz = c1*g + c2*z + c3
  const double sc = scale[c];
  c1[c] = (float)sc;
  c2[c] = (float)(-sc * s2 / N * (double)xa[c]);
  c3[c] = (float)(-sc * (s1 / N + (double)xb[c] * s2 / N));
}

extern "C" int tsr_bn_bwd_finalize(const float* slab, int entries, int C, double N, const float* scale,
                                   const float* xhat_a, const float* xhat_b, float* dgamma, float* dbeta,
                                   float* c1, float* c2, float* c3, double* work, void* stream) {
  if (!slab || !scale || !xhat_a || !xhat_b || !dgamma || !dbeta || !c1 || !c2 || !c3 || !work || entries <= 0 ||
      (C != 64 && C != 128))
    return TSR_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(RED_BLOCKS), dim3(256), 0, st, slab, (const float*)nullptr, entries,
                     C, 1, work);
  hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(C / 16), dim3(256), 0, st, work, RED_BLOCKS, C, N, scale, xhat_a, xhat_b,
                     dgamma, dbeta, c1, c2, c3);
  return tsr_check_launch();
}

// g[:, gcoff:gcoff+C] = c1*g + c2*z[:, zcoff:zcoff+C] + c3   (BN backward, elementwise, CB16, in place)
// One workgroup per (image, 16-channel block): its HW x 16 values are contiguous; a thread keeps one channel quad (c1..c3 in
// 12 registers, no index arithmetic per element) and has UNR independent 16-B load pairs in flight.
template <bool B16>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(float* __restrict__ g, int g_ctot, int g_coff,
                                                           const float* __restrict__ z, int z_ctot, int z_coff,
                                                           const float* __restrict__ c1, const float* __restrict__ c2,
                                                           const float* __restrict__ c3, int C, int HW,
                                                           float* __restrict__ out_amax) {
  constexpr int UNR = 5;
  const int nblk = C >> 4;
  const int b = blockIdx.x / nblk, blk = blockIdx.x - b * nblk;
  const int c = blk * 16 + (threadIdx.x & 3) * 4;
  const size_t gbase = (((size_t)b * (g_ctot >> 4) + ((g_coff >> 4) + blk)) * HW) * 16;
  const size_t zbase = (((size_t)b * (z_ctot >> 4) + ((z_coff >> 4) + blk)) * HW) * 16;
  const f32x4 k1 = *(const f32x4*)(c1 + c), k2 = *(const f32x4*)(c2 + c), k3 = *(const f32x4*)(c3 + c);
  const int n4 = HW * 4;
  float amax = 0.f;
  for (int i0 = threadIdx.x; i0 < n4; i0 += 256 * UNR) {
    f32x4 gv[UNR], zv[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int i = i0 + u * 256;
      const size_t o = (size_t)(i < n4 ? i : i0) * 4;       // (tail: re-read the first item, not stored)
      gv[u] = tsr_ld4<B16>(g, gbase + o);
      zv[u] = tsr_ld4<B16>(z, zbase + o);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int i = i0 + u * 256;
      if (i < n4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          gv[u][j] = fmaf(k1[j], gv[u][j], fmaf(k2[j], zv[u][j], k3[j]));
          amax = fmaxf(amax, fabsf(gv[u][j]));
        }
        tsr_st4<B16>(g, gbase + (size_t)i * 4, gv[u]);
      }
    }
  }
  if (out_amax) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    if ((threadIdx.x & 63) == 0) publish_amax(out_amax, amax);
  }
}

extern "C" int tsr_bn_bwd_apply(float* g, int g_ctot, int g_coff, const float* z, int z_ctot, int z_coff,
                                const float* c1, const float* c2, const float* c3, int C, int B, int HW,
                                float* out_amax, void* stream) {
  if (!g || !z || !c1 || !c2 || !c3 || (C & 15) || (g_ctot & 15) || (g_coff & 15) || (z_ctot & 15) || (z_coff & 15) ||
      g_coff + C > g_ctot || z_coff + C > z_ctot)
    return TSR_ERR_ARG;
  if (B <= 0 || HW <= 0 || (long long)B * (C >> 4) > 0x7fffffffLL) return TSR_ERR_ARG;
  hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(B * (C >> 4)), dim3(256), 0, (hipStream_t)stream,
                     g, g_ctot, g_coff, z, z_ctot, z_coff, c1, c2, c3, C, HW, out_amax);
  return tsr_check_launch();
}

// the same on bf16 CB16 tensors (training with bf16 activation storage): fp32 arithmetic, bf16 load / store
extern "C" int tsr_bn_bwd_apply_b16(void* g, int g_ctot, int g_coff, const void* z, int z_ctot, int z_coff,
                                    const float* c1, const float* c2, const float* c3, int C, int B, int HW,
                                    void* stream) {
  if (!g || !z || !c1 || !c2 || !c3 || (C & 15) || (g_ctot & 15) || (g_coff & 15) || (z_ctot & 15) || (z_coff & 15) ||
      g_coff + C > g_ctot || z_coff + C > z_ctot)
    return TSR_ERR_ARG;
  if (B <= 0 || HW <= 0 || (long long)B * (C >> 4) > 0x7fffffffLL) return TSR_ERR_ARG;
  hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(B * (C >> 4)), dim3(256), 0, (hipStream_t)stream,
                     (float*)g, g_ctot, g_coff, (const float*)z, z_ctot, z_coff, c1, c2, c3, C, HW, (float*)nullptr);
  return tsr_check_launch();
}

// ------------------------------------------------------------------------------------------
// stem backward: dW[64][3][3][3] of Upsample+Conv2d(3->64) (no dgrad: taxels carry no grad)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void bilin_src_t(int dst, float scale, int n_in, int& i0, int& i1, float& lam) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  i0 = i0 < n_in - 1 ? i0 : n_in - 1;
  i1 = i0 + 1 < n_in ? i0 + 1 : n_in - 1;
  lam = src - (float)i0;
}

// one workgroup per (image split, band of rows); thread = (co = tid&63, pixel phase = tid>>6).  `bands` > 1 for tall images:
// a workgroup then holds only its band of the upsampled image (+ one halo row each side) in LDS -- at 100 x 100 the whole
// image is 125 KB, i.e. one workgroup per CU -- and the slab entries are (band, image split) pairs, nsplit in all.
template <bool B16>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ lr, int lr_ctot, int lr_coff,
                                                         int hin, int win, int sf, const float* __restrict__ dz,
                                                         int dz_ctot, int dz_coff, float* __restrict__ slab, int B,
                                                         int nsplit, int bands) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Himg = hin * sf, W = win * sf, WP = W + 2;
  const int RBAND = (Himg + bands - 1) / bands;
  const int yb0 = blockIdx.y * RBAND;
  const int H = Himg - yb0 < RBAND ? Himg - yb0 : RBAND;      // rows of this band
  const int HP = RBAND + 2;
  float* tax = smem;                                   // [3][hin*win]
  float* up = smem + ((3 * hin * win + 3) & ~3);       // [3][HP][WP] zero padded: band rows yb0 - 1 .. yb0 + RBAND
  const int tid = threadIdx.x;
  const int co = tid & 63, ph = tid >> 6;
  const int HW = Himg * W;
  float acc[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) acc[k] = 0.f;
  const float sc = 1.0f / (float)sf;
  const int oc = dz_coff + co;
  nsplit /= bands;                                     // image splits
  for (int b = blockIdx.x; b < B; b += nsplit) {
    __syncthreads();
    for (int i = tid; i < 3 * hin * win; i += 256) tax[i] = lr[((size_t)b * lr_ctot + lr_coff) * hin * win + i];
    __syncthreads();
    for (int i = tid; i < 3 * HP * WP; i += 256) {
      const int c = i / (HP * WP), rem = i - c * (HP * WP);
      const int gy = yb0 + rem / WP - 1, gx = rem % WP - 1;
      float v = 0.f;
      if (gy >= 0 && gy < Himg && gx >= 0 && gx < W) {
        int ya, yb, xa, xb; float ly, lx;
        bilin_src_t(gy, sc, hin, ya, yb, ly);
        bilin_src_t(gx, sc, win, xa, xb, lx);
        const float* t = tax + c * hin * win;
        const float top = (1.f - lx) * t[ya * win + xa] + lx * t[ya * win + xb];
        const float bot = (1.f - lx) * t[yb * win + xa] + lx * t[yb * win + xb];
        v = (1.f - ly) * top + ly * bot;
      }
      up[i] = v;
    }
    __syncthreads();
    const size_t dzo = (((size_t)b * (dz_ctot >> 4) + (oc >> 4)) * HW) * 16 + (oc & 15);
    // A wave walks whole rows (y = ph, ph + 4, ..) in chunks of PU pixels.  The PU gradient loads of a chunk are requested
    // together (one dependent 2- / 4-byte load per pixel made the loop a chain of memory latencies: 0.78 ms per launch at
    // B = 2048 for 0.42 GB), and the 3x3x3 window of the upsampled image slides in registers: column x + 3 replaces column x
    // (9 LDS reads per pixel instead of 27; the reads are broadcasts -- all 64 lanes of a wave share the pixel).
    constexpr int PU = 8;
    for (int y = ph; y < H; y += 4) {
      for (int x0 = 0; x0 < W; x0 += PU) {
        float d[PU];
#pragma unroll
        for (int u = 0; u < PU; ++u) {
          const int x = x0 + u;
          const float v = tsr_ld1<B16>(dz, dzo + (size_t)((yb0 + y) * W + (x < W ? x : x0)) * 16);
          d[u] = x < W ? v : 0.f;                  // (a chunk's tail beyond the row adds nothing; its window reads stay in the padding)
        }
        float col[3][9];                           // col[(x - x0) % 3][c * 3 + kh] = up[c][y + kh][x]
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int r = 0; r < 9; ++r) col[j][r] = up[((r / 3) * HP + y + (r % 3)) * WP + x0 + j];
#pragma unroll
        for (int u = 0; u < PU; ++u) {
#pragma unroll
          for (int r = 0; r < 9; ++r)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) acc[r * 3 + kw] = fmaf(d[u], col[(u + kw) % 3][r], acc[r * 3 + kw]);
          if (u + 1 < PU) {
            const int xn = x0 + u + 3 < WP ? x0 + u + 3 : WP - 1;          // (past the padded row only under a zero gradient)
#pragma unroll
            for (int r = 0; r < 9; ++r) col[u % 3][r] = up[((r / 3) * HP + y + (r % 3)) * WP + xn];
          }
        }
      }
    }
  }
  // combine the 4 pixel phases through LDS, write this split's OIHW partial
  __syncthreads();
  float* red = smem;   // reuse: [4][64][27]
#pragma unroll
  for (int k = 0; k < 27; ++k) red[(ph * 64 + co) * 27 + k] = acc[k];
  __syncthreads();
  for (int i = tid; i < 64 * 27; i += 256)
    slab[((size_t)blockIdx.y * nsplit + blockIdx.x) * 64 * 27 + i] =
        (red[i] + red[64 * 27 + i]) + (red[2 * 64 * 27 + i] + red[3 * 64 * 27 + i]);
}

static int stem_wgrad_impl(const float* lr, int lr_ctot, int lr_coff, int hin, int win, int sf,
                           const float* dz, int dz_ctot, int dz_coff, float* slab, int nsplit, int B,
                           void* stream, bool b16) {
  if (!lr || !dz || !slab || nsplit <= 0 || B <= 0 || (dz_ctot & 15) || (dz_coff & 15) || dz_coff + 64 > dz_ctot)
    return TSR_ERR_ARG;
  const int H = hin * sf, W = win * sf;
  // bands of rows per image: the band (+ halo) within 32 KB of LDS, if the splits divide (slab entries = nsplit either way)
  int bands = 1;
  while (bands < 8 && (size_t)3 * ((H + bands - 1) / bands + 2) * (W + 2) * 4 > 32 * 1024 && nsplit % (2 * bands) == 0 &&
         nsplit / (2 * bands) >= 1)
    bands *= 2;
  size_t fl = ((3 * hin * win + 3) & ~3) + (size_t)3 * ((H + bands - 1) / bands + 2) * (W + 2);
  if (fl < 4 * 64 * 27) fl = 4 * 64 * 27;
  if (fl * 4 > 160 * 1024) return TSR_ERR_ARG;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)stem_wgrad_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)stem_wgrad_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  if (b16)
    hipLaunchKernelGGL(stem_wgrad_kernel<true>, dim3(nsplit / bands, bands), dim3(256), fl * 4, (hipStream_t)stream, lr, lr_ctot,
                       lr_coff, hin, win, sf, dz, dz_ctot, dz_coff, slab, B, nsplit, bands);
  else
    hipLaunchKernelGGL(stem_wgrad_kernel<false>, dim3(nsplit / bands, bands), dim3(256), fl * 4, (hipStream_t)stream, lr, lr_ctot,
                       lr_coff, hin, win, sf, dz, dz_ctot, dz_coff, slab, B, nsplit, bands);
  return tsr_check_launch();
}

extern "C" int tsr_stem_wgrad(const float* lr, int lr_ctot, int lr_coff, int hin, int win, int sf,
                              const float* dz, int dz_ctot, int dz_coff, float* slab, int nsplit, int B,
                              void* stream) {
  return stem_wgrad_impl(lr, lr_ctot, lr_coff, hin, win, sf, dz, dz_ctot, dz_coff, slab, nsplit, B, stream, false);
}

// dz is a bf16 CB16 tensor (training with bf16 activation storage)
extern "C" int tsr_stem_wgrad_b16(const float* lr, int lr_ctot, int lr_coff, int hin, int win, int sf,
                                  const void* dz, int dz_ctot, int dz_coff, float* slab, int nsplit, int B,
                                  void* stream) {
  return stem_wgrad_impl(lr, lr_ctot, lr_coff, hin, win, sf, (const float*)dz, dz_ctot, dz_coff, slab, nsplit, B, stream, true);
}

// ------------------------------------------------------------------------------------------
// head backward.  out = relu(conv3x3(h0; w[1][C][3][3])), dout given NCHW (B,1,H,W).
//   dpre = dout*[out>0];  dh0[c,q] = sum_tap dpre[q - tap + 1]*w[c][tap];  dz_h0 = dh0*[h0>0]
//   dW[c][tap] = sum_{b,q} dpre[q - tap + 1]*h0[c][q]  (flipped correlation: out pixel p = q - (tap-1))
// ------------------------------------------------------------------------------------------
template <bool B16>
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                       const float* __restrict__ h0, int h_ctot, int cin,
                                                       const float* __restrict__ w, float* __restrict__ dz,
                                                       int dz_ctot, int B, int H, int W, float* __restrict__ out_amax) {
  extern __shared__ __attribute__((aligned(16))) float wl[];   // [9][cin]
  const int tid = threadIdx.x;
  for (int i = tid; i < 9 * cin; i += 256) {
    const int c = i % cin, tap = i / cin;
    wl[i] = w[c * 9 + tap];
  }
  __syncthreads();
  const int HW = H * W, b = blockIdx.y, nblk = cin >> 4;
  // item = (pixel, channel quad), looping over the channel blocks: the nine masked dout neighbours of the pixel are
  // gathered ONCE per thread and reused for every block (round 2 made the block part of the item: each of the 32 threads
  // of a pixel repeated the 18 small loads -- 3.4 ms per step at B = 2048, 1 TB/s on a kernel that moves 3.4 GB).  The
  // four lanes of a quad still cover one 64-B CB16 line, a wave 16 whole lines per load / store instruction.
  const int item = blockIdx.x * 256 + tid;
  float amax = 0.f;
  if (item < HW * 4) {
    const int qd = item & 3, q = item >> 2;
    const int y = q / W, x = q - y * W;
    float dp[9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int py = y - kh + 1, px = x - kw + 1;
        float v = 0.f;
        if (py >= 0 && py < H && px >= 0 && px < W) {
          const size_t o = (size_t)b * HW + py * W + px;
          v = out[o] > 0.f ? dout[o] : 0.f;
        }
        dp[kh * 3 + kw] = v;
      }
    for (int blk = 0; blk < nblk; ++blk) {
      const size_t eo = (((size_t)b * (h_ctot >> 4) + blk) * HW + q) * 16 + qd * 4;
      const size_t zo = (((size_t)b * (dz_ctot >> 4) + blk) * HW + q) * 16 + qd * 4;
      const f32x4 hv = tsr_ld4<B16>(h0, eo);
      f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const f32x4 wv = *(const f32x4*)(wl + t * cin + blk * 16 + qd * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) s4[j] = fmaf(dp[t], wv[j], s4[j]);
      }
      f32x4 r;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        r[j] = hv[j] > 0.f ? s4[j] : 0.f;
        amax = fmaxf(amax, fabsf(r[j]));
      }
      tsr_st4<B16>(dz, zo, r);
    }
  }
  if (out_amax) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    if ((tid & 63) == 0) publish_amax(out_amax, amax);
  }
}

// thread = channel c (cin <= 256 => one thread per channel, remaining threads idle), grid = splits
template <bool B16>
__global__ __launch_bounds__(256) void head_wgrad_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                         const float* __restrict__ h0, int h_ctot, int cin,
                                                         float* __restrict__ slab, int B, int H, int W, int nsplit) {
  extern __shared__ __attribute__((aligned(16))) float dpt[];   // [(H+2)*(W+2)] masked dout, zero padded
  const int tid = threadIdx.x;
  const int HW = H * W, WP = W + 2;
  const int ngrp = 256 / cin;                     // pixel phases
  const int c = tid % cin, ph = tid / cin;
  float acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = 0.f;
  for (int b = blockIdx.x; b < B; b += nsplit) {
    __syncthreads();
    for (int i = tid; i < (H + 2) * WP; i += 256) {
      const int py = i / WP - 1, px = i % WP - 1;
      float v = 0.f;
      if (py >= 0 && py < H && px >= 0 && px < W) {
        const size_t o = (size_t)b * HW + py * W + px;
        v = out[o] > 0.f ? dout[o] : 0.f;
      }
      dpt[i] = v;
    }
    __syncthreads();
    if (ph < ngrp) {
      const size_t ho = (((size_t)b * (h_ctot >> 4) + (c >> 4)) * HW) * 16 + (c & 15);
      // rows y = ph, ph + ngrp, .. in chunks of PU pixels: the chunk's activation loads are requested together (one dependent
      // 2- / 4-byte load per pixel was a chain of memory latencies) and the 3x3 window of the masked output gradient slides in
      // registers (out pixel p = q - (tap - 1); padded index + 1: tap (kh, kw) of pixel x reads column x - kw + 2)
      constexpr int PU = 8;
      for (int y = ph; y < H; y += ngrp) {
        for (int x0 = 0; x0 < W; x0 += PU) {
          float hv[PU];
#pragma unroll
          for (int u = 0; u < PU; ++u) {
            const int x = x0 + u;
            const float v = tsr_ld1<B16>(h0, ho + (size_t)(y * W + (x < W ? x : x0)) * 16);
            hv[u] = x < W ? v : 0.f;
          }
          float col[3][3];                     // col[j % 3][kh] = dpt[y - kh + 2][x0 + j]
#pragma unroll
          for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) col[j][kh] = dpt[(y - kh + 2) * WP + x0 + j];
#pragma unroll
          for (int u = 0; u < PU; ++u) {
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
              for (int kw = 0; kw < 3; ++kw) acc[kh * 3 + kw] = fmaf(hv[u], col[(u + 2 - kw) % 3][kh], acc[kh * 3 + kw]);
            if (u + 1 < PU) {
              const int xn = x0 + u + 3 < WP ? x0 + u + 3 : WP - 1;
#pragma unroll
              for (int kh = 0; kh < 3; ++kh) col[u % 3][kh] = dpt[(y - kh + 2) * WP + xn];
            }
          }
        }
      }
    }
  }
  __syncthreads();
  float* red = dpt;     // [ngrp][cin][9]
  if (ph < ngrp)
#pragma unroll
    for (int t = 0; t < 9; ++t) red[(ph * cin + c) * 9 + t] = acc[t];
  __syncthreads();
  for (int i = tid; i < cin * 9; i += 256) {
    float s = 0.f;
    for (int k = 0; k < ngrp; ++k) s += red[k * cin * 9 + i];
    slab[(size_t)blockIdx.x * cin * 9 + i] = s;
  }
}

template <bool B16>
static int head_bwd_impl(const float* dout, const float* out, const float* h0, int h_ctot, int cin,
                         const float* w_oihw, float* dz_h0, int dz_ctot, float* wslab, int nsplit,
                         int B, int H, int W, float* dz_amax, void* stream) {
  if (!dout || !out || !h0 || !w_oihw || !dz_h0 || !wslab || nsplit <= 0 || (cin & 15) || cin > 256 || cin > h_ctot ||
      cin > dz_ctot || (h_ctot & 15) || (dz_ctot & 15))
    return TSR_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int items = H * W * 4;
  hipLaunchKernelGGL(head_bwd_kernel<B16>, dim3((items + 255) / 256, B), dim3(256), (size_t)9 * cin * 4, st, dout, out,
                     h0, h_ctot, cin, w_oihw, dz_h0, dz_ctot, B, H, W, dz_amax);
  size_t fl = (size_t)(H + 2) * (W + 2);
  const size_t redf = (size_t)(256 / cin) * cin * 9;
  if (fl < redf) fl = redf;
  if (fl * 4 > 64 * 1024) return TSR_ERR_ARG;
  hipLaunchKernelGGL(head_wgrad_kernel<B16>, dim3(nsplit), dim3(256), fl * 4, st, dout, out, h0, h_ctot, cin, wslab, B, H,
                     W, nsplit);
  return tsr_check_launch();
}

extern "C" int tsr_head_bwd(const float* dout, const float* out, const float* h0, int h_ctot, int cin,
                            const float* w_oihw, float* dz_h0, int dz_ctot, float* wslab, int nsplit,
                            int B, int H, int W, float* dz_amax, void* stream) {
  return head_bwd_impl<false>(dout, out, h0, h_ctot, cin, w_oihw, dz_h0, dz_ctot, wslab, nsplit, B, H, W, dz_amax, stream);
}

// h0 and dz_h0 are bf16 CB16 tensors (training with bf16 activation storage); dout / out stay fp32 NCHW
extern "C" int tsr_head_bwd_b16(const float* dout, const float* out, const void* h0, int h_ctot, int cin,
                                const float* w_oihw, void* dz_h0, int dz_ctot, float* wslab, int nsplit,
                                int B, int H, int W, void* stream) {
  return head_bwd_impl<true>(dout, out, (const float*)h0, h_ctot, cin, w_oihw, (float*)dz_h0, dz_ctot, wslab, nsplit, B, H,
                             W, nullptr, stream);
}

// ------------------------------------------------------------------------------------------
// target preparation: HR_raw (B,1,hin,win) * inv_scale -> bilinear (align_corners=False) to (H,W)
// (train/tactileSR_train.py:44-45: HR/HR_scale_num, F.interpolate(size=(4sf,4sf)))
// ------------------------------------------------------------------------------------------
__global__ void target_prep_kernel(const float* __restrict__ hr, float* __restrict__ out, float inv_scale, int B,
                                   int hin, int win, int H, int W) {
  const size_t total = (size_t)B * H * W;
  const float sy = (float)hin / (float)H, sx = (float)win / (float)W;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int x = i % W; size_t r = i / W;
    const int y = r % H; const int b = r / H;
    int ya, yb, xa, xb; float ly, lx;
    bilin_src_t(y, sy, hin, ya, yb, ly);
    bilin_src_t(x, sx, win, xa, xb, lx);
    const float* t = hr + (size_t)b * hin * win;
    const float v00 = t[ya * win + xa] * inv_scale, v01 = t[ya * win + xb] * inv_scale;
    const float v10 = t[yb * win + xa] * inv_scale, v11 = t[yb * win + xb] * inv_scale;
    out[i] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
  }
}

extern "C" int tsr_target_prep(const float* hr_raw, float* out, float inv_scale, int B, int hin, int win, int H,
                               int W, void* stream) {
  if (!hr_raw || !out || B <= 0) return TSR_ERR_ARG;
  const size_t total = (size_t)B * H * W;
  const size_t g = (total + 255) / 256;
  hipLaunchKernelGGL(target_prep_kernel, dim3(g > 8192 ? 8192 : (int)g), dim3(256), 0, (hipStream_t)stream, hr_raw,
                     out, inv_scale, B, hin, win, H, W);
  return tsr_check_launch();
}

// ------------------------------------------------------------------------------------------
// MSE loss (mean over all elements) forward + backward in one pass:
//   partial[block] = sum (y-t)^2 (fp64), dy = gscale * 2 (y-t) / n ; loss = sum(partial)/n (second kernel)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ y, const float* __restrict__ t,
                                                  float* __restrict__ dy, size_t n, float gcoef,
                                                  double* __restrict__ part) {
  __shared__ double sh[256];
  double s = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float d = y[i] - t[i];
    s += (double)d * (double)d;
    if (dy) dy[i] = gcoef * d;
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}

__global__ void mse_final_kernel(const double* __restrict__ part, int nb, double inv_n, float* __restrict__ loss) {
  double s = 0;
  for (int k = 0; k < nb; ++k) s += part[k];
  loss[0] = (float)(s * inv_n);
}

extern "C" int tsr_mse_fwd_bwd(const float* y, const float* target, float* dy, float* loss, long long n,
                               float grad_scale, double* work, void* stream) {
  if (!y || !target || !loss || !work || n <= 0) return TSR_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int nb = 256;
  hipLaunchKernelGGL(mse_kernel, dim3(nb), dim3(256), 0, st, y, target, dy, (size_t)n,
                     (float)(2.0 * grad_scale / (double)n), work);
  hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(1), 0, st, work, nb, 1.0 / (double)n, loss);
  return tsr_check_launch();
}

// ------------------------------------------------------------------------------------------
// Adam with L2-in-gradient weight decay (torch.optim.Adam, not AdamW), one tensor per launch.
//   g' = g + wd*p ; m = b1 m + (1-b1) g' ; v = b2 v + (1-b2) g'^2
//   p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// ------------------------------------------------------------------------------------------
__global__ void adam_l2_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                               float* __restrict__ v, size_t n, float lr, float b1, float b2, float omb1, float omb2,
                               float eps, float wd, float bc1, float bc2_sqrt) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float w = p[i];
    const float gg = fmaf(wd, w, g[i]);
    const float mm = b1 * m[i] + omb1 * gg;
    const float vv = b2 * v[i] + omb2 * gg * gg;
    m[i] = mm;
    v[i] = vv;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    p[i] = w - (lr / bc1) * (mm / denom);
  }
}

extern "C" int tsr_adam_l2_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n,
                                float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                                void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0 || step <= 0) return TSR_ERR_ARG;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const size_t g = ((size_t)n + 255) / 256;
  hipLaunchKernelGGL(adam_l2_kernel, dim3(g > 4096 ? 4096 : (int)g), dim3(256), 0, (hipStream_t)stream, param, grad,
                     exp_avg, exp_avg_sq, (size_t)n, lr, beta1, beta2, (float)(1.0 - (double)beta1),
                     (float)(1.0 - (double)beta2), eps, weight_decay, (float)bc1, (float)sqrt(bc2));
  return tsr_check_launch();
}

// One launch for all parameter tensors: block b owns chunk record b (<= 4096 contiguous elements of one tensor).
// 16-B accesses when the four pointers are 16-B aligned (torch allocations and the gradient arena are).
__global__ __launch_bounds__(256) void adam_l2_multi_kernel(const tsr_adam_chunk* __restrict__ chunks, float lr,
                                                            float b1, float b2, float omb1, float omb2, float eps,
                                                            float wd, float bc1, float bc2_sqrt) {
  const tsr_adam_chunk c = chunks[blockIdx.x];
  const float step_size = lr / bc1;
  auto upd = [&](float w, float g, float& m, float& v) {
    const float gg = fmaf(wd, w, g);
    m = b1 * m + omb1 * gg;
    v = b2 * v + omb2 * gg * gg;
    return w - step_size * (m / (sqrtf(v) / bc2_sqrt + eps));
  };
  const bool vec = ((((size_t)c.param | (size_t)c.grad | (size_t)c.exp_avg | (size_t)c.exp_avg_sq) & 15) == 0);
  const int n4 = vec ? (c.n >> 2) : 0;
  for (int i = threadIdx.x; i < n4; i += 256) {
    f32x4 w = ((f32x4*)c.param)[i], m = ((f32x4*)c.exp_avg)[i], v = ((f32x4*)c.exp_avg_sq)[i];
    const f32x4 g = ((const f32x4*)c.grad)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float mj = m[j], vj = v[j];
      w[j] = upd(w[j], g[j], mj, vj);
      m[j] = mj;
      v[j] = vj;
    }
    ((f32x4*)c.param)[i] = w;
    ((f32x4*)c.exp_avg)[i] = m;
    ((f32x4*)c.exp_avg_sq)[i] = v;
  }
  for (int i = n4 * 4 + threadIdx.x; i < c.n; i += 256) {
    float m = c.exp_avg[i], v = c.exp_avg_sq[i];
    c.param[i] = upd(c.param[i], c.grad[i], m, v);
    c.exp_avg[i] = m;
    c.exp_avg_sq[i] = v;
  }
}

extern "C" int tsr_adam_l2_multi(const tsr_adam_chunk* chunks, int n_chunks, float lr, double beta1, double beta2,
                                 float eps, float weight_decay, int step, void* stream) {
  if (!chunks || n_chunks <= 0 || step <= 0) return TSR_ERR_ARG;
  // betas arrive as doubles and every derived constant is formed in double like torch does on the host:
  // (1.f - 0.999f) is 1.3e-5 off 0.001
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  hipLaunchKernelGGL(adam_l2_multi_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, chunks, lr, (float)beta1,
                     (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), eps, weight_decay, (float)bc1,
                     (float)sqrt(bc2));
  return tsr_check_launch();
}

// ------------------------------------------------------------------------------------------
// Welford batch statistics of a CB16 channel slice (for the VALU stem, which has no stats
// epilogue): entry = (image, 64-pixel chunk); slab/slab_cnt in the tsr_conv2d_ex epi_mode-1 format.
// ------------------------------------------------------------------------------------------
template <bool B16>
__global__ __launch_bounds__(256) void cb16_stats_kernel(const float* __restrict__ z, int z_ctot, int z_coff,
                                                         int HW, int chunks, float* __restrict__ slab,
                                                         float* __restrict__ slab_cnt) {
  __shared__ float sh[256 * 3];
  const int tid = threadIdx.x, c = tid & 63, pg = tid >> 6;
  const int e = blockIdx.x, b = e / chunks, ch = e - b * chunks;
  const int oc = z_coff + c;
  const size_t zo = (((size_t)b * (z_ctot >> 4) + (oc >> 4)) * HW) * 16 + (oc & 15);
  float v[16];
  float cnt = 0.f, sum = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int p = ch * 64 + pg * 16 + k;
    v[k] = 0.f;
    if (p < HW) { v[k] = tsr_ld1<B16>(z, zo + (size_t)p * 16); cnt += 1.f; sum += v[k]; }
  }
  const float mean = cnt > 0.f ? sum / cnt : 0.f;
  float m2 = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int p = ch * 64 + pg * 16 + k;
    if (p < HW) { const float d = v[k] - mean; m2 = fmaf(d, d, m2); }
  }
  sh[tid * 3] = cnt; sh[tid * 3 + 1] = mean; sh[tid * 3 + 2] = m2;
  __syncthreads();
  if (pg == 0) {
    float n = cnt, mu = mean, M = m2;
    for (int k = 1; k < 4; ++k) {
      const float n2 = sh[(k * 64 + c) * 3], mu2 = sh[(k * 64 + c) * 3 + 1], M2 = sh[(k * 64 + c) * 3 + 2];
      const float nt = n + n2;
      if (nt > 0.f) {
        const float d = mu2 - mu;
        M = M + M2 + d * d * (n * n2 / nt);
        mu = mu + d * (n2 / nt);
      }
      n = nt;
    }
    slab[((size_t)e * 64 + c) * 2] = mu;
    slab[((size_t)e * 64 + c) * 2 + 1] = M;
    if (c == 0) slab_cnt[e] = n;
  }
}

extern "C" int tsr_cb16_stats_entries(int B, int HW) { return B * ((HW + 63) / 64); }

extern "C" int tsr_cb16_stats(const float* z, int z_ctot, int z_coff, int B, int HW, float* slab, float* slab_cnt,
                              void* stream) {
  if (!z || !slab || !slab_cnt || (z_ctot & 15) || (z_coff & 15) || z_coff + 64 > z_ctot || B <= 0) return TSR_ERR_ARG;
  const int chunks = (HW + 63) / 64;
  hipLaunchKernelGGL(cb16_stats_kernel<false>, dim3(B * chunks), dim3(256), 0, (hipStream_t)stream, z, z_ctot, z_coff, HW,
                     chunks, slab, slab_cnt);
  return tsr_check_launch();
}

// z is a bf16 CB16 tensor (training with bf16 activation storage: the stem's stored pre-activation)
extern "C" int tsr_cb16_stats_b16(const void* z, int z_ctot, int z_coff, int B, int HW, float* slab, float* slab_cnt,
                                  void* stream) {
  if (!z || !slab || !slab_cnt || (z_ctot & 15) || (z_coff & 15) || z_coff + 64 > z_ctot || B <= 0) return TSR_ERR_ARG;
  const int chunks = (HW + 63) / 64;
  hipLaunchKernelGGL(cb16_stats_kernel<true>, dim3(B * chunks), dim3(256), 0, (hipStream_t)stream, (const float*)z, z_ctot,
                     z_coff, HW, chunks, slab, slab_cnt);
  return tsr_check_launch();
}

// ------------------------------------------------------------------------------------------
// Per-sample PSNR / SSIM of eval_func (train/tactileSR_train.py:87-94; utility/tools.py:49-81).
//   PSNR = 10 log10(max^2 / (sum (a-b)^2 / psnr_div)),  psnr_div = shape[0]*shape[1] of the tensor the
//   reference passes ((1,H,W) in eval_func -> H: the reference's /40 quirk; (H,W) -> H*W).
//   SSIM = single global window with C1, C2.
// One workgroup per sample, fp64 accumulation.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void psnr_ssim_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        int n, double psnr_div, double maxv, double C1, double C2,
                                                        float* __restrict__ psnr, float* __restrict__ ssim) {
  __shared__ double sh[256 * 6];
  const float* pa = a + (size_t)blockIdx.x * n;
  const float* pb = b + (size_t)blockIdx.x * n;
  double s[6] = {0, 0, 0, 0, 0, 0};
  for (int i = threadIdx.x; i < n; i += 256) {
    const double x = pa[i], y = pb[i];
    s[0] += (x - y) * (x - y); s[1] += x; s[2] += y; s[3] += x * x; s[4] += y * y; s[5] += x * y;
  }
  for (int k = 0; k < 6; ++k) sh[threadIdx.x * 6 + k] = s[k];
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st)
      for (int k = 0; k < 6; ++k) sh[threadIdx.x * 6 + k] += sh[(threadIdx.x + st) * 6 + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double mse = sh[0] / psnr_div;
    psnr[blockIdx.x] = (float)(10.0 * log10(maxv * maxv / mse));
    const double mu1 = sh[1] / n, mu2 = sh[2] / n;
    const double s1 = sh[3] / n - mu1 * mu1, s2 = sh[4] / n - mu2 * mu2, s12 = sh[5] / n - mu1 * mu2;
    ssim[blockIdx.x] = (float)(((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2)));
  }
}

extern "C" int tsr_psnr_ssim(const float* a, const float* b, int B, int n, double psnr_div, double max_value,
                             double C1, double C2, float* psnr, float* ssim, void* stream) {
  if (!a || !b || !psnr || !ssim || B <= 0 || n <= 0 || psnr_div <= 0) return TSR_ERR_ARG;
  hipLaunchKernelGGL(psnr_ssim_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, a, b, n, psnr_div, max_value, C1,
                     C2, psnr, ssim);
  return tsr_check_launch();
}

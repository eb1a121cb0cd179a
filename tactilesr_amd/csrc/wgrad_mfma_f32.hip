// Weight gradient of the 64/128-channel convolutions on the fp32 matrix cores.
//
// Replaces the wgrad half of torch's convolution_backward for every conv that
// tsr_conv2d_* runs forward (the 38 nn.Conv2d of TactileSR minus the 3->64 stems and the
// 128->1 head; reference cpu/trainer.py:352-353 -> autograd of model/tactileSR_model.py).
//
//   dW[co][ci][kh][kw] = sum_{b,y,x} dz[b,co,y,x] * a[b,ci,y+kh-P,x+kw-P]
//
// GEMM view: M = co, N = ci, K = pixels (B*H*W), one GEMM per tap.  A workgroup owns a
// (64 co) x (64 ci) x (one kernel row kh) slice of dW -- KS accumulator tiles per wave, wave
// (wc,wi) = 32-co x 32-ci sub-block -- and sweeps the 8x8 patches of every image of its batch
// split: per patch the 64x64 dz tile and the 8 x (8+KS-1) x 64 input tile (with the
// producer's train-mode BN+ReLU applied on load, zero padded) are staged in LDS pixel-major,
// so both MFMA operands are conflict-free ds_read_b32 rows (channel on the lane, pixel = k).
// The dz fragment is shared by the KS taps of the row.  Partial dW per batch split goes to a
// slab and is summed in fp64 by tsr_reduce_splits (deterministic; no float atomics).
#include "tsr_common.h"

struct WgradArgs {
  const float* a;  int a_ctot; int a_coff; int cin;
  const float* a_scale; const float* a_shift;
  const float* dz; int dz_ctot; int dz_coff; int cout;
  float* slab;      // [nsplit][cout][cin][KS*KS]
  float* bslab;     // [nsplit][cout] or NULL
  int B, H, W, nsplit;
  int tiles_x, tiles_y;
};

template <int KS>
__global__ __launch_bounds__(256, 3) void wgrad_mfma_f32_kernel(const WgradArgs g) {
  constexpr int P = KS / 2;
  constexpr int HWD = 8 + KS - 1;   // staged input columns
  constexpr int DS = 68;            // floats per pixel row (64 ch + 4 pad)
  constexpr int DZ_F = 64 * DS;
  constexpr int A_F = 8 * HWD * DS;
  constexpr int NDZ = 64 * 16;      // float4 items of the dz tile
  constexpr int NA = 8 * HWD * 16;  // float4 items of the input tile
  constexpr int NITD = NDZ / 256;
  constexpr int NITA = (NA + 255) / 256;
  __shared__ __attribute__((aligned(16))) float lds[DZ_F + A_F];
  float* dzt = lds;
  float* at = lds + DZ_F;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wc = wave >> 1, wi = wave & 1;
  const int h = lane >> 5, li = lane & 31;

  const int nci = g.cin >> 6;
  int bid = blockIdx.x;
  const int cib = bid % nci; bid /= nci;
  const int nco = g.cout >> 6;
  const int cob = bid % nco; bid /= nco;
  const int kh = bid % KS;
  const int sp = bid / KS;

  const int HW = g.H * g.W;
  const int a_blocks = g.a_ctot >> 4, dz_blocks = g.dz_ctot >> 4;
  const int a_c0 = g.a_coff + cib * 64, dz_c0 = g.dz_coff + cob * 64;

  f32x16 acc[KS];
#pragma unroll
  for (int k = 0; k < KS; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  float bsum = 0.f;
  const bool do_bias = g.bslab && cib == 0 && kh == 0 && wi == 0;

  // work items = (image, patch), split evenly (+-1 patch) over the nsplit batch splits
  const int tpi = g.tiles_x * g.tiles_y;
  const long total_items = (long)g.B * tpi;
  const long per = (total_items + g.nsplit - 1) / g.nsplit;
  const long it0 = (long)sp * per;
  const long it1 = it0 + per < total_items ? it0 + per : total_items;
  for (long item = it0; item < it1; ++item) {
    {
      {
        const int b = (int)(item / tpi);
        const int trem = (int)(item - (long)b * tpi);
        const int ty = trem / g.tiles_x, tx = trem - ty * g.tiles_x;
        const int y0 = ty * 8, x0 = tx * 8;
        f32x4 dv[NITD], av[NITA];
#pragma unroll
        for (int k = 0; k < NITD; ++k) {
          const int it = tid + k * 256;
          const int q = it & 3, px = (it >> 2) & 63, blk = it >> 8;
          const int gy = y0 + (px >> 3), gx = x0 + (px & 7);
          dv[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (gy < g.H && gx < g.W)
            dv[k] = *(const f32x4*)(g.dz + (((size_t)b * dz_blocks + (dz_c0 >> 4) + blk) * HW + gy * g.W + gx) * 16 + q * 4);
        }
#pragma unroll
        for (int k = 0; k < NITA; ++k) {
          const int it = tid + k * 256;
          av[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (it < NA) {
            const int q = it & 3, px = (it >> 2) % (8 * HWD), blk = (it >> 2) / (8 * HWD);
            const int ry = px / HWD, rx = px - ry * HWD;
            const int gy = y0 + ry + kh - P, gx = x0 + rx - P;
            if (gy >= 0 && gy < g.H && gx >= 0 && gx < g.W) {
              f32x4 v = *(const f32x4*)(g.a + (((size_t)b * a_blocks + (a_c0 >> 4) + blk) * HW + gy * g.W + gx) * 16 + q * 4);
              if (g.a_scale) {
                const int cq = cib * 64 + blk * 16 + q * 4;
                const f32x4 sc = *(const f32x4*)(g.a_scale + cq);
                const f32x4 sh = *(const f32x4*)(g.a_shift + cq);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = tsr_relu(fmaf(v[j], sc[j], sh[j]));
              }
              av[k] = v;
            }
          }
        }
        __syncthreads();   // previous patch's readers are done
#pragma unroll
        for (int k = 0; k < NITD; ++k) {
          const int it = tid + k * 256;
          const int q = it & 3, px = (it >> 2) & 63, blk = it >> 8;
          *(f32x4*)(dzt + px * DS + blk * 16 + q * 4) = dv[k];
        }
#pragma unroll
        for (int k = 0; k < NITA; ++k) {
          const int it = tid + k * 256;
          if (it < NA) {
            const int q = it & 3, px = (it >> 2) % (8 * HWD), blk = (it >> 2) / (8 * HWD);
            *(f32x4*)(at + px * DS + blk * 16 + q * 4) = av[k];
          }
        }
        __syncthreads();
        const float* ap = dzt + h * DS + wc * 32 + li;
        const float* bp = at + h * DS + wi * 32 + li;
#pragma unroll 2
        for (int y = 0; y < 8; ++y) {
#pragma unroll
          for (int xp = 0; xp < 4; ++xp) {
            const float av1 = ap[(y * 8 + 2 * xp) * DS];
            if (do_bias) bsum += av1;
#pragma unroll
            for (int kw = 0; kw < KS; ++kw) {
              const float bv1 = bp[(y * HWD + 2 * xp + kw) * DS];
              acc[kw] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, bv1, acc[kw], 0, 0, 0);
            }
          }
        }
      }
    }
  }

  // partial dW of this split: OIHW
  const int T = KS * KS;
  float* sl = g.slab + (size_t)sp * g.cout * g.cin * T;
  const int ci = cib * 64 + wi * 32 + li;
#pragma unroll
  for (int kw = 0; kw < KS; ++kw) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cob * 64 + wc * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      sl[((size_t)co * g.cin + ci) * T + kh * KS + kw] = acc[kw][r];
    }
  }
  if (do_bias) {
    bsum += __shfl_xor(bsum, 32);
    if (h == 0) g.bslab[(size_t)sp * g.cout + cob * 64 + wc * 32 + li] = bsum;
  }
}

// out[i] = sum_s slab[s][i]  (fp64 accumulation, deterministic)
// out[i] = alpha * sum_k slab[k][i], fp64, fixed order (deterministic).  A workgroup owns 32 outputs; its 8 groups of
// 32 lanes each walk every 8th split (two independent chains per lane) and meet in LDS -- a lane per output with a
// serial walk over up to 340 splits leaves small layers with ~100 latency-bound workgroups.
__global__ __launch_bounds__(256) void reduce_splits_kernel(const float* __restrict__ slab, float* __restrict__ out,
                                                            size_t n, int nsplit, float alpha) {
  __shared__ double part[8][32];
  const int il = threadIdx.x & 31, kp = threadIdx.x >> 5;
  for (size_t i0 = (size_t)blockIdx.x * 32; i0 < n; i0 += (size_t)gridDim.x * 32) {
    const size_t i = i0 + il;
    double s0 = 0.0, s1 = 0.0;
    if (i < n) {
      int k = kp;
      for (; k + 8 < nsplit; k += 16) {
        s0 += (double)slab[(size_t)k * n + i];
        s1 += (double)slab[(size_t)(k + 8) * n + i];
      }
      if (k < nsplit) s0 += (double)slab[(size_t)k * n + i];
    }
    __syncthreads();
    part[kp][il] = s0 + s1;
    __syncthreads();
    if (kp == 0 && i < n) {
      double s = part[0][il];
#pragma unroll
      for (int q = 1; q < 8; ++q) s += part[q][il];
      out[i] = (float)(s * alpha);
    }
  }
}

extern "C" int tsr_reduce_splits(const float* slab, float* out, long long n, int nsplit, float alpha, void* stream) {
  if (!slab || !out || n <= 0 || nsplit <= 0) return TSR_ERR_ARG;
  const size_t g = ((size_t)n + 31) / 32;
  hipLaunchKernelGGL(reduce_splits_kernel, dim3(g > 65535 ? 65535 : (int)g), dim3(256), 0, (hipStream_t)stream, slab,
                     out, (size_t)n, nsplit, alpha);
  return tsr_check_launch();
}

extern "C" int tsr_conv2d_wgrad(const float* a, int a_ctot, int a_coff, int cin,
                                const float* a_scale, const float* a_shift,
                                const float* dz, int dz_ctot, int dz_coff, int cout, int ks,
                                float* slab, float* bias_slab, int nsplit,
                                int B, int H, int W, void* stream) {
  if (!a || !dz || !slab || B <= 0 || H <= 0 || W <= 0 || nsplit <= 0) return TSR_ERR_ARG;
  if ((cin & 63) || (cout & 63) || (a_ctot & 15) || (a_coff & 15) || (dz_ctot & 15) || (dz_coff & 15) ||
      a_coff + cin > a_ctot || dz_coff + cout > dz_ctot || (ks != 1 && ks != 3 && ks != 5))
    return TSR_ERR_ARG;
  if ((a_scale != nullptr) != (a_shift != nullptr)) return TSR_ERR_ARG;
  WgradArgs g;
  g.a = a; g.a_ctot = a_ctot; g.a_coff = a_coff; g.cin = cin; g.a_scale = a_scale; g.a_shift = a_shift;
  g.dz = dz; g.dz_ctot = dz_ctot; g.dz_coff = dz_coff; g.cout = cout;
  g.slab = slab; g.bslab = bias_slab; g.B = B; g.H = H; g.W = W; g.nsplit = nsplit;
  g.tiles_x = (W + 7) / 8; g.tiles_y = (H + 7) / 8;
  const int grid = nsplit * ks * (cout >> 6) * (cin >> 6);
  hipStream_t st = (hipStream_t)stream;
  if (ks == 1) hipLaunchKernelGGL((wgrad_mfma_f32_kernel<1>), dim3(grid), dim3(256), 0, st, g);
  else if (ks == 3) hipLaunchKernelGGL((wgrad_mfma_f32_kernel<3>), dim3(grid), dim3(256), 0, st, g);
  else hipLaunchKernelGGL((wgrad_mfma_f32_kernel<5>), dim3(grid), dim3(256), 0, st, g);
  return tsr_check_launch();
}

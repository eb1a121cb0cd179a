// tPSFNet forward model on the CDNA4 matrix cores (reference model/tPSFNet.py:78-141).
//
// The 99x99 PSF is separable, psf[u][v] = a g(u) g(v), and its support (+-49) covers the whole 100x100 image,
// so the "same" convolution of one sample is two dense 100x100x100 matrix products with a symmetric Toeplitz
// matrix G[i][j] = g(49 + i - j) (0 for |i-j| > 49):
//      HR_raw = a * G * D * G                       (:85-89)
// Both GEMMs run on v_mfma_f32_32x32x16_f16 with the operands split into two power-of-two-scaled fp16 planes
// (x*s = h1+h2, three products h2g1 + h1g2 + h1g1, fp32 accumulate: 22+ significand bits, the same arithmetic as
// the conv kernels' "fp16x3" form; scales from max|D| and a closed-form bound of sum g).
//   GEMM1  R = D*G   A = D rows straight from global memory into registers (each wave owns 32 image rows),
//                    B = Toeplitz fragments read from 8 shifted copies of the 1-D table (aligned ds_read_b128)
//   GEMM2  HR = G*R  A = the same table, B = R^T written to LDS as fp16 planes from GEMM1's accumulators
// then, still from registers: plateau fill (:86,95-97), HR store, separable Gaussian-masked pooling (:129-141).
// One workgroup (4 waves) per sample, persistent over the batch; 52 KB LDS -> 3 workgroups per CU.
// Per sample the kernel moves 40 KB depth in + 40 KB HR + 39 KB psf out: the roofline is HBM.
#include "tsr_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#define PS 99
#define HS 100
#define NPIX (HS * HS)
#define KP (100.0f / 4802.0f)      // 10^2 / (49^2 + 49^2)
#define KM (100.0f / 15138.0f)     // 10^2 / (87^2 + 87^2)

namespace {
constexpr int TC_STRIDE = 544;              // bytes per shifted copy of the Toeplitz table (272 halves; 34 chunks)
constexpr int T_PLANE = 8 * TC_STRIDE;      // 8 copies, copy r holds T[m + r]
constexpr int T_OFF = 128;                  // T[i] = g(49 + i - T_OFF)
constexpr int RT_STRIDE = 208;              // bytes per row of R^T (104 halves; 13 chunks: conflict-free b128 reads)
constexpr int RT_PLANE = HS * RT_STRIDE;
constexpr float G_SCALE = 8192.0f;          // g <= 1 -> [.., 2^13]

// power-of-two scale that brings m into [2^13, 2^14) (1 for m == 0)
__device__ __forceinline__ float pow2_scale_to_2p13(float m) {
  if (!(m > 0.f)) return 1.f;
  int e = (int)((__float_as_uint(m) >> 23) & 0xFF) - 127;
  int be = 13 - e + 127;
  be = be < 1 ? 1 : (be > 254 ? 254 : be);
  return __uint_as_float((unsigned)be << 23);
}

__device__ __forceinline__ f32x16 mfma16(f16x8 a, f16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
}  // namespace

__global__ __launch_bounds__(256, 2) void tpsf_fwd_mfma_kernel(const float* __restrict__ depth,
                                                               const float* __restrict__ ab,   // (B,3)
                                                               float* __restrict__ HR, float* __restrict__ LRd,
                                                               float* __restrict__ psf, int B) {
  __shared__ __attribute__((aligned(16))) char Tc[2 * T_PLANE];       // Toeplitz table copies; later pooling partials
  __shared__ __attribute__((aligned(16))) char RT[2 * RT_PLANE + 16]; // R^T planes (+ finite tail for the K over-read)
  __shared__ float g[128];        // g[0..98]
  __shared__ float ea[400];       // [4][100] mask factors
  __shared__ __attribute__((aligned(16))) unsigned char pmask[128 * 16];   // plateau bit per pixel: row y, bit x
  __shared__ float red[16];
  static_assert(2 * T_PLANE >= 4 * 4 * 128 * 4, "pooling partials alias the table");

  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, li = lane & 31;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave index, uniform
  if (tid < 4) ((float*)(RT + 2 * RT_PLANE))[tid] = 0.f;
  for (int i = tid; i < 128 * 4; i += 256) ((unsigned*)pmask)[i] = 0u;    // bytes x >= 104 are never written again
  const int y = 32 * w + li;                       // image row of this lane in the A (and GEMM2 A) layout
  const int q8 = (li + 7) & ~7, rcopy = q8 - li;   // table copy whose 16-B chunks line up with this lane's diagonal
  const int laneT = rcopy * TC_STRIDE + 2 * (T_OFF + 8 * h - q8);

  // depth rows -> registers (A layout: row y, k = 16 ks + 8 h + 0..7); the next sample's rows are fetched under
  // the current sample's epilogue
  f32x4 dA[7][2];
  auto load_depth = [&](int b) {
    const float* dp = depth + (size_t)b * NPIX + y * HS + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 7; ++ks)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        dA[ks][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (y < HS && 16 * ks + 8 * h + 4 * q < HS) dA[ks][q] = *(const f32x4*)(dp + 16 * ks + 4 * q);
      }
  };
  if ((int)blockIdx.x < B) load_depth(blockIdx.x);

  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const float alpha = ab[b * 3 + 0], beta = ab[b * 3 + 1], gamma = ab[b * 3 + 2];
    const float cpsf = KP / (beta * beta);
    const float cm = KM / gamma;

    float mabs = 0.f, mx = -INFINITY;
#pragma unroll
    for (int ks = 0; ks < 7; ++ks)
#pragma unroll
      for (int q = 0; q < 2; ++q)
        if (y < HS && 16 * ks + 8 * h + 4 * q < HS) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            mabs = fmaxf(mabs, fabsf(dA[ks][q][j]));
            mx = fmaxf(mx, dA[ks][q][j]);
          }
        }
    for (int i = tid; i < PS; i += 256) { const float t = (float)(i - 49); g[i] = expf(-cpsf * t * t); }
    for (int i = tid; i < 400; i += 256) {
      const int a = i / 100, x = i - a * 100;
      const float t = (float)(x - (12 + 25 * a));
      ea[i] = expf(-cm * t * t);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mabs = fmaxf(mabs, __shfl_xor(mabs, o));
      mx = fmaxf(mx, __shfl_xor(mx, o));
    }
    if (lane == 0) { red[w] = mabs; red[4 + w] = mx; }
    __syncthreads();                                                      // (1) g, ea, red
    mabs = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float dmax = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));

    // ---- Toeplitz table, 8 shifted copies x 2 planes
    for (int i = tid; i < 8 * (TC_STRIDE / 2); i += 256) {
      const int r = i / (TC_STRIDE / 2), m = i - r * (TC_STRIDE / 2);
      const int d = m + r - T_OFF;
      const float v = (d >= -49 && d <= 49) ? g[49 + d] * G_SCALE : 0.f;
      const _Float16 hi = (_Float16)v;
      *(_Float16*)(Tc + r * TC_STRIDE + 2 * m) = hi;
      *(_Float16*)(Tc + T_PLANE + r * TC_STRIDE + 2 * m) = (_Float16)(v - (float)hi);
    }
    // ---- psf output (B,1,99,99)
    float* pp = psf + (size_t)b * PS * PS;
    {
      int u = tid / PS, v = tid - u * PS;          // i = tid + 256 n: (u, v) advance by (2, 58) with carry
      for (int i = tid; i < PS * PS; i += 256) {
        pp[i] = alpha * (g[u] * g[v]);
        v += 256 - 2 * PS;
        u += 2;
        if (v >= PS) { v -= PS; ++u; }
      }
    }

    // ---- operand scales; split the depth rows into fp16 planes
    const float sD = pow2_scale_to_2p13(mabs);
    const float gbound = fminf(99.0f, 1.0f + sqrtf(3.14159265f / cpsf));   // sum_d g(d) <= 1 + sqrt(pi / c)
    const float sR = pow2_scale_to_2p13(mabs * gbound);
    const float thr = dmax - 1e-3f;               // plateau: depth > depth.max() - 1e-3
    f16x8 a_hi[7], a_lo[7];
#pragma unroll
    for (int ks = 0; ks < 7; ++ks) {
      unsigned bits = 0u;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = dA[ks][j >> 2][j & 3];
        bits |= (d > thr ? 1u : 0u) << j;
        const float t = d * sD;
        const _Float16 hi = (_Float16)t;
        a_hi[ks][j] = hi;
        a_lo[ks][j] = (_Float16)(t - (float)hi);
      }
      if (16 * ks + 8 * h < 104) pmask[y * 16 + 2 * ks + h] = (unsigned char)bits;   // rows / columns >= 100: depth 0
    }
    __syncthreads();                                                      // (2) table copies

    // ---- GEMM1: R = D * G
    f32x16 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    {
      // 28 (k step, column tile) steps of 3 MFMAs; the table fragments run two steps ahead in a 3-slot register
      // ring, and a scheduling barrier per step keeps hipcc from hoisting all 56 ds_reads to the top
      f16x8 bh[3], bl[3];
#define G1_LOAD(i)                                                                                   \
  {                                                                                                  \
    bh[(i) % 3] = *(const f16x8*)(Tc + laneT + 32 * ((i) / 4) - 64 * ((i) % 4));                    \
    bl[(i) % 3] = *(const f16x8*)(Tc + T_PLANE + laneT + 32 * ((i) / 4) - 64 * ((i) % 4));          \
  }
      G1_LOAD(0);
      G1_LOAD(1);
#pragma unroll
      for (int i = 0; i < 28; ++i) {
        if (i + 2 < 28) G1_LOAD(i + 2);
        const int ks = i / 4, nt = i % 4;
        acc[nt] = mfma16(a_lo[ks], bh[i % 3], acc[nt]);
        acc[nt] = mfma16(a_hi[ks], bl[i % 3], acc[nt]);
        acc[nt] = mfma16(a_hi[ks], bh[i % 3], acc[nt]);
        __builtin_amdgcn_sched_barrier(0);
      }
#undef G1_LOAD
    }
    // ---- R^T planes -> LDS (rows y' >= 100 of the accumulators are exact zeros: they pad K to 104)
    {
      const float f1 = sR / (sD * G_SCALE);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int x = 32 * nt + li;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int y0 = 32 * w + 8 * rq + 4 * h;
          f16x4 hi, lo;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float t = acc[nt][rq * 4 + j] * f1;
            hi[j] = (_Float16)t;
            lo[j] = (_Float16)(t - (float)hi[j]);
          }
          if (x < HS && y0 < 104) {
            *(f16x4*)(RT + x * RT_STRIDE + 2 * y0) = hi;
            *(f16x4*)(RT + RT_PLANE + x * RT_STRIDE + 2 * y0) = lo;
          }
        }
      }
    }
    __syncthreads();                                                      // (3) R^T

    // ---- GEMM2: HR = G * R   (A = table with nt -> w; K columns >= 104 of the last step are masked in A)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    {
      f16x8 gh[2], gl[2], rh[3], rl[3];
      const int rtl0 = (li < HS ? li : HS - 1) * RT_STRIDE + 16 * h;             // column tiles 0..2: x = 32 nt + li < 100
      const int rtl3 = (96 + li < HS ? 96 + li : HS - 1) * RT_STRIDE + 16 * h;   // tile 3: clamp x to 99
#define G2_LOADA(ks)                                                                                 \
  {                                                                                                  \
    gh[(ks) & 1] = *(const f16x8*)(Tc + laneT + 32 * (ks) - 64 * w);                                 \
    gl[(ks) & 1] = *(const f16x8*)(Tc + T_PLANE + laneT + 32 * (ks) - 64 * w);                       \
    if ((ks) == 6 && h == 1) {                                                                       \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) { gh[(ks) & 1][j] = (_Float16)0.f; gl[(ks) & 1][j] = (_Float16)0.f; } \
    }                                                                                                \
  }
#define G2_LOADB(i)                                                                                  \
  {                                                                                                  \
    const int off_ = ((i) % 4 == 3 ? rtl3 : rtl0 + ((i) % 4) * 32 * RT_STRIDE) + 32 * ((i) / 4);     \
    rh[(i) % 3] = *(const f16x8*)(RT + off_);                                                        \
    rl[(i) % 3] = *(const f16x8*)(RT + RT_PLANE + off_);                                             \
  }
      G2_LOADA(0);
      G2_LOADB(0);
      G2_LOADB(1);
#pragma unroll
      for (int i = 0; i < 28; ++i) {
        const int ks = i / 4, nt = i % 4;
        if (i + 2 < 28) G2_LOADB(i + 2);
        if (nt == 0 && ks + 1 < 7) G2_LOADA(ks + 1);
        acc[nt] = mfma16(gl[ks & 1], rh[i % 3], acc[nt]);
        acc[nt] = mfma16(gh[ks & 1], rl[i % 3], acc[nt]);
        acc[nt] = mfma16(gh[ks & 1], rh[i % 3], acc[nt]);
        __builtin_amdgcn_sched_barrier(0);
      }
#undef G2_LOADA
#undef G2_LOADB
    }

    if (b + (int)gridDim.x < B) load_depth(b + gridDim.x);   // the A planes are dead: their registers take the next rows

    // ---- plateau fill: HR[mask] = max(HR outside mask, 0), mask = depth > depth.max() - 1e-3
    // (lane constants re-derived from an opaque copy per sample: hipcc would otherwise hoist 64 loop-invariant
    //  64-bit addresses and predicates out of the persistent loop and spill them)
    int li_o = li, h_o = h;
    asm volatile("" : "+v"(li_o), "+v"(h_o));
    const int yb = 32 * w + 4 * h_o;               // accumulator row r -> image row yb + (r & 3) + 8 (r >> 2)
    const bool vx3 = li_o < HS - 96;               // column tile 3 holds x = 96..99 only
    const float f2 = alpha / (G_SCALE * sR);
    unsigned long long plateau = 0ull;
    float fmax_out = 0.f;          // tmp[mask] = 0 takes part in the max (the mask is never empty)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int yy = yb + (r & 3) + 8 * (r >> 2);
      const bool vy = yy < HS;
      const uint4 mrow = *(const uint4*)(pmask + yy * 16);
      const unsigned mw[4] = {mrow.x, mrow.y, mrow.z, mrow.w};
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const bool valid = vy && (nt < 3 || vx3);
        const bool pl = valid && ((mw[nt] >> li_o) & 1u);
        const float v = valid ? acc[nt][r] * f2 : 0.f;
        if (pl) plateau |= 1ull << (nt * 16 + r);
        else fmax_out = fmaxf(fmax_out, v);
        acc[nt][r] = v;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) fmax_out = fmaxf(fmax_out, __shfl_xor(fmax_out, o));
    if (lane == 0) red[8 + w] = fmax_out;
    __syncthreads();                                                      // (4) fill; every wave is past GEMM2
    const float fill = fmaxf(fmaxf(red[8], red[9]), fmaxf(red[10], red[11]));

    // ---- HR store + separable Gaussian-masked pooling partials  P[a][x] = sum_y ea[a][y] HR[y][x]
    float* hp = HR + (size_t)b * NPIX + li_o;
    float* Ppart = (float*)Tc;       // [4 waves][4 a][128 x]
    float s0 = 0.f;
    float pa[4][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int a = 0; a < 4; ++a) pa[nt][a] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int yy = yb + (r & 3) + 8 * (r >> 2);
      const bool vy = yy < HS;
      const int yc = vy ? yy : HS - 1;
      const float e[4] = {ea[yc], ea[100 + yc], ea[200 + yc], ea[300 + yc]};
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const float v = ((plateau >> (nt * 16 + r)) & 1ull) ? fill : acc[nt][r];   // 0 outside the image
        if (vy && (nt < 3 || vx3)) hp[yy * HS + 32 * nt] = v;
        s0 += v;
#pragma unroll
        for (int a = 0; a < 4; ++a) pa[nt][a] = fmaf(e[a], v, pa[nt][a]);
      }
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        pa[nt][a] += __shfl_xor(pa[nt][a], 32);
        if (h == 0) Ppart[(w * 4 + a) * 128 + 32 * nt + li] = pa[nt][a];
      }
    double s0d = (double)s0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s0d += __shfl_xor(s0d, o);
    if (lane == 0) red[12 + w] = (float)s0d;      // per-wave sums of <= 3200 values: fp32 holds them to 1e-7
    __syncthreads();                                                      // (5) pooling partials
    {
      const int pair = tid >> 4, sub = tid & 15, a = pair >> 2, c = pair & 3;
      float s = 0.f;
      for (int x = sub; x < HS; x += 16) {
        const float p = (Ppart[(0 * 4 + a) * 128 + x] + Ppart[(1 * 4 + a) * 128 + x]) +
                        (Ppart[(2 * 4 + a) * 128 + x] + Ppart[(3 * 4 + a) * 128 + x]);
        s = fmaf(p, ea[c * 100 + x], s);
      }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o);
      if (sub == 0) {
        const double S0 = ((double)red[12] + (double)red[13]) + ((double)red[14] + (double)red[15]);
        const float mn = expf(-100.0f / gamma);
        LRd[b * 16 + pair] = (s - mn * (float)S0) / (1.0f - mn) * 1e-4f;
      }
    }
    __syncthreads();                                                      // (6) LDS free for the next sample
  }
}

extern "C" int tpsf_forward(const float* depth, const float* alpha_beta, float* HR, float* LR_deg, float* psf,
                            int B, void* stream) {
  if (!depth || !alpha_beta || !HR || !LR_deg || !psf || B <= 0) return TSR_ERR_ARG;
  const int grid = B < 512 ? B : 512;          // 2 resident workgroups x 256 CUs, persistent over the batch
  hipLaunchKernelGGL(tpsf_fwd_mfma_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, depth, alpha_beta, HR,
                     LR_deg, psf, B);
  return tsr_check_launch();
}

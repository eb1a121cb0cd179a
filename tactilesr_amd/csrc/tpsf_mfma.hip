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
#include <type_traits>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#define PS 99
#define HS 100
#define NPIX (HS * HS)
#define KP (100.0f / 4802.0f)      // 10^2 / (49^2 + 49^2)
#define KM (100.0f / 15138.0f)     // 10^2 / (87^2 + 87^2)

namespace {
constexpr int TC_STRIDE = 480;              // bytes per shifted copy of the Toeplitz table (240 halves; 30 chunks)
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

// m = all ones -> a, 0 -> b; integer arithmetic only (the epilogues avoid bool selects: hipcc turns those into
// per-pixel branches whose live-through values it then spills)
__device__ __forceinline__ float bitsel(unsigned m, float a, float b) {
  return __uint_as_float((__float_as_uint(a) & m) | (__float_as_uint(b) & ~m));
}

__device__ __forceinline__ f32x16 mfma16(f16x8 a, f16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
}  // namespace

namespace {
// Toeplitz table of a symmetric 99-tap filter tab[0..98] (centre 49), times `scale`, as 8 shifted copies x 2 fp16
// planes: copy r holds T[m + r], T[i] = tab[49 + i - T_OFF] (0 outside the +-49 band)
__device__ __forceinline__ void build_toeplitz(char* Tc, const float* tab, float scale, int tid) {
  for (int i = tid; i < 8 * (TC_STRIDE / 2); i += 256) {
    const int r = i / (TC_STRIDE / 2), m = i - r * (TC_STRIDE / 2);
    const int d = m + r - T_OFF;
    const float v = (d >= -49 && d <= 49) ? tab[49 + d] * scale : 0.f;
    const _Float16 hi = (_Float16)v;
    *(_Float16*)(Tc + r * TC_STRIDE + 2 * m) = hi;
    *(_Float16*)(Tc + T_PLANE + r * TC_STRIDE + 2 * m) = (_Float16)(v - (float)hi);
  }
}

// acc[nt] += A(rows of this wave, fp16 planes in registers) * Toeplitz(Tc)   -- 28 (k step, column tile) steps of 3
// MFMAs; the table fragments run two steps ahead in a 3-slot register ring, and a scheduling barrier per step keeps
// hipcc from hoisting all 56 ds_reads to the top
__device__ __forceinline__ void gemm_rows_toeplitz(f32x16 (&acc)[4], const f16x8 (&a_hi)[7], const f16x8 (&a_lo)[7],
                                                   const char* Tc, int laneT) {
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

// R^T planes -> LDS: RT[x][y'] = acc * f1 split in two fp16 planes.  LW = the wave that owns rows 96..127: only its
// first row group exists (rows 96..99, and exact zeros for 100..103, which pad K to 104).  Column tile 3 holds
// x = 96..99 only: one predicated block for it instead of a predicate per write.
template <bool LW>
__device__ __forceinline__ void store_rt(const f32x16 (&acc)[4], float f1, char* RT, int w, int h, int li) {
  constexpr int NRQ = LW ? 1 : 4;
  auto put = [&](int nt) {
    const int x = 32 * nt + li;
#pragma unroll
    for (int rq = 0; rq < NRQ; ++rq) {
      const int y0 = 32 * w + 8 * rq + 4 * h;
      f16x4 hi, lo;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t = acc[nt][rq * 4 + j] * f1;
        hi[j] = (_Float16)t;
        lo[j] = (_Float16)(t - (float)hi[j]);
      }
      *(f16x4*)(RT + x * RT_STRIDE + 2 * y0) = hi;
      *(f16x4*)(RT + RT_PLANE + x * RT_STRIDE + 2 * y0) = lo;
    }
  };
  put(0);
  put(1);
  put(2);
  if (li < HS - 96) put(3);
}

// acc[nt] += Toeplitz(Tc)(rows of this wave) * R(RT)   (K columns >= 104 of the last step are masked in A)
__device__ __forceinline__ void gemm_toeplitz_rt(f32x16 (&acc)[4], const char* Tc, const char* RT, int laneT, int w,
                                                 int h, int li) {
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

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[4]) {
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
}
}  // namespace

#ifndef TPSF_FWD_OCC
#define TPSF_FWD_OCC 2      // resident workgroups per CU the forward kernel is compiled for
#endif
__global__ __launch_bounds__(256, TPSF_FWD_OCC) void tpsf_fwd_mfma_kernel(const float* __restrict__ depth,
                                                               const float* __restrict__ ab,   // (B,3)
                                                               float* __restrict__ HR, float* __restrict__ LRd,
                                                               float* __restrict__ psf, int B) {
  __shared__ __attribute__((aligned(16))) char Tc[2 * T_PLANE];       // Toeplitz table copies; later pooling partials
  __shared__ __attribute__((aligned(16))) char RT[2 * RT_PLANE + 16]; // R^T planes (+ finite tail for the K over-read)
  __shared__ float g[128];        // g[0..98]
  __shared__ float ea[400];       // [4][100] mask factors
  __shared__ __attribute__((aligned(16))) unsigned char pmask[HS * 16];   // plateau bit per pixel: row y, bit x
  __shared__ float red[16];
  static_assert(2 * T_PLANE >= 4 * 4 * HS * 4, "pooling partials alias the table");

  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, li = lane & 31;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave index, uniform
  if (tid < 4) ((float*)(RT + 2 * RT_PLANE))[tid] = 0.f;
  for (int i = tid; i < HS * 4; i += 256) ((unsigned*)pmask)[i] = 0u;    // bytes x >= 104 are never written again
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
#if TPSF_FWD_OCC == 2
  if ((int)blockIdx.x < B) load_depth(blockIdx.x);
#endif

  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const float alpha = ab[b * 3 + 0], beta = ab[b * 3 + 1], gamma = ab[b * 3 + 2];
    const float cpsf = KP / (beta * beta);
    const float cm = KM / gamma;
#if TPSF_FWD_OCC != 2
    load_depth(b);
#endif

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

    build_toeplitz(Tc, g, G_SCALE, tid);
    // ---- psf output (B,1,99,99)
    float* pp = psf + (size_t)b * PS * PS;
    if (tid < 2 * PS) {                              // two rows per pass: thread = (row parity, column v), g(v) in a register
      const int u0 = tid >= PS ? 1 : 0, v = tid - u0 * PS;
      const float gv = g[v];
      float* po = pp + u0 * PS + v;
      for (int u = u0; u < PS; u += 2, po += 2 * PS) *po = alpha * (g[u] * gv);      // exactly symmetric in (u, v)
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
      if (16 * ks + 8 * h < 104 && y < HS) pmask[y * 16 + 2 * ks + h] = (unsigned char)bits;
    }
    __syncthreads();                                                      // (2) table copies

    // ---- GEMM1: R = D * G;  R^T planes -> LDS;  GEMM2: HR = G * R
    f32x16 acc[4];
    zero_acc(acc);
    gemm_rows_toeplitz(acc, a_hi, a_lo, Tc, laneT);
    if (w < 3) store_rt<false>(acc, sR / (sD * G_SCALE), RT, w, h, li);
    else store_rt<true>(acc, sR / (sD * G_SCALE), RT, w, h, li);
    __syncthreads();                                                      // (3) R^T
    zero_acc(acc);
    gemm_toeplitz_rt(acc, Tc, RT, laneT, w, h, li);

#if TPSF_FWD_OCC == 2
    if (b + (int)gridDim.x < B) load_depth(b + gridDim.x);   // the A planes are dead: their registers take the next rows
#endif

    // ---- plateau fill: HR[mask] = max(HR outside mask, 0), mask = depth > depth.max() - 1e-3; then the HR store
    // and the separable Gaussian-masked pooling partials P[a][x] = sum_y ea[a][y] HR[y][x], all from the
    // accumulator registers.  Two compile-time forms: waves 0..2 own 32 full rows (no row masks at all), wave 3 owns
    // rows 96..99 only (accumulator rows r < 4 of the h = 0 half).  Column tile 3 holds x = 96..99: masked, and
    // stored under one predicate.  Lane constants are re-derived from an opaque copy per sample (hipcc would
    // otherwise hoist 64 loop-invariant addresses out of the persistent loop and spill them); integer masks instead
    // of bool selects (those become per-pixel branches with spilled live-through values).
    int li_o = li, h_o = h;
    asm volatile("" : "+v"(li_o), "+v"(h_o));
    const unsigned cm3 = 0u - ((unsigned)(li_o - (HS - 96)) >> 31);      // all ones where tile-3 column exists
    const float f2 = alpha / (G_SCALE * sR);
    float* hp = HR + (size_t)b * NPIX + li_o;
    float* Ppart = (float*)Tc;       // [4 waves][4 a][100 x]  (the table is dead after GEMM2; barrier (4) orders it)
    float s0 = 0.f;
    auto epilogue = [&](auto lastw) {
      constexpr bool LW = decltype(lastw)::value;
      constexpr int NR = LW ? 4 : 16;
      const int yb = 32 * w + 4 * h_o;               // accumulator row r -> image row yb + (r & 3) + 8 (r >> 2)
      const unsigned rowm = LW ? 0u - (unsigned)(h_o == 0) : ~0u;
      unsigned long long plateau = 0ull;
      float fmax_out = 0.f;          // tmp[mask] = 0 takes part in the max (the mask is never empty)
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int yy = yb + (r & 3) + 8 * (r >> 2);
        const uint4 mrow = *(const uint4*)(pmask + (LW ? (yy < HS - 1 ? yy : HS - 1) : yy) * 16);
        const unsigned mw[4] = {mrow.x, mrow.y, mrow.z, mrow.w};
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const unsigned vm = nt < 3 ? rowm : (rowm & cm3);
          const unsigned pli = (mw[nt] >> li_o) & 1u & vm;
          const float v = __uint_as_float(__float_as_uint(acc[nt][r] * f2) & vm);   // accumulators outside are finite
          plateau |= (unsigned long long)pli << (nt * 16 + r);
          fmax_out = fmaxf(fmax_out, bitsel(0u - pli, 0.f, v));
          acc[nt][r] = v;
        }
        asm volatile("" ::: "memory");     // keep the row's LDS reads and arithmetic in its iteration (hipcc
        __builtin_amdgcn_sched_barrier(0); // otherwise regroups the unrolled rows and spills hundreds of values)
      }
      {
        // opaque copy: hipcc otherwise keeps the per-pixel conditions alive as lane masks instead of these bits
        unsigned plo = (unsigned)plateau, phi = (unsigned)(plateau >> 32);
        asm volatile("" : "+v"(plo), "+v"(phi));
        plateau = ((unsigned long long)phi << 32) | plo;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) fmax_out = fmaxf(fmax_out, __shfl_xor(fmax_out, o));
      if (lane == 0) red[8 + w] = fmax_out;
      __syncthreads();                                                    // (4) fill; every wave is past GEMM2
      const float fill = fmaxf(fmaxf(red[8], red[9]), fmaxf(red[10], red[11]));

      float pa[4][4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int a = 0; a < 4; ++a) pa[nt][a] = 0.f;
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int yy = yb + (r & 3) + 8 * (r >> 2);
        const int yc = LW ? (yy < HS - 1 ? yy : HS - 1) : yy;
        const float e[4] = {ea[yc], ea[100 + yc], ea[200 + yc], ea[300 + yc]};
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const unsigned pm = 0u - (unsigned)((plateau >> (nt * 16 + r)) & 1ull);
          const float v = bitsel(pm, fill, acc[nt][r]);             // 0 outside the image
          acc[nt][r] = v;
          if (nt < 3) {
            if (!LW) hp[yy * HS + 32 * nt] = v;
            else if (h_o == 0) hp[yy * HS + 32 * nt] = v;
          }
          s0 += v;
#pragma unroll
          for (int a = 0; a < 4; ++a) pa[nt][a] = fmaf(e[a], v, pa[nt][a]);
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
      if (cm3 & rowm) {                    // x = 96..99 of this wave's rows
#pragma unroll
        for (int r = 0; r < NR; ++r) hp[(yb + (r & 3) + 8 * (r >> 2)) * HS + 96] = acc[3][r];
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          pa[nt][a] += __shfl_xor(pa[nt][a], 32);
          if (h == 0 && 32 * nt + li < HS) Ppart[(w * 4 + a) * HS + 32 * nt + li] = pa[nt][a];
        }
    };
    if (w < 3) epilogue(std::false_type());
    else epilogue(std::true_type());
    double s0d = (double)s0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s0d += __shfl_xor(s0d, o);
    if (lane == 0) red[12 + w] = (float)s0d;      // per-wave sums of <= 3200 values: fp32 holds them to 1e-7
    __syncthreads();                                                      // (5) pooling partials
    {
      const int pair = tid >> 4, sub = tid & 15, a = pair >> 2, c = pair & 3;
      float s = 0.f;
      for (int x = sub; x < HS; x += 16) {
        const float p = (Ppart[(0 * 4 + a) * HS + x] + Ppart[(1 * 4 + a) * HS + x]) +
                        (Ppart[(2 * 4 + a) * HS + x] + Ppart[(3 * 4 + a) * HS + x]);
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

// Backward of the forward model w.r.t. (alpha, beta, gamma), given dL/dLRd (B,16) -- autograd of reference
// model/tPSFNet.py:78-141 as train/tPSFNet_train.py:180-190 drives it -- in TWO kernels (round 3; the single
// persistent kernel of rounds 1-2 recomputed hr = a G D G, kept two accumulator sets and ~70 epilogue values live, needed
// the 512-register budget, still spilled, and ran one wave per SIMD with every latency exposed: 1.28 ms per 8192
// samples against 0.18 ms of MFMA time):
//
//  1. tpsf_bwd_pool_kernel -- everything that is a reduction over the STORED forward output HR (40 KB per sample, read
//     once, HBM-bound): d/d(gamma) through the Gaussian pooling masks, d/d(alpha) = sum over non-plateau pixels of
//     wgt * hr / alpha, and the per-pixel weight wgt(y,x) = dL/dHR off the plateau (0 on it: the plateau value is a
//     detached constant, :95-97), written to a work buffer for kernel 2.
//  2. tpsf_bwd_dhb_kernel -- d/d(beta): with H the Toeplitz matrix of h(t) = (t-49)^2 g(t),
//         -(d/dc)(G D G) = H D G + G D H,   c = Kp / b^2,
//     four Toeplitz GEMMs on the matrix cores (Rg = D G -> LDS, H Rg; Rh = D H -> LDS, G Rh), each product reduced against
//     wgt straight from its accumulator (the sum is linear in dhb, so the two terms never have to be added
//     element-wise: ONE accumulator set is live at a time).  Round 4: ONE workgroup per CU (512-register budget): the 64
//     weights a lane multiplies its accumulators with, and the depth rows of the next row GEMM, are requested a whole
//     column GEMM ahead of their use and held in registers unspilled -- loaded row by row inside the reduction each of
//     the 2 x 16 weight rows per sample paid a full memory round trip (waves parked 71 % of their cycles): 0.76 -> 0.51 ms
//     per 8192 samples (the kernel had measured the same at one and at two workgroups per CU before).
__global__ __launch_bounds__(256) void tpsf_bwd_pool_kernel(const float* __restrict__ depth, const float* __restrict__ ab,
                                                            const float* __restrict__ HR, const float* __restrict__ dLRd,
                                                            float* __restrict__ dab, float* __restrict__ wgt, int B) {
  __shared__ float ea[400], ea2[400], qd[400];       // [4][100]: mask factors, (x-cx)^2 times them, sum_c dl[a][c] ea_c(x)
  __shared__ float dl[16];
  __shared__ float redf[8];
  __shared__ double redd[4 * 34];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const float alpha = ab[b * 3 + 0], gamma = ab[b * 3 + 2];
    const float cm = KM / gamma;
    const float mn = expf(-100.0f / gamma);
    const float k0 = 1e-4f / (1.0f - mn);
    const float* dp = depth + (size_t)b * NPIX;
    const float* hp = HR + (size_t)b * NPIX;
    float mx = -INFINITY;                // pass 1: the plateau threshold (the rows are re-read from L2 in pass 2)
    {
      const f32x4* dp4 = (const f32x4*)dp;             // 40,000 B per sample: 16-B aligned; 2,500 quads, 10 per thread
#pragma unroll
      for (int k = 0; k < 10; ++k) {
        const int q = tid + 256 * k;
        const f32x4 v = dp4[q < NPIX / 4 ? q : 0];
        const float m4 = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
        mx = fmaxf(mx, q < NPIX / 4 ? m4 : -INFINITY);
      }
    }
    for (int i = tid; i < 400; i += 256) {
      const int a = i / 100, x = i - a * 100;
      const float t = (float)(x - (12 + 25 * a));
      const float e = expf(-cm * t * t);
      ea[i] = e;
      ea2[i] = t * t * e;
    }
    if (tid < 16) dl[tid] = dLRd[b * 16 + tid];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) redf[w] = mx;
    __syncthreads();                                                       // (1) ea, dl, max partials
    const float thr = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3])) - 1e-3f;
    for (int i = tid; i < 400; i += 256) {
      const int a = i / 100, x = i - a * 100;
      qd[i] = dl[a * 4 + 0] * ea[x] + dl[a * 4 + 1] * ea[100 + x] + dl[a * 4 + 2] * ea[200 + x] + dl[a * 4 + 3] * ea[300 + x];
    }
    float dlsum = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) dlsum += dl[t];
    __syncthreads();                                                       // (2) qd
    float S[4][4], Sd[4][4], s0 = 0.f, da = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) { S[a][c] = 0.f; Sd[a][c] = 0.f; }
    float* wp = wgt + (size_t)b * NPIX;
#pragma unroll 2
    for (int k = 0; k < 40; ++k) {
      const int p = tid + 256 * k;
      if (p < NPIX) {
        const int y = (int)(((unsigned)p * 5243u) >> 19), x = p - y * HS;      // p / 100 for p < 43,698
        const float v = hp[p];                                   // stored HR: the plateau already holds its fill value
        const float ey[4] = {ea[y], ea[100 + y], ea[200 + y], ea[300 + y]};
        const float ey2[4] = {ea2[y], ea2[100 + y], ea2[200 + y], ea2[300 + y]};
        const float ex[4] = {ea[x], ea[100 + x], ea[200 + x], ea[300 + x]};
        const float ex2[4] = {ea2[x], ea2[100 + x], ea2[200 + x], ea2[300 + x]};
        const float gsum = ey[0] * qd[x] + ey[1] * qd[100 + x] + ey[2] * qd[200 + x] + ey[3] * qd[300 + x];
        const float wv = dp[p] > thr ? 0.f : k0 * (gsum - mn * dlsum);
        wp[p] = wv;
        s0 += v;
        da = fmaf(wv, v, da);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const float t = v * ey[a], t2 = v * ey2[a];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            S[a][c] = fmaf(t, ex[c], S[a][c]);
            Sd[a][c] = fmaf(t2, ex[c], fmaf(t, ex2[c], Sd[a][c]));
          }
        }
      }
    }
    // ---- block reduction of the 34 partial sums in double (per-thread fp32 sums hold <= 40 terms)
    double vals[34];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) { vals[a * 4 + c] = (double)S[a][c]; vals[16 + a * 4 + c] = (double)Sd[a][c]; }
    vals[32] = (double)s0;
    vals[33] = (double)da;
    // wave sums.  The 32 mask sums go through a reduce-SCATTER butterfly: at offset 32 a lane sends the half of its values
    // its partner keeps and adds the half it receives, and so on down to offset 2 -- 16 + 8 + 4 + 2 + 1 exchanges -- after
    // which lane l holds value (l >> 1) summed over half the wave, and one more exchange finishes it: 32 double shuffles
    // instead of 32 x 6 (fixed order: deterministic).  s0 and da take the plain 6-step reduction.
#pragma unroll
    for (int half = 16, o = 32; half >= 1; half >>= 1, o >>= 1) {
      const bool up = (lane & o) != 0;
#pragma unroll
      for (int j = 0; j < half; ++j) {
        const double send = up ? vals[j] : vals[j + half];
        const double keep = up ? vals[j + half] : vals[j];
        vals[j] = keep + __shfl_xor(send, o);
      }
    }
    vals[0] += __shfl_xor(vals[0], 1);
    if ((lane & 1) == 0) redd[w * 34 + (lane >> 1)] = vals[0];
#pragma unroll
    for (int i = 32; i < 34; ++i) {
      double v = vals[i];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0) redd[w * 34 + i] = v;
    }
    __syncthreads();                                                       // (3) wave sums
    if (tid < 16) {
      const int pair = tid;
      const double Sv = (redd[pair] + redd[34 + pair]) + (redd[68 + pair] + redd[102 + pair]);
      const double Sdv = (redd[16 + pair] + redd[34 + 16 + pair]) + (redd[68 + 16 + pair] + redd[102 + 16 + pair]);
      const double S0 = (redd[32] + redd[34 + 32]) + (redd[68 + 32] + redd[102 + 32]);
      const double gm = gamma, mnd = mn;
      const double mnp = mnd * 100.0 / (gm * gm);                 // d mn / d gamma
      const double dS = Sdv * (double)KM / (gm * gm);             // d S_ac / d gamma
      const double num = Sv - mnd * S0, den = 1.0 - mnd;
      const double dLR = 1e-4 * ((dS - mnp * S0) * den + num * mnp) / (den * den);
      double dg = (double)dl[pair] * dLR;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) dg += __shfl_xor(dg, o);
      if (pair == 0) {
        const double DA = (redd[33] + redd[34 + 33]) + (redd[68 + 33] + redd[102 + 33]);
        dab[b * 3 + 0] = (float)(DA / (double)alpha);
        dab[b * 3 + 2] = (float)dg;
      }
    }
    __syncthreads();                                                       // (4) LDS free for the next sample
  }
}

#ifndef TPSF_BWD_OCC
#define TPSF_BWD_OCC 1      // one workgroup per CU: the 512-register budget holds the prefetched weights + depth rows unspilled
#endif
// (no __restrict__ on the inputs: hipcc treats loads through restrict-const pointers as invariant, hoists the 128 weight
// loads of the two reductions and the depth rows of the second row GEMM to the top of the sample iteration -- across the
// barriers -- and spills ~250 registers to keep them; possibly-aliasing pointers keep every load where it is written)
__global__ __launch_bounds__(256, TPSF_BWD_OCC) void tpsf_bwd_dhb_kernel(const float* depth, const float* ab,
                                                                         const float* wgt, float* dab, int B) {
  __shared__ __attribute__((aligned(16))) char Tg[2 * T_PLANE];
  __shared__ __attribute__((aligned(16))) char Th[2 * T_PLANE];
  __shared__ __attribute__((aligned(16))) char RT[2 * RT_PLANE + 16];
  __shared__ float g[128], hh[128];           // g[0..98], h[0..98]
  __shared__ float red[8];
  __shared__ double redd[4];

  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, li = lane & 31;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid < 4) ((float*)(RT + 2 * RT_PLANE))[tid] = 0.f;
  const int y = 32 * w + li;
  const int q8 = (li + 7) & ~7, rcopy = q8 - li;
  const int laneT = rcopy * TC_STRIDE + 2 * (T_OFF + 8 * h - q8);

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
    const float alpha = ab[b * 3 + 0], beta = ab[b * 3 + 1];
    const float cpsf = KP / (beta * beta);
    float mabs = 0.f;
#pragma unroll
    for (int ks = 0; ks < 7; ++ks)
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) mabs = fmaxf(mabs, fabsf(dA[ks][q][j]));      // (rows / columns outside are 0)
    for (int i = tid; i < PS; i += 256) {
      const float t = (float)(i - 49);
      const float gi = expf(-cpsf * t * t);
      g[i] = gi;
      hh[i] = t * t * gi;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mabs = fmaxf(mabs, __shfl_xor(mabs, o));
    if (lane == 0) red[w] = mabs;
    __syncthreads();                                                      // (1) tables, red
    mabs = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    // ---- operand scales (every wave derives the same values): sums / max of the two filters bound R = D*filter
    float sumg = lane < 50 ? g[49 + lane] : 0.f, sumh = lane < 50 ? hh[49 + lane] : 0.f;
    float hmax = sumh;
    if (lane > 0) { sumg *= 2.f; sumh *= 2.f; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      sumg += __shfl_xor(sumg, o);
      sumh += __shfl_xor(sumh, o);
      hmax = fmaxf(hmax, __shfl_xor(hmax, o));
    }
    const float sD = pow2_scale_to_2p13(mabs);
    const float sH = pow2_scale_to_2p13(hmax);
    const float sRg = pow2_scale_to_2p13(mabs * sumg * 1.0001f);
    const float sRh = pow2_scale_to_2p13(mabs * sumh * 1.0001f);
    build_toeplitz(Tg, g, G_SCALE, tid);
    build_toeplitz(Th, hh, sH, tid);

    auto depth_planes = [&](f16x8 (&a_hi)[7], f16x8 (&a_lo)[7]) {
#pragma unroll
      for (int ks = 0; ks < 7; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float t = dA[ks][j >> 2][j & 3] * sD;
          const _Float16 hi = (_Float16)t;
          a_hi[ks][j] = hi;
          a_lo[ks][j] = (_Float16)(t - (float)hi);
        }
    };
    __syncthreads();                                                      // (2) table copies

    // sum over this lane's accumulator pixels of wgt * acc (wgt is 0 on the plateau; rows / columns outside the image
    // are masked: their accumulators hold finite garbage of the padded GEMM).  The 64 weights of a lane are REQUESTED
    // BEFORE the column GEMM whose accumulators they multiply (load_w) and consumed after it (dot_w): loaded row by row
    // inside the reduction each of the 16 rows paid a full memory round trip -- 2 x 16 of them per sample, more than the
    // four GEMMs together (SQ counters: the waves of this kernel were parked 71 % of their cycles).  During the column
    // GEMM the row operands' planes are dead, so the 64 registers are there.
    int li_o = li, h_o = h;
    asm volatile("" : "+v"(li_o), "+v"(h_o));
    const float* wp = wgt + (size_t)b * NPIX;
    float wv[16][4];
    auto load_w = [&]() {
      const int yb = 32 * w + 4 * h_o;
      const int x3 = 96 + li_o < HS ? 96 + li_o : HS - 1;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int yy = yb + (r & 3) + 8 * (r >> 2);
        const unsigned ro = (unsigned)((yy < HS ? yy : HS - 1) * HS);      // uniform base + 32-bit lane offset
        wv[r][0] = wp[ro + (unsigned)li_o];
        wv[r][1] = wp[ro + 32u + (unsigned)li_o];
        wv[r][2] = wp[ro + 64u + (unsigned)li_o];
        wv[r][3] = wp[ro + (unsigned)x3];
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    auto dot_w = [&](const f32x16 (&ac)[4]) {
      const int yb = 32 * w + 4 * h_o;
      const float ok3 = 96 + li_o < HS ? 1.f : 0.f;
      float part = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int yy = yb + (r & 3) + 8 * (r >> 2);
        const float rowok = yy < HS ? 1.f : 0.f;
        part = fmaf(wv[r][0] * rowok, ac[0][r], part);
        part = fmaf(wv[r][1] * rowok, ac[1][r], part);
        part = fmaf(wv[r][2] * rowok, ac[2][r], part);
        part = fmaf(wv[r][3] * (rowok * ok3), ac[3][r], part);
      }
      return part;
    };

    f32x16 acc[4];
    {
      f16x8 a_hi[7], a_lo[7];
      depth_planes(a_hi, a_lo);
      zero_acc(acc);
      gemm_rows_toeplitz(acc, a_hi, a_lo, Tg, laneT);                     // Rg = D G
    }
    if (w < 3) store_rt<false>(acc, sRg / (sD * G_SCALE), RT, w, h, li);
    else store_rt<true>(acc, sRg / (sD * G_SCALE), RT, w, h, li);
    // the depth rows again for the second product (L2): requested here, a whole column GEMM ahead of their use (neither
    // they nor their planes are kept across the first product)
    load_depth(b);
    load_w();
    __syncthreads();                                                      // (3) R^T = Rg
    zero_acc(acc);
    gemm_toeplitz_rt(acc, Th, RT, laneT, w, h, li);                       // H Rg, scale sH * sRg
    double db = (double)dot_w(acc) / ((double)sH * (double)sRg);
    __syncthreads();                                                      // (4) every wave is done with Rg
    {
      f16x8 a_hi[7], a_lo[7];
      depth_planes(a_hi, a_lo);
      zero_acc(acc);
      gemm_rows_toeplitz(acc, a_hi, a_lo, Th, laneT);                     // Rh = D H
    }
    if (w < 3) store_rt<false>(acc, sRh / (sD * sH), RT, w, h, li);
    else store_rt<true>(acc, sRh / (sD * sH), RT, w, h, li);
    if (b + (int)gridDim.x < B) load_depth(b + gridDim.x);               // the planes are dead: the next sample's rows
    load_w();
    __syncthreads();                                                      // (5) R^T = Rh
    zero_acc(acc);
    gemm_toeplitz_rt(acc, Tg, RT, laneT, w, h, li);                       // G Rh, scale 2^13 * sRh
    db += (double)dot_w(acc) / ((double)G_SCALE * (double)sRh);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) db += __shfl_xor(db, o);
    if (lane == 0) redd[w] = db;
    __syncthreads();                                                      // (6) partial sums; LDS free for the next sample
    if (tid == 0) {
      const double DB = (redd[0] + redd[1]) + (redd[2] + redd[3]);
      // raw = sum D e^{-c r^2}: d raw / d beta = dhb * 2 Kp / beta^3 ; HR = alpha * raw
      dab[b * 3 + 1] = (float)(DB * (double)alpha * 2.0 * (double)KP / ((double)beta * beta * beta));
    }
  }
}

extern "C" int tpsf_forward(const float* depth, const float* alpha_beta, float* HR, float* LR_deg, float* psf,
                            int B, void* stream) {
  if (!depth || !alpha_beta || !HR || !LR_deg || !psf || B <= 0) return TSR_ERR_ARG;
  const int grid = B < 256 * TPSF_FWD_OCC ? B : 256 * TPSF_FWD_OCC;      // resident workgroups x 256 CUs, persistent
  hipLaunchKernelGGL(tpsf_fwd_mfma_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, depth, alpha_beta, HR,
                     LR_deg, psf, B);
  return tsr_check_launch();
}

extern "C" int tpsf_backward(const float* depth, const float* alpha_beta, const float* HR, const float* dLR_deg,
                             float* d_alpha_beta, float* work, int B, void* stream) {
  if (!depth || !alpha_beta || !HR || !dLR_deg || !d_alpha_beta || !work || B <= 0) return TSR_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(tpsf_bwd_pool_kernel, dim3(B < 2048 ? B : 2048), dim3(256), 0, st, depth, alpha_beta, HR, dLR_deg,
                     d_alpha_beta, work, B);
  const int grid = B < 256 * TPSF_BWD_OCC ? B : 256 * TPSF_BWD_OCC;      // persistent workgroups, TPSF_BWD_OCC per CU
  hipLaunchKernelGGL(tpsf_bwd_dhb_kernel, dim3(grid), dim3(256), 0, st, depth, alpha_beta, work, d_alpha_beta, B);
  return tsr_check_launch();
}

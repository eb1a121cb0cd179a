// Weight gradient on the bf16 matrix cores with split-fp32 operands (see conv_mfma_split16.hip for the
// arithmetic: x = x1+x2+x3 in bf16, six cross products per K step, fp32 accumulate -> fp32-equivalent).
// Same decomposition as wgrad_mfma_f32_kernel: a workgroup owns 64 co x 64 ci x one kernel row and sweeps
// (image, 8x8 patch) work items; M = co, N = ci, K = pixels.  v_mfma_f32_32x32x16_bf16 wants 8 consecutive K
// per lane, so the tiles are staged channel-major with the 8 patch ROWS of one column contiguous:
//   dzt[plane][x][co][8 y],  at[plane][x'][ci][8 y]   (x' = x + kw: a tap only moves the column index, the
// kernel-row shift kh is applied when the input tile is staged), i.e. one K = 16 step = 2 columns x 8 rows,
// every fragment one aligned, bank-conflict-free ds_read_b128.  The transpose + bf16 split happens in
// registers while staging (8 strided float4 loads per thread-item).
#include "tsr_common.h"
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <bool F16> struct WPlane;
template <> struct WPlane<false> {
  typedef __bf16 T; typedef bf16x8 V8;
  static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct WPlane<true> {
  typedef _Float16 T; typedef f16x8 V8;
  static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

// power-of-two scale that brings m into [2^13, 2^14) (1 for m == 0)
__device__ __forceinline__ float pow2_scale_to_2p13(float m) {
  if (!(m > 0.f)) return 1.f;
  int e = (int)((__float_as_uint(m) >> 23) & 0xFF) - 127;
  int be = 13 - e + 127;
  be = be < 1 ? 1 : (be > 254 ? 254 : be);
  return __uint_as_float((unsigned)be << 23);
}

struct WgradBArgs {
  const float* a;  int a_ctot; int a_coff; int cin;
  const float* a_scale; const float* a_shift;
  const float* dz; int dz_ctot; int dz_coff; int cout;
  float* slab; float* bslab;
  int B, H, W, nsplit;
  int tiles_x, tiles_y;
  const float* a_amax; const float* dz_amax;   // fp16 form: device scalars max|a| (raw), max|dz|
};

template <int NS, bool F16>
__device__ __forceinline__ void split_store8(const float (&v)[8], char* dst, int plane_stride, float sc) {
  typedef typename WPlane<F16>::T PT;
  typedef typename WPlane<F16>::V8 PV8;
  float r[8];
#pragma unroll
  for (int y = 0; y < 8; ++y) r[y] = F16 ? v[y] * sc : v[y];
#pragma unroll
  for (int p = 0; p < NS; ++p) {
    PV8 q;
#pragma unroll
    for (int y = 0; y < 8; ++y) {
      q[y] = (PT)r[y];
      r[y] -= (float)q[y];
    }
    *(PV8*)(dst + p * plane_stride) = q;
  }
}

template <int KS, int NS, bool F16>
__global__ __launch_bounds__(256, 2) void wgrad_mfma_bf16s_kernel(const WgradBArgs g) {
  typedef typename WPlane<F16>::V8 PV8;
  typedef typename WPlane<F16>::T PV4 __attribute__((ext_vector_type(4)));
  constexpr int P = KS / 2;
  constexpr int XH = 8 + KS - 1;
  constexpr int DZ_PLANE = 8 * 64 * 16;        // bytes per bf16 plane of the dz tile
  constexpr int A_PLANE = XH * 64 * 16;
  constexpr int NDZ = 8 * 16;                  // staging items (column, channel quad) of the dz tile
  constexpr int NA = XH * 16;
  constexpr int NPROD = NS == 3 ? 6 : (NS == 2 ? 3 : 1);
  constexpr int PA[6] = {NS == 3 ? 2 : (NS == 2 ? 1 : 0), 0, NS == 3 ? 1 : 0, 1, 0, 0};
  constexpr int PB[6] = {0, NS == 3 ? 2 : (NS == 2 ? 1 : 0), NS == 3 ? 1 : 0, 0, 1, 0};
  __shared__ __attribute__((aligned(16))) char lds[NS * (DZ_PLANE + A_PLANE) + 8 * 64 * 4];
  char* dzt = lds;
  char* at = lds + NS * DZ_PLANE;
  float* bred = (float*)(lds + NS * (DZ_PLANE + A_PLANE));    // [8 x][64 co] bias partials

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
  const bool do_bias = g.bslab && cib == 0 && kh == 0;
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};        // this thread's dz item: 4 channels, summed over rows and patches

  // fragment pieces: a lane holds 4 of the 8 patch rows (its half h) of one column = 8 B; an MFMA K step of 16 is
  // (2 columns x 8 rows), i.e. two pieces per lane.  The columns of consecutive taps kw overlap, so every piece is
  // read from LDS once per work item and the 16-B fragments are sliding windows over the piece registers (reading
  // whole 16-B fragments per (column pair, tap) moves 2.4x the LDS bytes).
  // LDS slot of channel ch within a column: ch ^ ((ch >> 4) & 3).  The staging writes of one instruction go to
  // channels 4q + j of block blk for (q, blk) = 16 lanes: unswizzled they fall on 4 of the 16 16-B bank groups
  // (16-way conflict, ~2.5 k LDS cycles per work item against 1.9 k MFMA cycles); the XOR with the block index
  // spreads them over all 16 and keeps 16 consecutive channels of a block on 16 different groups for the reads.
  auto swz = [](int ch) { return ch ^ ((ch >> 4) & 3); };
  const int pieceA = swz(wc * 32 + li) * 16 + 8 * h;
  const int pieceB = swz(wi * 32 + li) * 16 + 8 * h;

  // ---- staging roles (fixed per thread).  Round 0: thread -> one (column, 4-channel quad) item, 8 rows as
  // float4 (dz items 0..127, input-tile items 0..127).  Round 1: the remaining NA-128 input-tile items are
  // cut into single channels so that all threads share them evenly (8 scalar loads each).
  constexpr int NA1 = NA > 128 ? NA - 128 : 0;            // input-tile items left for round 1
  const bool r0_dz = tid < NDZ;
  const int r0_i = r0_dz ? tid : tid - NDZ;               // item index within its tile (a items < 128)
  const int r0_q = r0_i & 3, r0_blk = (r0_i >> 2) & 3, r0_x = r0_i >> 4;
  const bool r0_on = r0_dz || r0_i < NA;
  const int r1_item = 128 + (tid >> 2), r1_j = tid & 3;   // round-1: item (>=128), channel within quad
  const bool r1_on = (tid >> 2) < NA1;
  const int r1_q = r1_item & 3, r1_blk = (r1_item >> 2) & 3, r1_x = r1_item >> 4;
  f32x4 sc0 = {1.f, 1.f, 1.f, 1.f}, sh0 = {0.f, 0.f, 0.f, 0.f};
  float sc1 = 1.f, sh1 = 0.f;
  if (g.a_scale) {
    if (!r0_dz && r0_on) {
      sc0 = *(const f32x4*)(g.a_scale + cib * 64 + r0_blk * 16 + r0_q * 4);
      sh0 = *(const f32x4*)(g.a_shift + cib * 64 + r0_blk * 16 + r0_q * 4);
    }
    if (r1_on) {
      sc1 = g.a_scale[cib * 64 + r1_blk * 16 + r1_q * 4 + r1_j];
      sh1 = g.a_shift[cib * 64 + r1_blk * 16 + r1_q * 4 + r1_j];
    }
  }

  // fp16 planes: power-of-two scales of both operands (input: bound of the fused transform), undone at the end
  float s_a = 1.f, s_d = 1.f;
  if (F16) {
    float ma = g.a_amax ? *g.a_amax : 0.f;
    if (g.a_scale) {
      __shared__ float bnd[8];
      float ms = 0.f, mt = 0.f;
      for (int c = tid; c < 64; c += 256) {
        ms = fmaxf(ms, fabsf(g.a_scale[cib * 64 + c]));
        mt = fmaxf(mt, fabsf(g.a_shift[cib * 64 + c]));
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        ms = fmaxf(ms, __shfl_xor(ms, o));
        mt = fmaxf(mt, __shfl_xor(mt, o));
      }
      if (tid == 0) { bnd[0] = ms; bnd[1] = mt; }
      __syncthreads();
      ma = ma * bnd[0] + bnd[1];
    }
    s_a = pow2_scale_to_2p13(ma);
    s_d = pow2_scale_to_2p13(g.dz_amax ? *g.dz_amax : 0.f);
  }

  const int tpi = g.tiles_x * g.tiles_y;
  const long total_items = (long)g.B * tpi;
  const long per = (total_items + g.nsplit - 1) / g.nsplit;
  const long it0 = (long)sp * per;
  const long it1 = it0 + per < total_items ? it0 + per : total_items;

  f32x4 v0[8];
  float v1[8];
  unsigned ok0 = 0, ok1 = 0;      // per-row in-bounds masks of the loaded values (transform applies to those only)
  auto load_item = [&](long item) {
    const int b = (int)(item / tpi);
    const int trem = (int)(item - (long)b * tpi);
    const int ty = trem / g.tiles_x, tx = trem - ty * g.tiles_x;
    const int y0 = ty * 8, x0 = tx * 8;
    ok0 = ok1 = 0;
    if (r0_on) {
      if (r0_dz) {
        const int gx = x0 + r0_x;
        const float* base = g.dz + (((size_t)b * dz_blocks + (dz_c0 >> 4) + r0_blk) * HW) * 16 + r0_q * 4;
#pragma unroll
        for (int y = 0; y < 8; ++y) {
          const int gy = y0 + y;
          v0[y] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (gy < g.H && gx < g.W) v0[y] = *(const f32x4*)(base + (size_t)(gy * g.W + gx) * 16);
        }
      } else {
        const int gx = x0 + r0_x - P;
        const float* base = g.a + (((size_t)b * a_blocks + (a_c0 >> 4) + r0_blk) * HW) * 16 + r0_q * 4;
#pragma unroll
        for (int y = 0; y < 8; ++y) {
          const int gy = y0 + y + kh - P;
          v0[y] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (gy >= 0 && gy < g.H && gx >= 0 && gx < g.W) {
            v0[y] = *(const f32x4*)(base + (size_t)(gy * g.W + gx) * 16);
            ok0 |= 1u << y;
          }
        }
      }
    }
    if (NA1 > 0 && r1_on) {
      const int gx = x0 + r1_x - P;
      const float* base = g.a + (((size_t)b * a_blocks + (a_c0 >> 4) + r1_blk) * HW) * 16 + r1_q * 4 + r1_j;
#pragma unroll
      for (int y = 0; y < 8; ++y) {
        const int gy = y0 + y + kh - P;
        v1[y] = 0.f;
        if (gy >= 0 && gy < g.H && gx >= 0 && gx < g.W) {
          v1[y] = base[(size_t)(gy * g.W + gx) * 16];
          ok1 |= 1u << y;
        }
      }
    }
  };
  auto store_item = [&]() {
    if (r0_on) {
      char* dst = (r0_dz ? dzt : at) + (r0_x * 64) * 16;
      const int pstride = r0_dz ? DZ_PLANE : A_PLANE;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float col[8];
#pragma unroll
        for (int y = 0; y < 8; ++y) {
          col[y] = v0[y][j];
          if (!r0_dz && g.a_scale && ((ok0 >> y) & 1)) col[y] = tsr_relu(fmaf(col[y], sc0[j], sh0[j]));
        }
        if (r0_dz && do_bias) {
#pragma unroll
          for (int y = 0; y < 8; ++y) bsum[j] += col[y];
        }
        split_store8<NS, F16>(col, dst + swz(r0_blk * 16 + r0_q * 4 + j) * 16, pstride, r0_dz ? s_d : s_a);
      }
    }
    if (NA1 > 0 && r1_on) {
      float col[8];
#pragma unroll
      for (int y = 0; y < 8; ++y) {
        col[y] = v1[y];
        if (g.a_scale && ((ok1 >> y) & 1)) col[y] = tsr_relu(fmaf(col[y], sc1, sh1));
      }
      split_store8<NS, F16>(col, at + (r1_x * 64 + swz(r1_blk * 16 + r1_q * 4 + r1_j)) * 16, A_PLANE, s_a);
    }
  };

  if (it0 < it1) load_item(it0);
  for (long item = it0; item < it1; ++item) {
    __syncthreads();               // previous patch's fragment reads are done
    store_item();
    __syncthreads();
    if (item + 1 < it1) load_item(item + 1);     // next patch's global loads fly under this patch's MFMAs
    PV4 ap[NS][XH];                 // input-tile pieces, one per (plane, column)
#pragma unroll
    for (int x = 0; x < XH; ++x)
#pragma unroll
      for (int p = 0; p < NS; ++p) ap[p][x] = *(const PV4*)(at + p * A_PLANE + x * (64 * 16) + pieceB);
#pragma unroll
    for (int xp = 0; xp < 4; ++xp) {
      PV8 af[NS];
#pragma unroll
      for (int p = 0; p < NS; ++p) {
        const PV4 d0 = *(const PV4*)(dzt + p * DZ_PLANE + (2 * xp) * (64 * 16) + pieceA);
        const PV4 d1 = *(const PV4*)(dzt + p * DZ_PLANE + (2 * xp + 1) * (64 * 16) + pieceA);
        af[p] = __builtin_shufflevector(d0, d1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int kw = 0; kw < KS; ++kw) {
        PV8 bf[NS];
#pragma unroll
        for (int p = 0; p < NS; ++p)
          bf[p] = __builtin_shufflevector(ap[p][2 * xp + kw], ap[p][2 * xp + kw + 1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int t = 0; t < NPROD; ++t)
          acc[kw] = WPlane<F16>::mfma(af[PA[6 - NPROD + t]], bf[PB[6 - NPROD + t]], acc[kw]);
      }
    }
  }

  const int T = KS * KS;
  float* sl = g.slab + (size_t)sp * g.cout * g.cin * T;
  const int ci = cib * 64 + wi * 32 + li;
#pragma unroll
  for (int kw = 0; kw < KS; ++kw) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cob * 64 + wc * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      sl[((size_t)co * g.cin + ci) * T + kh * KS + kw] = F16 ? acc[kw][r] * (1.f / (s_a * s_d)) : acc[kw][r];
    }
  }
  if (do_bias) {      // thread tid < 128 holds sums of its (column x, 4 channels); reduce the 8 columns
    __syncthreads();
    if (tid < NDZ) {
      const int q = tid & 3, blk = (tid >> 2) & 3, x = tid >> 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) bred[x * 64 + blk * 16 + q * 4 + j] = bsum[j];
    }
    __syncthreads();
    if (tid < 64) {
      float s = 0.f;
#pragma unroll
      for (int x = 0; x < 8; ++x) s += bred[x * 64 + tid];
      g.bslab[(size_t)sp * g.cout + cob * 64 + tid] = s;
    }
  }
}

int tsr_conv2d_wgrad_tr16(const float* a, int a_ctot, int a_coff, int cin, const float* a_scale, const float* a_shift,
                          const float* dz, int dz_ctot, int dz_coff, int cout, int ks, int planes,
                          const float* a_amax, const float* dz_amax, float* slab, float* bias_slab, int nsplit,
                          int B, int H, int W, hipStream_t st);     // wgrad_mfma_tr16.hip

extern "C" int tsr_conv2d_wgrad_bf16s(const float* a, int a_ctot, int a_coff, int cin,
                                      const float* a_scale, const float* a_shift,
                                      const float* dz, int dz_ctot, int dz_coff, int cout, int ks, int planes,
                                      const float* a_amax, const float* dz_amax,
                                      float* slab, float* bias_slab, int nsplit, int B, int H, int W, void* stream) {
  // planes = 3: bf16 (fp32-equivalent, a_amax/dz_amax unused); planes = 1: plain bf16 operands (reduced
  // precision, the "bf16" configurations); planes = -2: fp16 two-plane split with the
  // power-of-two scales derived from the device scalars a_amax = max|a| (raw) and dz_amax = max|dz|
  // planes = -1: plain bf16 operands read from bf16 CB16 TENSORS (`a`, `dz` address bf16 elements): training with bf16
  // activation storage
  if (!a || !dz || !slab || B <= 0 || H <= 0 || W <= 0 || nsplit <= 0 ||
      (planes != 3 && planes != 1 && planes != -2 && planes != -1))
    return TSR_ERR_ARG;
  if (planes == -2 && (!a_amax || !dz_amax)) return TSR_ERR_ARG;
  if ((cin & 63) || (cout & 63) || (a_ctot & 15) || (a_coff & 15) || (dz_ctot & 15) || (dz_coff & 15) ||
      a_coff + cin > a_ctot || dz_coff + cout > dz_ctot || (ks != 1 && ks != 3 && ks != 5))
    return TSR_ERR_ARG;
  if ((a_scale != nullptr) != (a_shift != nullptr)) return TSR_ERR_ARG;
  // default: the transposed-LDS-read GEMM form (wgrad_mfma_tr16.hip); TSR_WGRAD_OLD=1 keeps the register-transpose
  // kernel below for A/B measurements
  static const bool use_old = getenv("TSR_WGRAD_OLD") != nullptr;
  if (!use_old || planes == -1)
    return tsr_conv2d_wgrad_tr16(a, a_ctot, a_coff, cin, a_scale, a_shift, dz, dz_ctot, dz_coff, cout, ks, planes, a_amax,
                                 dz_amax, slab, bias_slab, nsplit, B, H, W, (hipStream_t)stream);
  WgradBArgs g;
  g.a = a; g.a_ctot = a_ctot; g.a_coff = a_coff; g.cin = cin; g.a_scale = a_scale; g.a_shift = a_shift;
  g.dz = dz; g.dz_ctot = dz_ctot; g.dz_coff = dz_coff; g.cout = cout;
  g.slab = slab; g.bslab = bias_slab; g.B = B; g.H = H; g.W = W; g.nsplit = nsplit;
  g.tiles_x = (W + 7) / 8; g.tiles_y = (H + 7) / 8;
  g.a_amax = a_amax; g.dz_amax = dz_amax;
  const int grid = nsplit * ks * (cout >> 6) * (cin >> 6);
  hipStream_t st = (hipStream_t)stream;
  if (planes == 3) {
    if (ks == 1) hipLaunchKernelGGL((wgrad_mfma_bf16s_kernel<1, 3, false>), dim3(grid), dim3(256), 0, st, g);
    else if (ks == 3) hipLaunchKernelGGL((wgrad_mfma_bf16s_kernel<3, 3, false>), dim3(grid), dim3(256), 0, st, g);
    else hipLaunchKernelGGL((wgrad_mfma_bf16s_kernel<5, 3, false>), dim3(grid), dim3(256), 0, st, g);
  } else if (planes == 1) {
    if (ks == 1) hipLaunchKernelGGL((wgrad_mfma_bf16s_kernel<1, 1, false>), dim3(grid), dim3(256), 0, st, g);
    else if (ks == 3) hipLaunchKernelGGL((wgrad_mfma_bf16s_kernel<3, 1, false>), dim3(grid), dim3(256), 0, st, g);
    else hipLaunchKernelGGL((wgrad_mfma_bf16s_kernel<5, 1, false>), dim3(grid), dim3(256), 0, st, g);
  } else {
    if (ks == 1) hipLaunchKernelGGL((wgrad_mfma_bf16s_kernel<1, 2, true>), dim3(grid), dim3(256), 0, st, g);
    else if (ks == 3) hipLaunchKernelGGL((wgrad_mfma_bf16s_kernel<3, 2, true>), dim3(grid), dim3(256), 0, st, g);
    else hipLaunchKernelGGL((wgrad_mfma_bf16s_kernel<5, 2, true>), dim3(grid), dim3(256), 0, st, g);
  }
  return tsr_check_launch();
}

// Weight gradient as a plain MFMA GEMM over pixels, fed by TRANSPOSED LDS reads (gfx950 ds_read_b64_tr_b16).
//
//   dW[co][ci][kh][kw] = sum over (image, y, x) of dz[co][y][x] * a[ci][y+kh-P][x+kw-P]
//
// GEMM view per tap: M = co, N = ci, K = pixels.  The 16-bit MFMA wants 8 consecutive K (pixels) of one channel per
// lane, but activations live channel-minor in HBM (CB16: [pixel][16 ch]).  The previous kernel transposed in registers
// while staging (8 strided loads per thread item) and had to re-stage a row-shifted input tile per kernel row; both made
// it staging-bound (31 % of the f16 MFMA rate).  Here tiles are staged exactly as they sit in memory -- coalesced 16-B
// loads of whole CB16 lines, split into 16-bit planes, written [plane][16-ch block][pixel][16 ch] -- and the MFMA
// fragments come out of LDS already transposed: one ds_read_b64_tr_b16 hands each lane 4 pixels of its channel.  A tap
// (kh, kw) is then only an LDS address offset of the B operand: kh is folded into which 4 input rows are staged, kw is
// an immediate +32 B per column.
//
// Workgroup = CO x CI output channels x the KS taps of ONE kernel row (acc: KS tiles per 32x32 wave tile), sweeping
// (image, 4-row x 8-column patch) work items of its batch split; per item two K = 16 steps (2 rows x 8 columns each).
// LDS is double-buffered: the next item's global loads are issued before the MFMAs of the current one and converted /
// written after them, one barrier per item.  128 x 128 tiles (8 waves, wave = 64 co x 32 ci) stage 2x fewer bytes per
// MFMA than the 64 x 64 x one-row workgroups of the old kernel and at a fraction of the instruction cost.
//
// Bank conflicts: a 16-lane group reads 4 consecutive pixels x 32 B = 128 contiguous bytes; the two groups of a
// 32-lane half read the two 16-channel blocks of the wave's 32 channels, whose LDS regions are an odd multiple of 128 B
// apart (pixel counts padded to 4 mod 8), i.e. complementary halves of the 64-bank row for ANY start pixel: conflict
// free for every tap.  Staging writes are 8 B per lane, 512 contiguous bytes per wave: conflict free.
#include "tsr_common.h"
#include <type_traits>

typedef __bf16 tb16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 tb16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 th16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 th16x4 __attribute__((ext_vector_type(4)));
typedef short tv4i16 __attribute__((__vector_size__(4 * sizeof(short))));

template <bool F16> struct TPlane;
template <> struct TPlane<false> {
  typedef __bf16 T; typedef tb16x8 V8; typedef tb16x4 V4;
  static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct TPlane<true> {
  typedef _Float16 T; typedef th16x8 V8; typedef th16x4 V4;
  static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

struct WgradTArgs {
  const float* a;  int a_ctot; int a_coff; int cin;
  const float* a_scale; const float* a_shift;
  const float* dz; int dz_ctot; int dz_coff; int cout;
  float* slab; float* bslab;
  int B, H, W, nsplit;
  int tiles_x, tiles_y;                        // 8-column x 4-row patches
  const float* a_amax; const float* dz_amax;   // fp16 form: device scalars max|a| (raw), max|dz|
};

__device__ __forceinline__ float tr16_pow2_scale(float m) {   // power of two bringing m into [2^13, 2^14); 1 for 0
  if (!(m > 0.f)) return 1.f;
  int e = (int)((__float_as_uint(m) >> 23) & 0xFF) - 127;
  int be = 13 - e + 127;
  be = be < 1 ? 1 : (be > 254 ? 254 : be);
  return __uint_as_float((unsigned)be << 23);
}

constexpr int tr16_pad_px(int n) {     // pixel count of a block region: == 4 (mod 8) -> region stride == 128 (mod 256) B
  int p = n;
  while ((p & 7) != 4) ++p;
  return p;
}

template <int KS, int KHW, int CO, int CI, int WM, int NS, bool F16>
struct WgradTGeom {
  static constexpr int NWM = CO / WM, NWN = CI / 32, NWAVE = NWM * NWN, NT = 64 * NWAVE;
  static constexpr int MT = WM / 32;
  static constexpr int NKG = (KS + KHW - 1) / KHW;          // kernel-row groups (one workgroup each)
  static constexpr int NTAP = KHW * KS;                     // taps (accumulator tiles per 32x32 wave tile) per workgroup
  static constexpr int ACOLS = 8 + KS - 1, AROWS = 4 + KHW - 1;
  static constexpr int DZ_PX = 32, A_PX = AROWS * ACOLS;
  static constexpr int DZ_BLKB = tr16_pad_px(DZ_PX) * 32, A_BLKB = tr16_pad_px(A_PX) * 32;
  static constexpr int DZ_PLANEB = (CO / 16) * DZ_BLKB, A_PLANEB = (CI / 16) * A_BLKB;
  static constexpr int BUFB = NS * (DZ_PLANEB + A_PLANEB);
  // staging map: thread = (channel quad, pixel-in-group, 16-channel block, pixel group) -- one channel quad of one block
  // per thread for the whole kernel (BN scale / shift in 8 registers, 4 bias partials), groups of 4 consecutive pixels
  static constexpr int NPG_DZ = NT / 16 / (CO / 16), NPG_A = NT / 16 / (CI / 16);     // pixel groups per pass
  static constexpr int NIT_DZ = (DZ_PX / 4 + NPG_DZ - 1) / NPG_DZ, NIT_A = (A_PX / 4 + NPG_A - 1) / NPG_A;
  static constexpr int BRED_SLOTS = 4 * NPG_DZ;
  static constexpr int LDSB = 2 * BUFB > BRED_SLOTS * CO * 4 ? 2 * BUFB : BRED_SLOTS * CO * 4;
  static_assert(NT % 16 == 0 && (NT / 16) % (CO / 16) == 0 && (NT / 16) % (CI / 16) == 0 && A_PX % 4 == 0, "staging map");
};

// (A role-specialised form -- 4 MFMA-only waves + 4 staging-only waves per workgroup -- was measured in round 2 and is gone:
// 3.91 vs 3.46 ms at 5x5 128x128)
// IO16 (NS = 1, bf16): `a` and `dz` are bf16 CB16 tensors (training with bf16 activation storage): staging copies 8-B
// quads as they are (dz, and `a` without a fused transform) or applies relu(a*scale+shift) in fp32 and rounds back.
template <int KS, int KHW, int CO, int CI, int WM, int NS, bool F16, bool IO16 = false>
__global__ __launch_bounds__((CO / WM) * (CI / 32) * 64, 2) void wgrad_tr16_kernel(const WgradTArgs g) {
  static_assert(!IO16 || (NS == 1 && !F16), "bf16 tensors: one bf16 plane");
  typedef WgradTGeom<KS, KHW, CO, CI, WM, NS, F16> G;
  typedef typename TPlane<F16>::T PT;
  typedef typename TPlane<F16>::V8 PV8;
  typedef typename TPlane<F16>::V4 PV4;
  typedef __attribute__((address_space(3))) tv4i16* lds_v4;
  constexpr int P = KS / 2;
  constexpr int NT = G::NT, MT = G::MT;            // NT = staging threads (= MFMA threads)
  constexpr int NTT = NT;                            // threads of the workgroup
  constexpr int NPROD = NS == 3 ? 6 : (NS == 2 ? 3 : 1);
  // products ordered small -> large (plane 0 = most significant)
  constexpr int PA[6] = {NS == 3 ? 2 : (NS == 2 ? 1 : 0), 0, NS == 3 ? 1 : 0, 1, 0, 0};
  constexpr int PB[6] = {0, NS == 3 ? 2 : (NS == 2 ? 1 : 0), NS == 3 ? 1 : 0, 0, 1, 0};

  __shared__ __attribute__((aligned(16))) char lds[G::LDSB];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tid = threadIdx.x;
  const int wm = wave / G::NWN, wn = wave - wm * G::NWN;
  const int h = lane >> 5, li = lane & 31;

  const int nci = g.cin / CI, nco = g.cout / CO;
  // XCD-aware block order (speed only): the dispatcher deals blocks round-robin over the 8 XCDs; remap so that each XCD
  // runs a contiguous range of logical ids -- the KS kernel-row workgroups (and channel tiles) of one batch split read
  // the same dz / input pixels at the same time and now share one L2 instead of fetching them into five.
  int bid;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int cib = bid % nci; bid /= nci;
  const int cob = bid % nco; bid /= nco;
  const int kg = bid % G::NKG;          // kernel-row group: rows kh0 .. kh0 + KHW - 1 (the last group may be short)
  const int sp = bid / G::NKG;
  const int kh0 = kg * KHW;

  const int HW = g.H * g.W;
  const int a_blocks = g.a_ctot >> 4, dz_blocks = g.dz_ctot >> 4;
  const int a_c0 = g.a_coff + cib * CI, dz_c0 = g.dz_coff + cob * CO;
  const bool do_bias = g.bslab && cib == 0 && kg == 0;

  // fp16 planes: power-of-two scales of both operands (input: bound of the fused transform), undone at the end
  float s_a = 1.f, s_d = 1.f;
  if (F16) {
    float ma = g.a_amax ? *g.a_amax : 0.f;
    if (g.a_scale) {
      float* bnd = (float*)lds;
      float ms = 0.f, mt = 0.f;
      for (int c = threadIdx.x; c < CI; c += NTT) {
        ms = fmaxf(ms, fabsf(g.a_scale[cib * CI + c]));
        mt = fmaxf(mt, fabsf(g.a_shift[cib * CI + c]));
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        ms = fmaxf(ms, __shfl_xor(ms, o));
        mt = fmaxf(mt, __shfl_xor(mt, o));
      }
      if (lane == 0) { bnd[wave * 2] = ms; bnd[wave * 2 + 1] = mt; }
      __syncthreads();
      ms = 0.f; mt = 0.f;
      for (int w = 0; w < NTT / 64; ++w) { ms = fmaxf(ms, bnd[w * 2]); mt = fmaxf(mt, bnd[w * 2 + 1]); }
      __syncthreads();
      ma = ma * ms + mt;
    }
    s_a = tr16_pow2_scale(ma);
    s_d = tr16_pow2_scale(g.dz_amax ? *g.dz_amax : 0.f);
  }
  const float s_ah = g.a_scale ? 0.5f * s_a : s_a;      // staging scale of the input operand (virtual inputs arrive as 2 relu(.))

  // (in the role-specialised form the accumulators must be live in the MFMA role's branch ONLY -- zeroed, used and
  // written out there -- or the register allocator keeps 160 of them alive through the staging role's loop)
  f32x16 acc[MT][G::NTAP];
  auto zero_acc = [&]() {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int k = 0; k < G::NTAP; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][k][r] = 0.f;
  };

  // ---- staging roles: item i -> (quad = i & 3, pixel, block); consecutive threads walk a CB16 line, then the row
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};

  // work items of this split: a contiguous range of (image, patch row, patch column), walked with counters (no
  // division in the loop; the launcher guarantees B * patches < 2^31)
  const int tpi = g.tiles_x * g.tiles_y;
  const int total_items = g.B * tpi;
  const int per = (total_items + g.nsplit - 1) / g.nsplit;
  const int it0 = sp * per;
  const int it1 = it0 + per < total_items ? it0 + per : total_items;
  int nb = it0 / tpi, nty = (it0 - nb * tpi) / g.tiles_x, ntx = it0 - nb * tpi - nty * g.tiles_x;   // next item to LOAD

  // Per-thread staging constants (invariant over work items): element offset of each slot relative to the item's
  // (image, y0, x0) corner -- that corner is workgroup-uniform, so the per-item address is a scalar base + a 32-bit
  // VGPR offset -- and its (row, column) for the bounds tests.  Loads are BRANCH-FREE: an out-of-image slot reads its
  // block's corner pixel instead (always valid) and is zeroed by a select when it is stored (a conditional load makes
  // hipcc branch around every load and drain vmcnt per slot).
  const int sq = tid & 3, spl = (tid >> 2) & 3;
  const int blkd = (tid >> 4) % (CO / 16), pgd = (tid >> 4) / (CO / 16);
  const int blka = (tid >> 4) % (CI / 16), pga = (tid >> 4) / (CI / 16);
  const int safed = blkd * HW * 16 + sq * 4, safea = blka * HW * 16 + sq * 4;
  int offd[G::NIT_DZ], offa[G::NIT_A], rcd[G::NIT_DZ], rca[G::NIT_A], ldsd[G::NIT_DZ], ldsa[G::NIT_A];
#pragma unroll
  for (int j = 0; j < G::NIT_DZ; ++j) {
    const int px = 4 * (pgd + j * G::NPG_DZ) + spl;          // may run past the tile in the last pass: never stored
    offd[j] = safed + ((px >> 3) * g.W + (px & 7)) * 16;
    rcd[j] = ((px >> 3) << 8) | (px & 7);
    ldsd[j] = blkd * G::DZ_BLKB + px * 32 + sq * 8;
  }
#pragma unroll
  for (int j = 0; j < G::NIT_A; ++j) {
    const int px = 4 * (pga + j * G::NPG_A) + spl;
    const int r = px / G::ACOLS, cx = px - r * G::ACOLS;
    offa[j] = safea + ((r + kh0 - P) * g.W + cx - P) * 16;
    rca[j] = (r << 8) | cx;
    ldsa[j] = blka * G::A_BLKB + px * 32 + sq * 8;
  }
  // the fused input transform's (scale, shift) of this thread's channel quad; the fp16 / bf16 split forms carry the
  // staging scale in it (2 relu(t) = t + |t| is positively homogeneous), the bf16-tensor form keeps relu(bn(z)) itself
  f32x4 tsc = {0.f, 0.f, 0.f, 0.f}, tsh = {0.f, 0.f, 0.f, 0.f};
  if (g.a_scale) {
    const float pre = IO16 ? 1.f : s_ah;
    tsc = *(const f32x4*)(g.a_scale + cib * CI + blka * 16 + sq * 4) * pre;
    tsh = *(const f32x4*)(g.a_shift + cib * CI + blka * 16 + sq * 4) * pre;
  }

  f32x4 hd[G::NIT_DZ], ha[G::NIT_A];
  unsigned okd = 0, oka = 0;

  auto load_item = [&]() {
    const int b = nb, y0 = nty * 4, x0 = ntx * 8;
    if (++ntx == g.tiles_x) {
      ntx = 0;
      if (++nty == g.tiles_y) { nty = 0; ++nb; }
    }
    const size_t dzo = (((size_t)b * dz_blocks + (dz_c0 >> 4)) * HW + y0 * g.W + x0) * 16;
    const size_t ao = (((size_t)b * a_blocks + (a_c0 >> 4)) * HW + y0 * g.W + x0) * 16;
    const float* dzb = g.dz + dzo;
    const float* ab = g.a + ao;
    const __bf16* dzb16 = (const __bf16*)g.dz + dzo;
    const __bf16* ab16 = (const __bf16*)g.a + ao;
    okd = oka = 0;
#pragma unroll
    for (int j = 0; j < G::NIT_DZ; ++j) {
      const bool ok = (y0 + (rcd[j] >> 8) < g.H) & (x0 + (rcd[j] & 255) < g.W) & ((rcd[j] >> 8) < 4);
      if constexpr (IO16) {      // 4 bf16 = 8 B, carried in the low half of the slot
        const float2 t = *(const float2*)(dzb16 + (ok ? offd[j] : safed));
        hd[j][0] = t.x; hd[j][1] = t.y;
      } else {
        hd[j] = *(const f32x4*)(dzb + (ok ? offd[j] : safed));
      }
      okd |= (unsigned)ok << j;
    }
#pragma unroll
    for (int j = 0; j < G::NIT_A; ++j) {
      const int gy = y0 + (rca[j] >> 8) + kh0 - P, gx = x0 + (rca[j] & 255) - P;
      const bool ok = (gy >= 0) & (gy < g.H) & (gx >= 0) & (gx < g.W) & ((rca[j] >> 8) < G::AROWS);
      if constexpr (IO16) {
        const float2 t = *(const float2*)(ab16 + (ok ? offa[j] : safea));
        ha[j][0] = t.x; ha[j][1] = t.y;
      } else {
        ha[j] = *(const f32x4*)(ab + (ok ? offa[j] : safea));
      }
      oka |= (unsigned)ok << j;
    }
  };

  // `mult` carries both the power-of-two operand scale and the in-image mask (0 for a slot that read its safe pixel)
  auto split_store = [&](f32x4 v, char* dst, int plane_stride, float mult) {
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] *= mult;
#pragma unroll
    for (int p = 0; p < NS; ++p) {
      PV4 qv;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        qv[c] = (PT)v[c];
        v[c] -= (float)qv[c];
      }
      *(PV4*)(dst + p * plane_stride) = qv;
    }
  };

  auto store_item = [&](int buf) {
    char* dzt = lds + buf * G::BUFB;
    char* at = dzt + NS * G::DZ_PLANEB;
    if constexpr (IO16) {
      typedef __bf16 io_bf16x4 __attribute__((ext_vector_type(4)));
#pragma unroll
      for (int j = 0; j < G::NIT_DZ; ++j) {
        if (4 * (j + 1) * G::NPG_DZ <= G::DZ_PX || 4 * (pgd + j * G::NPG_DZ) < G::DZ_PX) {
          const bool ok = (okd >> j) & 1;
          const float2 raw = ok ? make_float2(hd[j][0], hd[j][1]) : make_float2(0.f, 0.f);
          if (do_bias) {
            const io_bf16x4 q = __builtin_bit_cast(io_bf16x4, raw);
#pragma unroll
            for (int c = 0; c < 4; ++c) bsum[c] += (float)q[c];
          }
          *(float2*)(dzt + ldsd[j]) = raw;
        }
      }
#pragma unroll
      for (int j = 0; j < G::NIT_A; ++j) {
        if (4 * (j + 1) * G::NPG_A <= G::A_PX || 4 * (pga + j * G::NPG_A) < G::A_PX) {
          const bool ok = (oka >> j) & 1;
          float2 raw = make_float2(ha[j][0], ha[j][1]);
          if (g.a_scale) {      // relu(bn(z)) of the stored bf16 pre-activation, fp32 arithmetic, back to bf16 (branch-free)
            const io_bf16x4 zq = __builtin_bit_cast(io_bf16x4, raw);
            io_bf16x4 aq;
#pragma unroll
            for (int c = 0; c < 4; ++c) aq[c] = (__bf16)tsr_relu(fmaf((float)zq[c], tsc[c], tsh[c]));
            raw = __builtin_bit_cast(float2, aq);
          }
          *(float2*)(at + ldsa[j]) = ok ? raw : make_float2(0.f, 0.f);
        }
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < G::NIT_DZ; ++j) {
      if (4 * (j + 1) * G::NPG_DZ <= G::DZ_PX || 4 * (pgd + j * G::NPG_DZ) < G::DZ_PX) {
        const f32x4 v = hd[j];
        const bool ok = (okd >> j) & 1;
        if (do_bias) {
          const float keep = ok ? 1.f : 0.f;
#pragma unroll
          for (int c = 0; c < 4; ++c) bsum[c] = fmaf(v[c], keep, bsum[c]);
        }
        split_store(v, dzt + ldsd[j], G::DZ_PLANEB, ok ? s_d : 0.f);
      }
    }
#pragma unroll
    for (int j = 0; j < G::NIT_A; ++j) {
      if (4 * (j + 1) * G::NPG_A <= G::A_PX || 4 * (pga + j * G::NPG_A) < G::A_PX) {
        f32x4 v = ha[j];
        const bool ok = (oka >> j) & 1;
        if (g.a_scale) {      // producer's train-mode BN + ReLU, fused into the load (in-image pixels only)
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] = tsr_relu_x2(fmaf(v[c], tsc[c], tsh[c]));      // 2 relu(.) s_ah
          split_store(v, at + ldsa[j], G::A_PLANEB, ok ? 1.f : 0.f);
        } else {
          split_store(v, at + ldsa[j], G::A_PLANEB, ok ? s_ah : 0.f);
        }
      }
    }
  };

  // ---- fragment addressing (ds_read_b64_tr_b16): lane = 16*G4 + 4*q + p supplies the address of pixel-row q,
  // channels 4p..4p+3 of its 16-channel block; it receives 4 pixels of channel (lane & 15) of that block
  const int g4 = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
  // dz (A operand), co tile m: blocks 2*(wm*MT+m) + (g4&1); K half h = g4>>1 -> patch row 2s+h; read rd -> cols 4rd..
  const int a_lane = ((wm * MT) * 2 + (g4 & 1)) * G::DZ_BLKB + ((g4 >> 1) * 8 + tq) * 32 + tp * 8;
  const int b_lane = (wn * 2 + (g4 & 1)) * G::A_BLKB + ((g4 >> 1) * G::ACOLS + tq) * 32 + tp * 8;   // + row, + kw

  // One (K step, tap) = NPROD x MT MFMAs.  B fragments are ping-pong prefetched one tap ahead and the prefetch reads are
  // interleaved with the running tap's MFMAs (left alone, hipcc hoists every tap's reads to the top of the item:
  // 40 more live registers and spills at 128 x 128 x 5 taps).
  auto mma_item = [&](int buf, auto&& between_steps) {
    const char* dzt = lds + buf * G::BUFB;
    const char* at = dzt + NS * G::DZ_PLANEB;
    auto load_b = [&](PV8* bf, int s, int tap) {
      const int khl = tap / KS, kw = tap - khl * KS;
#pragma unroll
      for (int p = 0; p < NS; ++p) {
        const char* base = at + p * G::A_PLANEB + b_lane + (2 * s + khl) * (G::ACOLS * 32) + kw * 32;
        const tv4i16 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(base));
        const tv4i16 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(base + 4 * 32));
        bf[p] = __builtin_shufflevector(__builtin_bit_cast(PV4, r0), __builtin_bit_cast(PV4, r1), 0, 1, 2, 3, 4, 5, 6, 7);
      }
    };
    auto load_a = [&](PV8 (*af)[NS], int s) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int p = 0; p < NS; ++p) {
          const char* base = dzt + p * G::DZ_PLANEB + a_lane + m * 2 * G::DZ_BLKB + s * (16 * 32);
          const tv4i16 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(base));
          const tv4i16 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(base + 4 * 32));
          af[m][p] = __builtin_shufflevector(__builtin_bit_cast(PV4, r0), __builtin_bit_cast(PV4, r1), 0, 1, 2, 3, 4, 5, 6, 7);
        }
    };
    PV8 af[MT][NS], bf[2][NS];
    load_a(af, 0);
    load_b(bf[0], 0, 0);
    constexpr int NU = 2 * G::NTAP;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int s = u / G::NTAP, tap = u - s * G::NTAP;
      const int cb = u & 1;
      const bool seam = tap == G::NTAP - 1 && s == 0;       // the staging block sits between the two K steps
      if (u + 1 < NU && !seam) load_b(bf[cb ^ 1], (u + 1) / G::NTAP, (u + 1) % G::NTAP);
      // (a short last row group computes its missing rows on zero-weighted garbage-free data: the rows exist in the
      // staged tile, their products are simply not written out)
#pragma unroll
      for (int t = 0; t < NPROD; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m)
          acc[m][tap] = TPlane<F16>::mfma(af[m][PA[6 - NPROD + t]], bf[cb][PB[6 - NPROD + t]], acc[m][tap]);
      if (u + 1 < NU && !seam) {
        constexpr int NRD = 2 * NS, NMF = NPROD * MT, PER = NMF / NRD > 0 ? NMF / NRD : 1;
#pragma unroll
        for (int i = 0; i < NRD; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);   // MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // DS read
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (seam) {       // no fragment is prefetched across the staging block: its registers are needed there
        between_steps();
        __builtin_amdgcn_sched_barrier(0);
        load_a(af, 1);
        load_b(bf[cb ^ 1], 1, 0);
      }
    }
  };

  __syncthreads();                         // (the scale-bound scratch is free)
  // ---- this split's partial dW: slab[sp][co][ci][kh][kw]
  auto write_slab = [&]() {
    constexpr int T = KS * KS;
    float* sl = g.slab + (size_t)sp * g.cout * g.cin * T;
    const float inv = F16 ? 1.f / (s_a * s_d) : 1.f;
    const int ci = cib * CI + wn * 32 + li;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int tap = 0; tap < G::NTAP; ++tap) {
        if (kh0 * KS + tap < T) {          // a short last row group holds rows beyond the kernel: not written
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = cob * CO + wm * WM + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            sl[((size_t)co * g.cin + ci) * T + kh0 * KS + tap] = acc[m][tap][r] * inv;
          }
        }
      }
  };

  if (it0 < it1) {
    load_item();
    store_item(0);
    if (it0 + 1 < it1) load_item();                  // item + 1 in flight from here on
  }
  __syncthreads();
  int cur = 0;
  zero_acc();
  for (int item = it0; item < it1; ++item) {
    const bool more = item + 1 < it1;
    mma_item(cur, [&]() {
      if (more) store_item(cur ^ 1);       // the other buffer: its last readers passed the previous barrier
      if (item + 2 < it1) load_item();     // the staging registers are free again: item + 2 flies for a whole item
    });
    __syncthreads();
    cur ^= 1;
  }
  write_slab();

  if (do_bias) {       // thread slot j holds the sums of 4 channels of (pixel, block): reduce the 32 pixels
    float* bred = (float*)lds;                 // [pixel slot = 4 pgd + spl][CO]
#pragma unroll
    for (int c = 0; c < 4; ++c) bred[(pgd * 4 + spl) * CO + blkd * 16 + sq * 4 + c] = bsum[c];
    __syncthreads();
    for (int c = threadIdx.x; c < CO; c += NTT) {
      float s = 0.f;
#pragma unroll
      for (int px = 0; px < G::BRED_SLOTS; ++px) s += bred[px * CO + c];
      g.bslab[(size_t)sp * g.cout + cob * CO + c] = s;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// K = 32 form (v_mfma_f32_16x16x32_f16), fp16x3 arithmetic: the default weight-gradient kernel.  Same tiles, same staging,
// same transposing LDS reads as wgrad_tr16_kernel; the matrix instruction is the 16x16x32 one because the chip holds a
// higher clock on it under load (conv_mfma_k32.hip; measured here with the instruction swapped in place: 5x5 128x128
// 3.42 -> 3.14 ms).  K = 32 = the whole 4-row x 8-column work item: lane group g = lane >> 4 supplies patch row g, two
// ds_read_b64_tr_b16 give a lane the 8 pixels of that row for its channel.  Consequences:
//   * the two 16-lane groups of a half-wave now read two ROWS of one channel block (not two blocks): rows are pitched
//     12 pixels (384 B == 128 mod 256) in both tiles, so they still cover complementary halves of the 64 banks;
//   * accumulators are 16x16 tiles (co tile x ci tile x tap x 4 registers: the same 160 / 144 registers), and to stay
//     inside the register file the fragments are NOT all resident: the item runs as three product sweeps over its
//     taps -- dz plane 1 x a plane 0, dz 0 x a 0, dz 0 x a 1 -- holding one dz plane (WM/16 fragments) and a
//     ping-pong pair of a-fragments; plane 0 of a is read in two sweeps (LDS reads per item 76 instead of 56, at
//     2 cycles each against 1,920 MFMA cycles).  The staging block (conversion + LDS writes of the next item) sits
//     between the first and the second sweep.
template <int KS, int KHW, int CO, int CI, int WM>
struct WgradKGeom {
  static constexpr int NWM = CO / WM, NWN = CI / 32, NWAVE = NWM * NWN, NT = 64 * NWAVE;
  static constexpr int MT = WM / 16;
  static constexpr int NKG = (KS + KHW - 1) / KHW;
  static constexpr int NTAP = KHW * KS;
  static constexpr int ACOLS = 8 + KS - 1, AROWS = 4 + KHW - 1;
  static constexpr int PITCH = 12;                                   // pixels per tile row in LDS (both tiles)
  static constexpr int DZ_PX = 32, A_PX = AROWS * ACOLS;             // staged pixels per block
  static constexpr int DZ_BLKB = tr16_pad_px(4 * PITCH) * 32, A_BLKB = tr16_pad_px(AROWS * PITCH) * 32;
  static constexpr int DZ_PLANEB = (CO / 16) * DZ_BLKB, A_PLANEB = (CI / 16) * A_BLKB;
  static constexpr int BUFB = 2 * (DZ_PLANEB + A_PLANEB);
  // staging map: thread = (channel quad q, pixel-in-group pl, 16-channel block, pixel group); a thread keeps ONE channel
  // quad of ONE block for the whole kernel (its BN scale / shift live in 8 registers, its bias partial sums in 4) and
  // walks groups of 4 consecutive pixels: 16 lanes = 4 pixels x 64 B in memory, 128 contiguous bytes per plane in LDS
  static constexpr int NPG_DZ = NT / 16 / (CO / 16), NPG_A = NT / 16 / (CI / 16);     // pixel groups per pass
  static constexpr int NIT_DZ = (DZ_PX / 4 + NPG_DZ - 1) / NPG_DZ, NIT_A = (A_PX / 4 + NPG_A - 1) / NPG_A;
  static constexpr int BRED_SLOTS = 4 * NPG_DZ;
  static constexpr int LDSB = 2 * BUFB > BRED_SLOTS * CO * 4 ? 2 * BUFB : BRED_SLOTS * CO * 4;
  static_assert(NT % 16 == 0 && (NT / 16) % (CO / 16) == 0 && (NT / 16) % (CI / 16) == 0 && A_PX % 4 == 0, "staging map");
};

template <int KS, int KHW, int CO, int CI, int WM>
__global__ __launch_bounds__((CO / WM) * (CI / 32) * 64, 2) void wgrad_k32_kernel(const WgradTArgs g) {
  typedef WgradKGeom<KS, KHW, CO, CI, WM> G;
  typedef _Float16 PT;
  typedef th16x8 PV8;
  typedef th16x4 PV4;
  typedef __attribute__((address_space(3))) tv4i16* lds_v4;
  constexpr int P = KS / 2;
  constexpr int NT = G::NT, MT = G::MT;

  __shared__ __attribute__((aligned(16))) char lds[G::LDSB];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tid = threadIdx.x;
  const int wm = wave / G::NWN, wn = wave - wm * G::NWN;

  const int nci = g.cin / CI, nco = g.cout / CO;
  int bid;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int cib = bid % nci; bid /= nci;
  const int cob = bid % nco; bid /= nco;
  const int kg = bid % G::NKG;
  const int sp = bid / G::NKG;
  const int kh0 = kg * KHW;

  const int HW = g.H * g.W;
  const int a_blocks = g.a_ctot >> 4, dz_blocks = g.dz_ctot >> 4;
  const int a_c0 = g.a_coff + cib * CI, dz_c0 = g.dz_coff + cob * CO;
  const bool do_bias = g.bslab && cib == 0 && kg == 0;

  float s_a = 1.f, s_d = 1.f;
  {
    float ma = g.a_amax ? *g.a_amax : 0.f;
    if (g.a_scale) {
      float* bnd = (float*)lds;
      float ms = 0.f, mt = 0.f;
      for (int c = threadIdx.x; c < CI; c += NT) {
        ms = fmaxf(ms, fabsf(g.a_scale[cib * CI + c]));
        mt = fmaxf(mt, fabsf(g.a_shift[cib * CI + c]));
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        ms = fmaxf(ms, __shfl_xor(ms, o));
        mt = fmaxf(mt, __shfl_xor(mt, o));
      }
      if (lane == 0) { bnd[wave * 2] = ms; bnd[wave * 2 + 1] = mt; }
      __syncthreads();
      ms = 0.f; mt = 0.f;
      for (int w = 0; w < NT / 64; ++w) { ms = fmaxf(ms, bnd[w * 2]); mt = fmaxf(mt, bnd[w * 2 + 1]); }
      __syncthreads();
      ma = ma * ms + mt;
    }
    s_a = tr16_pow2_scale(ma);
    s_d = tr16_pow2_scale(g.dz_amax ? *g.dz_amax : 0.f);
  }
  const float s_ah = g.a_scale ? 0.5f * s_a : s_a;      // staging scale of the input operand (virtual inputs arrive as 2 relu(.))

  f32x4 acc[MT][2][G::NTAP];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int k = 0; k < G::NTAP; ++k) acc[m][n][k] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float bsum[4] = {0.f, 0.f, 0.f, 0.f};

  const int tpi = g.tiles_x * g.tiles_y;
  const int total_items = g.B * tpi;
  const int per = (total_items + g.nsplit - 1) / g.nsplit;
  const int it0 = sp * per;
  const int it1 = it0 + per < total_items ? it0 + per : total_items;
  int nb = it0 / tpi, nty = (it0 - nb * tpi) / g.tiles_x, ntx = it0 - nb * tpi - nty * g.tiles_x;

  // staging slots: branch-free loads, scalar item base + 32-bit per-thread offsets (out-of-image slots read the thread's
  // own (block, quad) at the item's first pixel and are zeroed when stored)
  const int sq = tid & 3, spl = (tid >> 2) & 3;
  const int blkd = (tid >> 4) % (CO / 16), pgd = (tid >> 4) / (CO / 16);
  const int blka = (tid >> 4) % (CI / 16), pga = (tid >> 4) / (CI / 16);
  const int safed = blkd * HW * 16 + sq * 4, safea = blka * HW * 16 + sq * 4;
  int offd[G::NIT_DZ], offa[G::NIT_A], rcd[G::NIT_DZ], rca[G::NIT_A], ldsd[G::NIT_DZ], ldsa[G::NIT_A];
#pragma unroll
  for (int j = 0; j < G::NIT_DZ; ++j) {
    const int px = 4 * (pgd + j * G::NPG_DZ) + spl;          // may run past the tile in the last pass: never stored
    offd[j] = safed + ((px >> 3) * g.W + (px & 7)) * 16;
    rcd[j] = ((px >> 3) << 8) | (px & 7);
    ldsd[j] = blkd * G::DZ_BLKB + ((px >> 3) * G::PITCH + (px & 7)) * 32 + sq * 8;
  }
#pragma unroll
  for (int j = 0; j < G::NIT_A; ++j) {
    const int px = 4 * (pga + j * G::NPG_A) + spl;
    const int r = px / G::ACOLS, cx = px - r * G::ACOLS;
    offa[j] = safea + ((r + kh0 - P) * g.W + cx - P) * 16;
    rca[j] = (r << 8) | cx;
    ldsa[j] = blka * G::A_BLKB + (r * G::PITCH + cx) * 32 + sq * 8;
  }
  // the producer's BN scale / shift of this thread's channel quad, pre-multiplied by the staging scale
  // (2 relu(t) = t + |t| is positively homogeneous)
  f32x4 tsc = {0.f, 0.f, 0.f, 0.f}, tsh = {0.f, 0.f, 0.f, 0.f};
  if (g.a_scale) {
    tsc = *(const f32x4*)(g.a_scale + cib * CI + blka * 16 + sq * 4) * s_ah;
    tsh = *(const f32x4*)(g.a_shift + cib * CI + blka * 16 + sq * 4) * s_ah;
  }

  f32x4 hd[G::NIT_DZ], ha[G::NIT_A];
  unsigned okd = 0, oka = 0;

  auto load_item = [&]() {
    const int b = nb, y0 = nty * 4, x0 = ntx * 8;
    if (++ntx == g.tiles_x) {
      ntx = 0;
      if (++nty == g.tiles_y) { nty = 0; ++nb; }
    }
    const float* dzb = g.dz + (((size_t)b * dz_blocks + (dz_c0 >> 4)) * HW + y0 * g.W + x0) * 16;
    const float* ab = g.a + (((size_t)b * a_blocks + (a_c0 >> 4)) * HW + y0 * g.W + x0) * 16;
    okd = oka = 0;
#pragma unroll
    for (int j = 0; j < G::NIT_DZ; ++j) {
      const bool ok = (y0 + (rcd[j] >> 8) < g.H) & (x0 + (rcd[j] & 255) < g.W) & ((rcd[j] >> 8) < 4);
      hd[j] = *(const f32x4*)(dzb + (ok ? offd[j] : safed));
      okd |= (unsigned)ok << j;
    }
#pragma unroll
    for (int j = 0; j < G::NIT_A; ++j) {
      const int gy = y0 + (rca[j] >> 8) + kh0 - P, gx = x0 + (rca[j] & 255) - P;
      const bool ok = (gy >= 0) & (gy < g.H) & (gx >= 0) & (gx < g.W) & ((rca[j] >> 8) < G::AROWS);
      ha[j] = *(const f32x4*)(ab + (ok ? offa[j] : safea));
      oka |= (unsigned)ok << j;
    }
  };

  auto split_store = [&](f32x4 v, char* dst, int plane_stride, float mult) {
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] *= mult;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      PV4 qv;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        qv[c] = (PT)v[c];
        v[c] -= (float)qv[c];
      }
      *(PV4*)(dst + p * plane_stride) = qv;
    }
  };

  auto store_item = [&](int buf) {
    char* dzt = lds + buf * G::BUFB;
    char* at = dzt + 2 * G::DZ_PLANEB;
#pragma unroll
    for (int j = 0; j < G::NIT_DZ; ++j) {
      if (4 * (j + 1) * G::NPG_DZ <= G::DZ_PX || 4 * (pgd + j * G::NPG_DZ) < G::DZ_PX) {
        const f32x4 v = hd[j];
        const bool ok = (okd >> j) & 1;
        if (do_bias) {
          const float keep = ok ? 1.f : 0.f;
#pragma unroll
          for (int c = 0; c < 4; ++c) bsum[c] = fmaf(v[c], keep, bsum[c]);
        }
        split_store(v, dzt + ldsd[j], G::DZ_PLANEB, ok ? s_d : 0.f);
      }
    }
#pragma unroll
    for (int j = 0; j < G::NIT_A; ++j) {
      if (4 * (j + 1) * G::NPG_A <= G::A_PX || 4 * (pga + j * G::NPG_A) < G::A_PX) {
        f32x4 v = ha[j];
        const bool ok = (oka >> j) & 1;
        if (g.a_scale) {
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] = tsr_relu_x2(fmaf(v[c], tsc[c], tsh[c]));      // 2 relu(.) s_ah
          split_store(v, at + ldsa[j], G::A_PLANEB, ok ? 1.f : 0.f);
        } else {
          split_store(v, at + ldsa[j], G::A_PLANEB, ok ? s_ah : 0.f);
        }
      }
    }
  };

  // fragment addressing: lane = 16 g + 4 q + p supplies the address of pixel (row g, column c0 + q), channels 4p..4p+3 of
  // its block and receives pixels c0..c0+3 of row g for channel (lane & 15)
  const int gq = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
  const int a_lane = (wm * MT) * G::DZ_BLKB + (gq * G::PITCH + tq) * 32 + tp * 8;
  const int b_lane = (wn * 2) * G::A_BLKB + (gq * G::PITCH + tq) * 32 + tp * 8;

  auto mma_item = [&](int buf, auto late_c, auto&& between_sweeps) __attribute__((always_inline)) {
    constexpr int stage_after = decltype(late_c)::value;      // the staging block follows this sweep
    const char* dzt = lds + buf * G::BUFB;
    const char* at = dzt + 2 * G::DZ_PLANEB;
    auto load_b = [&](PV8* bf, int plane, int tap) {
      const int khl = tap / KS, kw = tap - khl * KS;
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const char* base = at + plane * G::A_PLANEB + b_lane + n * G::A_BLKB + (khl * G::PITCH + kw) * 32;
        const tv4i16 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(base));
        const tv4i16 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(base + 4 * 32));
        bf[n] = __builtin_shufflevector(__builtin_bit_cast(PV4, r0), __builtin_bit_cast(PV4, r1), 0, 1, 2, 3, 4, 5, 6, 7);
      }
    };
    auto load_a = [&](PV8* af, int plane) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const char* base = dzt + plane * G::DZ_PLANEB + a_lane + m * G::DZ_BLKB;
        const tv4i16 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(base));
        const tv4i16 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(base + 4 * 32));
        af[m] = __builtin_shufflevector(__builtin_bit_cast(PV4, r0), __builtin_bit_cast(PV4, r1), 0, 1, 2, 3, 4, 5, 6, 7);
      }
    };
    // sweeps (small terms first): dz plane 1 x a plane 0, dz 0 x a 1, dz 0 x a 0
    constexpr int SA[3] = {1, 0, 0}, SB[3] = {0, 1, 0};
    PV8 af[MT], bf[2][2];
#pragma unroll
    for (int sw = 0; sw < 3; ++sw) {
      // (prefetching sweep 2's first tap under sweep 1's last one spills at 128 x 128 x 5 taps: 3.28 -> 4.11 ms)
      if (sw != 2 || stage_after == 1) load_a(af, SA[sw]);      // sweeps 1 and 2 share dz plane 0 (re-read after a staging block)
      load_b(bf[0], SB[sw], 0);
#pragma unroll
      for (int tap = 0; tap < G::NTAP; ++tap) {
        const int cb = tap & 1;
        if (tap + 1 < G::NTAP) load_b(bf[cb ^ 1], SB[sw], tap + 1);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n)
            acc[m][n][tap] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[m], bf[cb][n], acc[m][n][tap], 0, 0, 0);
        if (tap + 1 < G::NTAP) {
          constexpr int NMF = MT * 2, PER = NMF / 4 > 0 ? NMF / 4 : 1;
          __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);       // the next tap's reads first: a whole tap of cover
          __builtin_amdgcn_sched_group_barrier(0x008, NMF, 0);
          (void)PER;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // the staging block: no fragment is live across it.  (Measured and dropped: staging the upper four waves of an
      // 8-wave tile one sweep later, so that the two waves of a SIMD convert and multiply at different times -- as a
      // wave-uniform run-time branch, without spills: 5x5 128x128 6.69 -> 6.87 ms, 3x3 128x64 2.40 -> 2.55.  The
      // position itself (after sweep 0 / 1 / 2 for all waves) is neutral or worse as well.)
      if (sw == stage_after) {
        between_sweeps();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  __syncthreads();
  if (it0 < it1) {
    load_item();
    store_item(0);
    if (it0 + 1 < it1) load_item();
  }
  __syncthreads();
  auto sweep_items = [&](auto late_c) __attribute__((always_inline)) {
    int cur = 0;
    for (int item = it0; item < it1; ++item) {
      const bool more = item + 1 < it1;
      // item + 1 is in the staging registers (requested a whole item ago); as soon as they have been converted into the
      // other LDS buffer they are reloaded with item + 2
      mma_item(cur, late_c, [&]() {
        if (more) store_item(cur ^ 1);
        if (item + 2 < it1) load_item();
      });
      __syncthreads();
      cur ^= 1;
    }
  };
  sweep_items(std::integral_constant<int, 0>());        // the staging block sits after sweep 0

  {   // this split's partial dW: slab[sp][co][ci][kh][kw]
    constexpr int T = KS * KS;
    float* sl = g.slab + (size_t)sp * g.cout * g.cin * T;
    const float inv = 1.f / (s_a * s_d);
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int ci = cib * CI + wn * 32 + n * 16 + (lane & 15);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int tap = 0; tap < G::NTAP; ++tap) {
          if (kh0 * KS + tap < T) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int co = cob * CO + wm * WM + m * 16 + 4 * gq + r;
              sl[((size_t)co * g.cin + ci) * T + kh0 * KS + tap] = acc[m][n][tap][r] * inv;
            }
          }
        }
    }
  }

  if (do_bias) {
    float* bred = (float*)lds;                 // [pixel slot = 4 pgd + spl][CO]
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) bred[(pgd * 4 + spl) * CO + blkd * 16 + sq * 4 + c] = bsum[c];
    __syncthreads();
    for (int c = threadIdx.x; c < CO; c += NT) {
      float s = 0.f;
#pragma unroll
      for (int px = 0; px < G::BRED_SLOTS; ++px) s += bred[px * CO + c];
      g.bslab[(size_t)sp * g.cout + cob * CO + c] = s;
    }
  }
}

// Tile / row-group choice per layer shape (2-plane and 1-plane forms; the 3-plane bf16x6 cross-check path keeps one
// row per workgroup).  The accumulators bound it: 32x32 wave tiles x taps x 16 registers.
//   5x5 128->128 : 128 x 128, one row  (8 waves, wave = 64 co x 32 ci, 10 tiles)     1 workgroup / CU
//   3x3 128->128 : 128 x  64, ALL rows (8 waves, wave = 32 x 32, 9 tiles)            1 workgroup / CU
//   5x5  64-> 64 :  64 x  64, two rows (4 waves, 10 tiles; row groups {0,1},{2,3},{4}) 2 workgroups / CU
//   3x3  64-> 64 :  64 x  64, ALL rows (4 waves, 9 tiles)                            2 workgroups / CU
//   1x1          : 128 x 128 / 64 x 64
template <int KS, int NS> struct WgradTCfg {
  static constexpr bool wide = NS <= 2;
  static constexpr int KHW_BIG = !wide ? 1 : (KS == 3 ? 3 : 1);
  static constexpr int CI_BIG = !wide ? 64 : (KS == 3 ? 64 : 128);
  static constexpr int WM_BIG = !wide ? 32 : (KS == 3 ? 32 : 64);
  static constexpr int KHW_SMALL = !wide ? 1 : (KS == 5 ? 2 : (KS == 3 ? 3 : 1));
};

static bool tr16_big(int cout, int cin, int ci_big) { return (cout % 128) == 0 && (cin % ci_big) == 0; }

template <int KS, int NS, bool F16, bool IO16 = false>
static int launch_tr16(const WgradTArgs& g, hipStream_t st) {
  typedef WgradTCfg<KS, NS> C;
  if constexpr (F16 && NS == 2) {            // fp16x3: the K = 32 (16x16x32 MFMA) form
    if (tr16_big(g.cout, g.cin, C::CI_BIG)) {
      typedef WgradKGeom<KS, C::KHW_BIG, 128, C::CI_BIG, C::WM_BIG> G;
      const int grid = g.nsplit * G::NKG * (g.cout / 128) * (g.cin / C::CI_BIG);
      hipLaunchKernelGGL((wgrad_k32_kernel<KS, C::KHW_BIG, 128, C::CI_BIG, C::WM_BIG>), dim3(grid), dim3(G::NT), 0, st, g);
      return tsr_check_launch();
    }
    typedef WgradKGeom<KS, C::KHW_SMALL, 64, 64, 32> G;
    const int grid = g.nsplit * G::NKG * (g.cout / 64) * (g.cin / 64);
    hipLaunchKernelGGL((wgrad_k32_kernel<KS, C::KHW_SMALL, 64, 64, 32>), dim3(grid), dim3(G::NT), 0, st, g);
    return tsr_check_launch();
  } else {                                    // bf16x6 (NS = 3) and one-plane bf16 (NS = 1, fp32 or bf16 tensors)
    if (tr16_big(g.cout, g.cin, C::CI_BIG)) {
      typedef WgradTGeom<KS, C::KHW_BIG, 128, C::CI_BIG, C::WM_BIG, NS, F16> G;
      const int grid = g.nsplit * G::NKG * (g.cout / 128) * (g.cin / C::CI_BIG);
      hipLaunchKernelGGL((wgrad_tr16_kernel<KS, C::KHW_BIG, 128, C::CI_BIG, C::WM_BIG, NS, F16, IO16>), dim3(grid), dim3(G::NT), 0, st, g);
      return tsr_check_launch();
    }
    typedef WgradTGeom<KS, C::KHW_SMALL, 64, 64, 32, NS, F16> G;
    const int grid = g.nsplit * G::NKG * (g.cout / 64) * (g.cin / 64);
    hipLaunchKernelGGL((wgrad_tr16_kernel<KS, C::KHW_SMALL, 64, 64, 32, NS, F16, IO16>), dim3(grid), dim3(G::NT), 0, st, g);
    return tsr_check_launch();
  }
}

int tsr_wgrad_b16k_1x1_wgs(int cout, int cin);          // wgrad_b16k.hip

template <int KS> static int tr16_wgs(int cout, int cin, int planes) {
  if (planes == 3) {
    typedef WgradTCfg<KS, 3> C;
    if (tr16_big(cout, cin, C::CI_BIG)) return ((KS + C::KHW_BIG - 1) / C::KHW_BIG) * (cout / 128) * (cin / C::CI_BIG);
    return ((KS + C::KHW_SMALL - 1) / C::KHW_SMALL) * (cout / 64) * (cin / 64);
  }
  typedef WgradTCfg<KS, 2> C;
  if (tr16_big(cout, cin, C::CI_BIG)) return ((KS + C::KHW_BIG - 1) / C::KHW_BIG) * (cout / 128) * (cin / C::CI_BIG);
  return ((KS + C::KHW_SMALL - 1) / C::KHW_SMALL) * (cout / 64) * (cin / 64);
}

// workgroups one batch split of this layer launches (the caller sizes nsplit so that splits x this fills the chip)
extern "C" int tsr_conv2d_wgrad_wgs_per_split(int cout, int cin, int ks, int planes) {
  if (planes == -1 && ks == 1 && tsr_wgrad_b16k_1x1_wgs(cout, cin)) return tsr_wgrad_b16k_1x1_wgs(cout, cin);      // wgrad_b16k.hip
  return ks == 1 ? tr16_wgs<1>(cout, cin, planes) : (ks == 3 ? tr16_wgs<3>(cout, cin, planes) : tr16_wgs<5>(cout, cin, planes));
}

// recommended batch splits: one resident round of workgroups (1 per CU for the 512-thread tiles, 2 for the 256-thread
// ones), never more than there are (image, patch) work items
extern "C" int tsr_conv2d_wgrad_splits(int cout, int cin, int ks, int planes, int B, int H, int W) {
  const int wgs = tsr_conv2d_wgrad_wgs_per_split(cout, cin, ks, planes);
  const int ci_big = planes == 3 ? 64 : (ks == 3 ? 64 : 128);
  const bool big = tr16_big(cout, cin, ci_big) || (planes == -1 && ks == 1 && tsr_wgrad_b16k_1x1_wgs(cout, cin));   // 8-wave tiles
  const long items = (long)B * ((H + 3) / 4) * ((W + 7) / 8);
  long ns = (big ? 256 : 512) / wgs;
  if (ns < 1) ns = 1;
  if (ns > items) ns = items;
  return (int)ns;
}

bool tsr_wgrad_b16k_ok(int cout, int cin, int ks, int H, int W, int a_ctot, int dz_ctot);            // wgrad_b16k.hip
int tsr_wgrad_b16k(const void* a, int a_ctot, int a_coff, int cin, const float* a_scale, const float* a_shift, const void* dz,
                   int dz_ctot, int dz_coff, int cout, int ks, float* slab, float* bias_slab, int nsplit, int B, int H, int W,
                   hipStream_t st);

int tsr_conv2d_wgrad_tr16(const float* a, int a_ctot, int a_coff, int cin, const float* a_scale, const float* a_shift,
                          const float* dz, int dz_ctot, int dz_coff, int cout, int ks, int planes,
                          const float* a_amax, const float* dz_amax, float* slab, float* bias_slab, int nsplit,
                          int B, int H, int W, hipStream_t st) {
  WgradTArgs g;
  g.a = a; g.a_ctot = a_ctot; g.a_coff = a_coff; g.cin = cin; g.a_scale = a_scale; g.a_shift = a_shift;
  g.dz = dz; g.dz_ctot = dz_ctot; g.dz_coff = dz_coff; g.cout = cout;
  g.slab = slab; g.bslab = bias_slab; g.B = B; g.H = H; g.W = W; g.nsplit = nsplit;
  g.tiles_x = (W + 7) / 8; g.tiles_y = (H + 3) / 4;
  g.a_amax = a_amax; g.dz_amax = dz_amax;
  if (planes == 3) {
    if (ks == 1) return launch_tr16<1, 3, false>(g, st);
    if (ks == 3) return launch_tr16<3, 3, false>(g, st);
    return launch_tr16<5, 3, false>(g, st);
  }
  if (planes == 1) {
    if (ks == 1) return launch_tr16<1, 1, false>(g, st);
    if (ks == 3) return launch_tr16<3, 1, false>(g, st);
    return launch_tr16<5, 1, false>(g, st);
  }
  if (planes == -1) {       // bf16 tensors (activation storage of the "bf16" configurations), one bf16 plane
    // no fused input transform: nothing to compute while staging -- the LDS-DMA kernel (same tiles, splits and slabs)
    // (the 1x1 `confusion` too: its input transform runs in LDS behind the DMA)
    if ((!a_scale || ks == 1) && tsr_wgrad_b16k_ok(cout, cin, ks, H, W, a_ctot, dz_ctot))
      return tsr_wgrad_b16k(a, a_ctot, a_coff, cin, a_scale, a_shift, dz, dz_ctot, dz_coff, cout, ks, slab, bias_slab, nsplit, B,
                            H, W, st);
    if (ks == 1) return launch_tr16<1, 1, false, true>(g, st);
    if (ks == 3) return launch_tr16<3, 1, false, true>(g, st);
    return launch_tr16<5, 1, false, true>(g, st);
  }
  if (ks == 1) return launch_tr16<1, 2, true>(g, st);
  if (ks == 3) return launch_tr16<3, 2, true>(g, st);
  return launch_tr16<5, 2, true>(g, st);
}

// C-ABI entry of the 16-bit-MFMA weight gradient (every `planes` form lands in launch_tr16 above)
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
  return tsr_conv2d_wgrad_tr16(a, a_ctot, a_coff, cin, a_scale, a_shift, dz, dz_ctot, dz_coff, cout, ks, planes, a_amax,
                               dz_amax, slab, bias_slab, nsplit, B, H, W, (hipStream_t)stream);
}

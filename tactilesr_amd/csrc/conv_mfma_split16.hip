// Split-operand implicit-GEMM convolution on the CDNA4 16-bit matrix cores (v_mfma_f32_32x32x16_{bf16,f16}),
// a drop-in alternative to conv_mfma_f32_kernel (same tiling, same epilogues).
//
// gfx950 runs 16-bit MFMA at 16x the fp32-MFMA rate, so fp32 operands are split on the fly into NS 16-bit planes
// and each K = 16 step issues only the cross products that matter, all accumulated in fp32 inside the MFMA:
//   fp16, NS = 2 ("fp16x3", default eval path): operands are first scaled by exact powers of two into fp16's
//        range (weights at pack time; activations from the max|x| scalar their producer published), then
//        x*sx = h1+h2, w*sw = g1+g2 (2 x 11 significand bits; round-to-nearest residuals make that 23 bits) and
//        h1g1 + h1g2 + h2g1; the scales are undone exactly in the epilogue.  3/16 of the fp32-MFMA cost; measured
//        network-level error vs fp64 <= the fp32-MFMA path's on every fixture.
//   bf16, NS = 3 ("bf16x6"): x = x1+x2+x3 (24 bits, fp32's exponent range, no scaling), six products
//        x1w1 + x1w2 + x2w1 + x1w3 + x3w1 + x2w2: error <= the fp32 MFMA path's (1.4e-7 vs 2.9e-7 at K = 3200).
//        Used for training (gradients span too wide a range for fp16 planes without per-tensor bookkeeping).
//   bf16, NS = 2 ("bf16x3", ~4e-6 per layer) and NS = 1 (plain bf16): reduced-precision modes, never the parity path.
// Activations stay fp32 CB16 in HBM (the path is MFMA-bound, not HBM-bound); the split happens while the halo
// slab is staged into LDS, weights are split once by tsr_pack_conv_weight_{bf16s,f16s}.
//
// Tiling is the fp32 kernel's: 8x8 patch x 2 images (M = 128) x all C_out per 256-thread workgroup, waves
// 2 (image) x 2 (C_out half), per C_in block of 16 (= one MFMA K step) the halo slab sits in LDS as
// [pixel][plane][16 x 16-bit] with a 112/64/48-B pixel stride and a row stride chosen so that the ds_read_b128
// A fragments of all 64 lanes are bank-conflict free; fragments are ping-pong prefetched one tap ahead and
// interleaved with the MFMAs; per step the [plane][2][C_out][8] weight slab goes through a 3-slot LDS ring.
#include "tsr_common.h"
#include "conv_args.h"
#include "conv_epilogue.h"
#include "conv_fuse1x1.h"
#include "tactilesr_hip.h"
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// 16-bit plane element: bf16 (fp32 exponent range, 8-bit significand: 3 planes = fp32) or fp16 (11-bit
// significand: 2 planes = 22+ bits, the operands are pre-scaled by a power of two into fp16's range).
template <bool F16> struct Plane;
template <> struct Plane<false> {
  typedef __bf16 T; typedef bf16x8 V8; typedef bf16x4 V4;
  static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Plane<true> {
  typedef _Float16 T; typedef f16x8 V8; typedef f16x4 V4;
  static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

template <int NS> struct SplitGeom;
template <> struct SplitGeom<1> { static constexpr int PIXS = 3, RMOD = 8; };
template <> struct SplitGeom<2> { static constexpr int PIXS = 4, RMOD = 1; };
template <> struct SplitGeom<3> { static constexpr int PIXS = 7, RMOD = 8; };

// taps per barrier step: enough MFMA work between two barriers (>= 24 MFMAs per wave where LDS allows)
__host__ __device__ constexpr int taps_per_step(int ks, int cout, int ns) {
  const int t = ks * ks;
  // measured: ns = 3 is fastest at 1 (occupancy), 2 and 1 gain from grouping; one plane x 128 channels: 2 taps keep
  // the 4-image workgroups of the bf16-storage path at two per CU
  // (one plane x 128 channels x 3x3: 3 taps = a kernel row per step -- 9 taps in 3 steps with no zero-padded tap slot,
  //  where 2 per step cost a fifth step with one wasted tap; 61.6 KB of LDS, still two workgroups per CU: bf16-storage eval
  //  14.3 -> 13.1 ms per six launches, same-box A/B.  The 64-channel one-plane 3x3 stays at 3: a whole block per step (9)
  //  needs 79 KB and loses the third resident workgroup per CU, 6.7 -> 7.9 ms.  5x5 x 128 channels: 3 taps per step too --
  //  9 steps with two padded tap slots (7 % more MFMAs) still beat 13 steps of 2: 24.5 -> 23.7 ms per six launches; a whole
  //  kernel row per step (5, no padding) would need a 60 KB ring + the 30.7 KB halo: one workgroup per CU)
  const int want = ns == 3 ? 1 : (ns == 2 ? 2 : (cout == 128 ? 3 : ks));
  return want < t ? want : t;
}

constexpr int row_slots(int hh, int pixs, int rmod) {
  int rs = hh * pixs;
  while ((rs & 15) != rmod) ++rs;
  return rs;
}

// WN = waves across C_out: 2 -> 2 images per workgroup, wave = (image, C_out half); 1 -> 4 images per workgroup, wave
// = image x all C_out (used for C_out = 64, where a half would leave a single 32-wide block per wave).
// DBH = double-buffered halo (3x3 kernels, C_in/16 even): the next channel block's slab is staged into the other
// halo buffer in the middle of the current block (loads at its first step, conversion + LDS writes at its
// second-to-last), so a block boundary costs no extra barrier pair and the first tap's fragments are prefetched
// like any other tap's.  Blocks are processed in pairs: buffer and fragment-set parity stay compile-time.
// IO16 (NS = 1 only, inference): activations live in HBM as bf16 CB16 -- the halo slab is COPIED into LDS (8 B per
// (pixel, channel quad), no conversion), the epilogue rounds to bf16 on store.  BASELINE's "bf16" configurations.
// FUSE2 (fp16x3, C_out = 128, inference): the MSRB's 1x1 `confusion` half is applied to the tile before it leaves the
// workgroup (conv_fuse1x1.h); out / res then describe the 64-channel result.
// PAIR (bf16 storage, 5x5 geometry, 128 output channels, inference): the two stage-1 convolutions of an MSRB -- 3x3 64->64
// and 5x5 64->64 on the SAME input (model/tactileSR_model.py:167-175,198-200) -- as one launch on one staged halo: a 5x5
// conv to 128 channels whose first 64 (the 3x3 conv: C_out blocks 0, 1 of every wave) have no weight on the 16 outer taps;
// on those taps neither their weight fragments are read nor their MFMAs issued.  `cat1` comes out in torch.cat order.
template <int KS, int COUT, int NS, bool EXT, bool F16, int WN, bool DBH = false, bool IO16 = false, bool FUSE2 = false,
          bool PAIR = false>
__global__ __launch_bounds__(256, 2) void conv_mfma_split16_kernel(const ConvArgs a) {
  static_assert(!PAIR, "the bf16-storage stage-1 pair runs csrc/conv_b16k.hip");
  static_assert(!IO16 || (NS == 1 && !F16), "bf16 activation storage: plain bf16 operands");
  static_assert(!FUSE2 || (!EXT && COUT == 128 && NS == 2 && F16 && WN == 2 && !IO16),
                "fused 1x1: 128 channels, inference, fp16x3 (2 images / workgroup); bf16 storage: csrc/conv_b16k.hip");
  typedef typename Plane<F16>::T PT;
  typedef typename Plane<F16>::V8 PV8;
  typedef typename Plane<F16>::V4 PV4;
  constexpr int IMG = 4 / WN;
  constexpr int P = KS / 2;
  constexpr int HH = 8 + KS - 1;
  constexpr int T = KS * KS;
  constexpr int NB = COUT / (32 * WN);
  constexpr int PIXB = SplitGeom<NS>::PIXS * 16;                               // bytes per halo pixel
  constexpr int ROWB = row_slots(HH, SplitGeom<NS>::PIXS, SplitGeom<NS>::RMOD) * 16;
  constexpr int IMGB = HH * ROWB;
  constexpr int HALO_B = IMG * IMGB;
  constexpr int TPS = taps_per_step(KS, COUT, NS);
  constexpr int NSTEP = (T + TPS - 1) / TPS;                                   // barrier steps per channel block
  constexpr int WTAP_B = NS * 16 * COUT * 2;                                   // bytes per (block, tap)
  constexpr int WSLAB_B = TPS * WTAP_B;                                        // bytes per step
  constexpr int WITEMS = WSLAB_B / 16;
  constexpr int WV = (WITEMS + 255) / 256;
  constexpr int NITEM = IMG * HH * HH * 4;
  constexpr int NIT = (NITEM + 255) / 256;
  constexpr int NPROD = NS == 3 ? 6 : (NS == 2 ? 3 : 1);
  // products ordered small -> large so the fp32 accumulator sees the low-order terms first
  constexpr int PA[6] = {NS == 3 ? 2 : (NS == 2 ? 1 : 0), NS == 3 ? 0 : 0, NS == 3 ? 1 : 0, 1, 0, 0};
  constexpr int PB[6] = {0, NS == 3 ? 2 : (NS == 2 ? 1 : 0), NS == 3 ? 1 : 0, 0, 1, 0};

  static_assert(!DBH || (NSTEP >= 3 && (T & 1)), "double-buffered halo needs >= 3 steps per block and an odd tap count");
  constexpr int NHB = DBH ? 2 : 1;
  constexpr int MAIN_LDS = NHB * HALO_B + 3 * WSLAB_B;
  constexpr int FUSE_LDS = !FUSE2 ? 0 : Fuse1x1Geom::BYTES + 64;
  __shared__ __attribute__((aligned(16))) char lds[MAIN_LDS > FUSE_LDS ? MAIN_LDS : FUSE_LDS];
  char* halo = lds;
  char* wbuf = lds + NHB * HALO_B;  // 3-slot ring: slab s lives in slot s % 3

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = WN == 2 ? wave >> 1 : wave, wn = WN == 2 ? wave & 1 : 0;
  const int h = lane >> 5, li = lane & 31;

  // fp16 planes: power-of-two input scale from the producer's max|x| (m*sx in [2^13, 2^14)), undone exactly in
  // the epilogue together with the pack-time weight scale
  float sx = 1.f, accmul = 1.f;
  if (F16) {
    float m = a.in_amax ? *a.in_amax : 0.f;
    if (EXT && a.in_scale) {
      // the input is virtual, relu(x*s_c + t_c): bound its magnitude by max|x| * max|s_c| + max|t_c|
      __shared__ float bnd[8];
      float ms = 0.f, mt = 0.f;
      for (int c = threadIdx.x; c < a.cin; c += 256) {
        ms = fmaxf(ms, fabsf(a.in_scale[c]));
        mt = fmaxf(mt, fabsf(a.in_shift[c]));
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        ms = fmaxf(ms, __shfl_xor(ms, o));
        mt = fmaxf(mt, __shfl_xor(mt, o));
      }
      if ((threadIdx.x & 63) == 0) { bnd[(threadIdx.x >> 6) * 2] = ms; bnd[(threadIdx.x >> 6) * 2 + 1] = mt; }
      __syncthreads();
      ms = fmaxf(fmaxf(bnd[0], bnd[2]), fmaxf(bnd[4], bnd[6]));
      mt = fmaxf(fmaxf(bnd[1], bnd[3]), fmaxf(bnd[5], bnd[7]));
      m = m * ms + mt;
    }
    if (m > 0.f) {
      int e = (int)((__float_as_uint(m) >> 23) & 0xFF) - 127;
      int be = 13 - e + 127;
      be = be < 1 ? 1 : (be > 254 ? 254 : be);
      sx = __uint_as_float((unsigned)be << 23);
    }
    float w_inv = a.w_inv_scale;
    if (a.w_amax) {            // same power of two the device-side pack derived from the same scalar
      const float wm = *a.w_amax;
      w_inv = 1.f;
      if (wm > 0.f && wm < 3.0e38f) {
        int be = 127 - (13 - ((int)((__float_as_uint(wm) >> 23) & 0xFF) - 127));
        be = be < 1 ? 1 : (be > 254 ? 254 : be);
        w_inv = __uint_as_float((unsigned)be << 23);
      }
    }
    accmul = w_inv / sx;
  }

  int bid;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tpi = a.tiles_x * a.tiles_y;
  const int ig = bid / tpi;
  const int trem = bid - ig * tpi;
  const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
  const int y0 = ty * 8, x0 = tx * 8, b0 = ig * IMG;
  const int HW = a.H * a.W;
  const int in_blocks = a.in_ctot >> 4;

  int st_src[NIT], st_dst[NIT];
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    const int it = tid + k * 256;
    st_src[k] = -1;
    st_dst[k] = -1;
    if (it < NITEM) {
      const int qd = it & 3, px = it >> 2;
      const int img = px / (HH * HH), rem = px - img * (HH * HH);
      const int hy = rem / HH, hx = rem - hy * HH;
      const int gy = y0 - P + hy, gx = x0 - P + hx, b = b0 + img;
      st_dst[k] = img * IMGB + hy * ROWB + hx * PIXB + qd * 8;
      if (b < a.B && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
        st_src[k] = ((img * in_blocks) * HW + gy * a.W + gx) * 16 + qd * 4;
    }
  }
  const size_t in_base_idx = ((size_t)b0 * in_blocks + (a.in_coff >> 4)) * HW * 16;
  const float* in_base = a.in + in_base_idx;
  const __bf16* in_base16 = (const __bf16*)a.in + in_base_idx;

  const int laneA = wm * IMGB + (li >> 3) * ROWB + (li & 7) * PIXB + h * 16;
  const int laneB = (h * COUT + wn * (COUT / WN) + li) * 16;

  f32x16 acc[2][NB];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;

  const int nchunk = a.cin >> 4;
  const int S = nchunk * NSTEP;
  const char* wsrc = (const char*)a.wp;
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, 0x7fffffff, 0x00020000);
  const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- helpers (all loops fully unrolled: fragment registers are plain SSA values)
  // BN scale / shift of this thread's channel quad of the block in flight (every staging item of a thread carries the
  // same quad): requested WITH the slab, not at the block boundary where the conversion would wait for them
  f32x4 hsc = {1.f, 1.f, 1.f, 1.f}, hsh = {0.f, 0.f, 0.f, 0.f};
  auto load_halo = [&](int c, f32x4* hv) {
    if (EXT && a.in_scale) {
      hsc = *(const f32x4*)(a.in_scale + c * 16 + (tid & 3) * 4);
      hsh = *(const f32x4*)(a.in_shift + c * 16 + (tid & 3) * 4);
    }
    if (IO16) {     // 4 bf16 = 8 B per item, carried in the low half of the f32x4 slot
      const __bf16* inc = in_base16 + (size_t)c * HW * 16;
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        hv[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (st_src[k] >= 0) {
          const float2 t = *(const float2*)(inc + st_src[k]);
          hv[k][0] = t.x;
          hv[k][1] = t.y;
        }
      }
      return;
    }
    const float* inc = in_base + (size_t)c * HW * 16;
    // (a branch-free form -- dummy address + zero scale, as in conv_mfma_k32.hip -- measured 1-7 % slower here: every
    // address is live at once and the one-plane variants start to spill)
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      hv[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (st_src[k] >= 0) hv[k] = *(const f32x4*)(inc + st_src[k]);
    }
  };
  auto store_halo = [&](const f32x4* hv, int c, int hb) {
    if (IO16) {
#pragma unroll
      for (int k = 0; k < NIT; ++k)
        if (st_dst[k] >= 0) {
          if (EXT && a.in_scale && st_src[k] >= 0) {
            // producer's train-mode BN + ReLU on the stored bf16 pre-activation, fp32 arithmetic, rounded to the bf16
            // MFMA operand (training with bf16 activation storage)
            const bf16x4 zq = __builtin_bit_cast(bf16x4, make_float2(hv[k][0], hv[k][1]));
            bf16x4 aq;
#pragma unroll
            for (int j = 0; j < 4; ++j) aq[j] = (__bf16)tsr_relu(fmaf((float)zq[j], hsc[j], hsh[j]));
            *(bf16x4*)(halo + hb * HALO_B + st_dst[k]) = aq;
          } else {
            *(float2*)(halo + hb * HALO_B + st_dst[k]) = make_float2(hv[k][0], hv[k][1]);
          }
        }
      return;
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      if (st_dst[k] >= 0) {
        f32x4 v = hv[k];
        if (EXT && a.in_scale && st_src[k] >= 0) {   // producer's train-mode BN+ReLU, fused into the load
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = tsr_relu(fmaf(v[j], hsc[j], hsh[j]));
        }
        if (F16) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] *= sx;
        }
#pragma unroll
        for (int p = 0; p < NS; ++p) {
          PV4 bq;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            bq[j] = (PT)v[j];
            v[j] -= (float)bq[j];
          }
          *(PV4*)(halo + hb * HALO_B + st_dst[k] + p * 32) = bq;
        }
      }
    }
  };
  // W ring helpers; the (v+1)*256 <= WITEMS test folds the bounds check away for full passes
#define LOAD_W(sidx)                                                                     \
  {                                                                                      \
    const f32x4* src_ = (const f32x4*)(wsrc + (size_t)(sidx) * WSLAB_B);                  \
    _Pragma("unroll") for (int v = 0; v < WV; ++v)                                       \
      if ((v + 1) * 256 <= WITEMS || tid + v * 256 < WITEMS) wreg[v] = src_[tid + v * 256]; \
  }
#define STORE_W(slot)                                                                    \
  {                                                                                      \
    char* wb_ = wbuf + (slot) * WSLAB_B;                                                 \
    _Pragma("unroll") for (int v = 0; v < WV; ++v)                                       \
      if ((v + 1) * 256 <= WITEMS || tid + v * 256 < WITEMS) ((f32x4*)wb_)[tid + v * 256] = wreg[v]; \
  }
  // Inference launches (!EXT): the weight slab goes global -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR hop, no
  // ds_write; see conv_mfma_k32.hip): slab s+2 is requested at the start of step s into ring slot (s+2)%3 and awaited before
  // the barrier that ends the step.  Training launches keep the register staging above (measured faster there).
  // (MUBUF form -- buffer_load_dwordx4 ... lds -- not global_load_lds: hipcc treats the FLAT-encoded instruction as a
  // possible LDS access through FLAT and then turns every later counted lgkmcnt wait of the step into lgkmcnt(0))
#define DMA_W(sidx, slot)                                                                \
  {                                                                                      \
    const int vo_ = (sidx) * WSLAB_B + tid * 16;                                         \
    char* dst_ = wbuf + (slot) * WSLAB_B + wave_s * 1024;                                \
    _Pragma("unroll") for (int v = 0; v < WV; ++v)                                       \
      if ((v + 1) * 256 <= WITEMS || tid + v * 256 < WITEMS)                             \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)(dst_ + v * 4096), 16, \
                                                 vo_ + v * 4096, 0, 0, 0);                \
  }
  // vmcnt wait that leaves the n_ youngest vector-memory operations (the next block's halo loads) in flight
#define DMA_WAIT_N(n_) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n_) & 15) | (((n_) >> 4) << 14))
  // fragments of one (block, tap) step -> register set `set` (compile-time index after unrolling)
#define LOAD_FRAGS(set, slot, tapoff, kh_, kw_, hb_)                                     \
  {                                                                                      \
    const char* wb_ = wbuf + (slot) * WSLAB_B + (tapoff) * WTAP_B;                       \
    _Pragma("unroll") for (int p = 0; p < NS; ++p) {                                     \
      _Pragma("unroll") for (int mb = 0; mb < 2; ++mb)                                   \
        fa[set][p][mb] = *(const PV8*)(halo + (hb_) * HALO_B + laneA + (4 * mb + (kh_)) * ROWB + (kw_) * PIXB + p * 32); \
      _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)                                  \
        if (!PAIR || nb >= 2 || ((kh_) >= 1 && (kh_) <= 3 && (kw_) >= 1 && (kw_) <= 3))  \
          fb[set][p][nb] = *(const PV8*)(wb_ + laneB + p * (2 * COUT * 16) + nb * (32 * 16)); \
    }                                                                                    \
  }

  // ---- prologue: halo(0), W(0), W(1) in LDS; W(2) in flight
  // (LDS-DMA in the fp32-storage training instantiations: neutral here, -5 ms/step in the K = 32 kernel; the one-plane
  // bf16-storage training form moves 3x the weight bytes per MFMA and gains from it like its inference form does)
  constexpr bool WDMA = !EXT || IO16;
  f32x4 hv[NIT], wreg[WDMA ? 1 : WV];
  load_halo(0, hv);
  if constexpr (WDMA) {
    DMA_W(0, 0);
    if (S > 1) DMA_W(1, 1);
    store_halo(hv, 0, 0);
    DMA_WAIT_N(0);
  } else {
    LOAD_W(0);
    STORE_W(0);
    if (S > 1) { LOAD_W(1); STORE_W(1); }
    store_halo(hv, 0, 0);
    if (S > 2) LOAD_W(2);
  }
  __syncthreads();

  PV8 fa[2][NS][2], fb[2][NS][NB];        // ping-pong fragment sets, statically indexed
  LOAD_FRAGS(0, 0, 0, 0, 0, 0);

  int s = 0;
  int slot = 0;                            // s % 3, kept incrementally
  // one channel block; P = parity of the block (compile time in DBH mode: halo buffer and first fragment set)
  auto block = [&](int c, auto parity) {
    constexpr int P = decltype(parity)::value;
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
      const int slot1 = slot == 2 ? 0 : slot + 1;
      const int slot2 = slot1 == 2 ? 0 : slot1 + 1;
      if (WDMA && s + 2 < S) DMA_W(s + 2, slot2);                   // slot (s+2)%3 was last read one barrier ago
      if (DBH && st == 0 && c + 1 < nchunk) load_halo(c + 1, hv);   // next block's slab: in flight for NSTEP-2 steps
#pragma unroll
      for (int tt = 0; tt < TPS; ++tt) {
        const int t = st * TPS + tt;
        if (t < T) {
          const int cur = (t + P * T) & 1, nxt = cur ^ 1;
          constexpr bool PF = true;
          if (t + 1 < T) {        // next tap's fragments: its weights were published by an earlier barrier
            const int nkh = (t + 1) / KS, nkw = (t + 1) - nkh * KS;
            if (tt + 1 < TPS) { LOAD_FRAGS(nxt, slot, tt + 1, nkh, nkw, P); }
            else { LOAD_FRAGS(nxt, slot1, 0, nkh, nkw, P); }
          } else if (DBH) {
            // first tap of the next block out of the other halo buffer (published two barriers ago; after the
            // last block this reads stale but mapped LDS and the values are dropped)
            LOAD_FRAGS(nxt, slot1, 0, 0, 0, P ^ 1);
          } else if (c + 1 < nchunk) {
            load_halo(c + 1, hv); // next block's slab: global loads fly under this tap's MFMAs
          }
          // pair form: the 3x3 conv (C_out blocks 0, 1) has no weight outside the inner 3x3 taps
          const int ckh = t / KS, ckw = t - ckh * KS;
          const bool inner = !PAIR || (ckh >= 1 && ckh <= 3 && ckw >= 1 && ckw <= 3);
          const int nkh1 = (t + 1) / KS, nkw1 = (t + 1) - nkh1 * KS;
          const bool inner_next = !PAIR || t + 1 >= T || (nkh1 >= 1 && nkh1 <= 3 && nkw1 >= 1 && nkw1 <= 3);
#pragma unroll
          for (int q = 0; q < NPROD; ++q)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
              for (int nb = 0; nb < NB; ++nb)
                if (inner || nb >= 2)
                  acc[mb][nb] = Plane<F16>::mfma(fa[cur][PA[6 - NPROD + q]][mb], fb[cur][PB[6 - NPROD + q]][nb],
                                                 acc[mb][nb]);
          if (PF && (t + 1 < T || DBH)) {
            // interleave the next tap's fragment reads with this tap's MFMAs (hipcc otherwise sinks all ds_reads
            // below the MFMA block, exposing their latency in front of the barrier's lgkmcnt(0) every step)
            constexpr int MFX = 1;
            // (compile-time group sizes: the pair form's half-work taps simply leave some groups unfilled)
            constexpr int NRD = NS * (2 + NB), NMF = NPROD * 2 * NB, PER = PAIR ? 1 : (NMF / NRD > 0 ? NMF / NRD : 1) * MFX;
            (void)inner_next;
#pragma unroll
            for (int i = 0; i < NRD; ++i) {
              __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);   // MFMA
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // DS read
            }
          }
        }
      }
      if (DBH && st == NSTEP - 2 && c + 1 < nchunk) store_halo(hv, c + 1, P ^ 1);   // other buffer: last read a block ago
      if constexpr (WDMA) {
        // (the step that issued the next block's halo loads lets them fly on)
        if ((DBH ? st == 0 : st + 1 == NSTEP) && c + 1 < nchunk) { DMA_WAIT_N(NIT); } else { DMA_WAIT_N(0); }
      } else {
        if (s + 2 < S) STORE_W(slot2);           // slot (s+2)%3 was last read one barrier ago
        if (s + 3 < S) LOAD_W(s + 3);
      }
      __syncthreads();
      if (!DBH && st + 1 == NSTEP && c + 1 < nchunk) {
        store_halo(hv, c + 1, 0);  // every wave is past its last read of the old slab (barrier above)
        __syncthreads();
        LOAD_FRAGS(0, slot1, 0, 0, 0, 0);      // T is odd: a block always starts on fragment set 0
      }
      ++s;
      slot = slot1;
    }
  };
  if (DBH) {
    for (int c = 0; c < nchunk; c += 2) {      // launch guarantees an even block count
      block(c, std::integral_constant<int, 0>());
      block(c + 1, std::integral_constant<int, 1>());
    }
  } else {
    for (int c = 0; c < nchunk; ++c) block(c, std::integral_constant<int, 0>());
  }
#undef LOAD_W
#undef STORE_W
#undef DMA_W
#undef DMA_WAIT_N
#undef LOAD_FRAGS

  if constexpr (FUSE2) conv_fuse1x1_epilogue(a, acc, lds, b0, y0, x0, wm, wn, h, li, HW, accmul);
  else conv_epilogue<COUT, EXT, WN, IO16>(a, acc, bid, b0, y0, x0, wm, wn, h, li, HW, accmul);
}

// OIHW fp32 -> [C_in/16][step][tap in step][plane][2 (k half)][C_out][8] bf16 split planes; taps are
// grouped TPS per barrier step (taps_per_step), the tail of the last step is zero.
// dgrad mode (ci0 >= 0): the packed conv is W'[n][k=co][kh][kw] = W[co][ci0+n][K-1-kh][K-1-kw] with
// "cout" := nprime and "cin" := cout_f (see tsr_pack_conv_weight_dgrad).
template <bool F16>
__global__ void pack_conv_weight_bf16s_kernel(const float* __restrict__ w, typename Plane<F16>::T* __restrict__ wp,
                                              int cout, int cin, int ks, int ns, int tps, int ci0, int cin_f,
                                              float wscale, const float* __restrict__ w_amax, int k32) {
  if (w_amax) {                // device-side scale: 2^(13 - floor(log2 max|w|)), 1 for a zero / non-finite maximum
    const float wm = *w_amax;
    wscale = 1.f;
    if (wm > 0.f && wm < 3.0e38f) {
      int be = 127 + 13 - ((int)((__float_as_uint(wm) >> 23) & 0xFF) - 127);
      be = be < 1 ? 1 : (be > 254 ? 254 : be);
      wscale = __uint_as_float((unsigned)be << 23);
    }
  }
  typedef typename Plane<F16>::T PT;
  const int T = ks * ks;
  const int nstep = (T + tps - 1) / tps;
  const int TP = nstep * tps;                          // padded tap count
  const int cin_p = k32 ? ((cin + 31) & ~31) : cin;    // k32: channel blocks in pairs (the phantom block is all zero)
  const size_t total = (size_t)cout * cin_p * TP;      // one thread per (padded) fp32 weight -> ns outputs
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int j = i & 7;
    size_t r = i >> 3;
    const int n = r % cout; r /= cout;
    const int kh = r & 1; r >>= 1;
    const int tap = r % TP;
    const int chunk = r / TP;
    const int ci = chunk * 16 + kh * 8 + j;
    float v = 0.f;
    if (tap < T && ci < cin) v = ci0 < 0 ? w[((size_t)n * cin + ci) * T + tap]
                                         : w[((size_t)ci * cin_f + ci0 + n) * T + (T - 1 - tap)];
    v *= wscale;
    // position of this (block, tap) slab in the stream.  k32 = 1 (conv_mfma_k32.hip; tps = 1, blocks in pairs): the even
    // block's taps 0..T-2, the odd block's taps 0..T-2, then the cross pair (last tap of the even, of the odd block)
    size_t pos = (size_t)chunk * TP + tap;
    if (k32 == 1) {
      const int odd = chunk & 1;
      pos = (size_t)(chunk >> 1) * 2 * T + (tap < T - 1 ? odd * (T - 1) + tap : 2 * T - 2 + odd);
    } else if (k32 == 2) {     // 3x3 (double-buffered halo): even block taps 0..T-2, cross pair, odd block taps 0..T-2
      const int odd = chunk & 1;
      pos = (size_t)(chunk >> 1) * 2 * T + (tap < T - 1 ? odd * (T + 1) + tap : T - 1 + odd);
    }
    for (int p = 0; p < ns; ++p) {
      const PT bq = (PT)v;
      v -= (float)bq;
      wp[(((pos * ns + p) * 2 + kh) * cout + n) * 8 + j] = bq;
    }
  }
}

// fp16x3 3x3 / 5x5 convolutions run on conv_mfma_k32.hip (16x16x32 MFMA, tap pairs); the 1x1 layers and the other
// arithmetic modes stay on the 32x32x16 kernel of this file.  Pack and launch ask the same question (use_k32).
int tsr_conv_k32(const ConvArgs& a, int cout, int ks, bool ext, hipStream_t st);      // conv_mfma_k32.hip
int tsr_conv_k32_fuse1x1(const ConvArgs& a, int ks, hipStream_t st);
// C_out = 64 convs run the K = 32 kernel too since round 3 (4 images x all 64 channels per workgroup): 3x3 eval 1.62 ->
// 1.46 ms per launch at B = 4096, train (B = 2048) 3x3 dgrad 1.02 -> 0.96 ms, 5x5 forward 1.61 -> 1.45, 5x5 dgrad
// 1.74 -> 1.63.  In round 2 the training instantiation of the 5x5 form ran 3x slower: one of the kernel's lambdas was
// not inlined there, so its by-reference captures (the argument block, the staging index arrays) lived in scratch;
// the lambdas are always_inline now.
static bool use_k32(int ks, int kdim, int cout) {
  (void)kdim;          // an odd channel-block count is padded with a zero-weight block (pack) / a re-read block (kernel)
  return ks > 1 && (cout == 128 || cout == 64);
}
// images per workgroup (= statistics-slab entries per workgroup) of the fp16x3 kernel that runs (cout, ks)
int tsr_conv_k32_images(int cout);            // conv_mfma_k32.hip
int tsr_conv_f16s_images(int cout, int ks) {
  if (use_k32(ks, 0, cout)) return tsr_conv_k32_images(cout);
  return (ks > 1 && cout == 64) ? 4 : 2;
}
static int k32_mode(int ks, int kdim, int cout) { return use_k32(ks, kdim, cout) ? (ks == 3 ? 2 : 1) : 0; }

// bf16 elements a packed weight needs (taps padded to a multiple of the step size)
extern "C" long long tsr_conv_weight_bf16s_elems(int cout, int cin, int ks, int nsplit) {
  const int tps = taps_per_step(ks, cout, nsplit);
  const int T = ks * ks;
  const long long a = (long long)nsplit * cout * cin * (((T + tps - 1) / tps) * tps);
  const long long b = (long long)nsplit * cout * ((cin + 31) & ~31) * T;      // K = 32 kernel: channel blocks in pairs
  return a > b ? a : b;
}

extern "C" int tsr_pack_conv_weight_bf16s(const float* w_oihw, void* w_packed, int cout, int cin, int ks,
                                          int nsplit, void* stream) {
  if (!w_oihw || !w_packed || (cin & 15) || (cout != 64 && cout != 128) || (ks != 1 && ks != 3 && ks != 5) ||
      nsplit < 1 || nsplit > 3)
    return TSR_ERR_ARG;
  const int tps = taps_per_step(ks, cout, nsplit);
  const size_t total = (size_t)cout * cin * (((ks * ks + tps - 1) / tps) * tps);
  const int grid = (int)((total + 255) / 256);
  hipLaunchKernelGGL(pack_conv_weight_bf16s_kernel<false>, dim3(grid > 4096 ? 4096 : grid), dim3(256), 0,
                     (hipStream_t)stream, w_oihw, (__bf16*)w_packed, cout, cin, ks, nsplit, tps, -1, 0, 1.0f, nullptr, 0);
  return tsr_check_launch();
}

// fp16 two-plane packing: planes of w*wscale (wscale a power of two chosen by the caller so that
// max|w|*wscale lies in [2^13, 2^14)); tsr_conv2d_fwd_f16s gets 1/wscale back.
extern "C" int tsr_pack_conv_weight_f16s(const float* w_oihw, void* w_packed, int cout, int cin, int ks,
                                         float wscale, void* stream) {
  if (!w_oihw || !w_packed || (cin & 15) || (cout != 64 && cout != 128) || (ks != 1 && ks != 3 && ks != 5) ||
      !(wscale > 0.f))
    return TSR_ERR_ARG;
  const int k32 = k32_mode(ks, cin, cout);
  const int tps = k32 ? 1 : taps_per_step(ks, cout, 2);
  const size_t total = (size_t)cout * cin * (((ks * ks + tps - 1) / tps) * tps);
  const int grid = (int)((total + 255) / 256);
  hipLaunchKernelGGL(pack_conv_weight_bf16s_kernel<true>, dim3(grid > 4096 ? 4096 : grid), dim3(256), 0,
                     (hipStream_t)stream, w_oihw, (_Float16*)w_packed, cout, cin, ks, 2, tps, -1, 0, wscale, nullptr, k32);
  return tsr_check_launch();
}

extern "C" int tsr_pack_conv_weight_f16s_dev(const float* w_oihw, void* w_packed, int cout, int cin, int ks,
                                             const float* w_amax, void* stream) {
  if (!w_oihw || !w_packed || !w_amax || (cin & 15) || (cout != 64 && cout != 128) || (ks != 1 && ks != 3 && ks != 5))
    return TSR_ERR_ARG;
  const int k32 = k32_mode(ks, cin, cout);
  const int tps = k32 ? 1 : taps_per_step(ks, cout, 2);
  const size_t total = (size_t)cout * cin * (((ks * ks + tps - 1) / tps) * tps);
  const int grid = (int)((total + 255) / 256);
  hipLaunchKernelGGL(pack_conv_weight_bf16s_kernel<true>, dim3(grid > 4096 ? 4096 : grid), dim3(256), 0,
                     (hipStream_t)stream, w_oihw, (_Float16*)w_packed, cout, cin, ks, 2, tps, -1, 0, 1.0f, w_amax, k32);
  return tsr_check_launch();
}

extern "C" int tsr_pack_conv_weight_dgrad_f16s_dev(const float* w_oihw, void* w_packed, int cout, int cin, int ks,
                                                   int ci0, int nprime, const float* w_amax, void* stream) {
  if (!w_oihw || !w_packed || !w_amax || (cout & 15) || (nprime != 64 && nprime != 128) || ci0 < 0 ||
      ci0 + nprime > cin || (ks != 1 && ks != 3 && ks != 5))
    return TSR_ERR_ARG;
  const int k32 = k32_mode(ks, cout, nprime);
  const int tps = k32 ? 1 : taps_per_step(ks, nprime, 2);
  const size_t total = (size_t)nprime * cout * (((ks * ks + tps - 1) / tps) * tps);
  const int grid = (int)((total + 255) / 256);
  hipLaunchKernelGGL(pack_conv_weight_bf16s_kernel<true>, dim3(grid > 4096 ? 4096 : grid), dim3(256), 0,
                     (hipStream_t)stream, w_oihw, (_Float16*)w_packed, nprime, cout, ks, 2, tps, ci0, cin, 1.0f, w_amax, k32);
  return tsr_check_launch();
}

extern "C" int tsr_pack_conv_weight_dgrad_bf16s(const float* w_oihw, void* w_packed, int cout, int cin, int ks,
                                                int ci0, int nprime, int nsplit, void* stream) {
  if (!w_oihw || !w_packed || (cout & 15) || (nprime != 64 && nprime != 128) || ci0 < 0 || ci0 + nprime > cin ||
      (ks != 1 && ks != 3 && ks != 5) || nsplit < 1 || nsplit > 3)
    return TSR_ERR_ARG;
  const int tps = taps_per_step(ks, nprime, nsplit);
  const size_t total = (size_t)nprime * cout * (((ks * ks + tps - 1) / tps) * tps);
  const int grid = (int)((total + 255) / 256);
  hipLaunchKernelGGL(pack_conv_weight_bf16s_kernel<false>, dim3(grid > 4096 ? 4096 : grid), dim3(256), 0,
                     (hipStream_t)stream, w_oihw, (__bf16*)w_packed, nprime, cout, ks, nsplit, tps, ci0, cin, 1.0f, nullptr, 0);
  return tsr_check_launch();
}

template <int KS, int COUT, int NS, bool EXT, bool F16>
static int launch_bf16s(const ConvArgs& a, hipStream_t st) {
  // C_out = 64, fp16 3x3 / 5x5: 4 images per workgroup, every wave owns all 64 channels of one image (measured: the
  // 1x1 and the 3-plane bf16 forms are faster as pairs).  The training epilogues index their statistics slabs by
  // (workgroup, image slot): tsr_conv2d_slab_entries_ex tells the caller how many entries this form writes.
  // One plane (plain bf16, the reduced-precision train mode): 4 images for both channel counts -- with one product
  // per MAC the weight stream per MFMA is what bounds the 2-image form (see launch_b16).
  if constexpr (F16 && NS == 2 && KS > 1) {
    if (use_k32(KS, a.cin, COUT)) return tsr_conv_k32(a, COUT, KS, EXT, st);
  }
  constexpr int WN = (KS > 1 && ((COUT == 64 && F16) || NS == 1)) ? 1 : 2;
  constexpr int IMG = 4 / WN;
  const int grid = ((a.B + IMG - 1) / IMG) * a.tiles_x * a.tiles_y;
  if constexpr (KS == 3 && WN == 2) {      // (4-image workgroups: a second slab costs occupancy; measured slower)
    if (((a.cin >> 4) & 1) == 0) {     // double-buffered halo (needs the channel blocks in pairs)
      hipLaunchKernelGGL((conv_mfma_split16_kernel<KS, COUT, NS, EXT, F16, WN, true>), dim3(grid), dim3(256), 0, st, a);
      return tsr_check_launch();
    }
  }
  hipLaunchKernelGGL((conv_mfma_split16_kernel<KS, COUT, NS, EXT, F16, WN, false>), dim3(grid), dim3(256), 0, st, a);
  return tsr_check_launch();
}

template <int NS, bool EXT, bool F16 = false>
static int dispatch_bf16s(const ConvArgs& a, int cout, int ks, hipStream_t st) {
  if (cout == 64) {
    if (ks == 1) return launch_bf16s<1, 64, NS, EXT, F16>(a, st);
    if (ks == 3) return launch_bf16s<3, 64, NS, EXT, F16>(a, st);
    if (ks == 5) return launch_bf16s<5, 64, NS, EXT, F16>(a, st);
  } else if (cout == 128) {
    if (ks == 1) return launch_bf16s<1, 128, NS, EXT, F16>(a, st);
    if (ks == 3) return launch_bf16s<3, 128, NS, EXT, F16>(a, st);
    if (ks == 5) return launch_bf16s<5, 128, NS, EXT, F16>(a, st);
  }
  return TSR_ERR_ARG;
}

extern "C" int tsr_pack_conv_weight_dgrad_f16s(const float* w_oihw, void* w_packed, int cout, int cin, int ks,
                                               int ci0, int nprime, float wscale, void* stream) {
  if (!w_oihw || !w_packed || (cout & 15) || (nprime != 64 && nprime != 128) || ci0 < 0 || ci0 + nprime > cin ||
      (ks != 1 && ks != 3 && ks != 5) || !(wscale > 0.f))
    return TSR_ERR_ARG;
  const int k32 = k32_mode(ks, cout, nprime);
  const int tps = k32 ? 1 : taps_per_step(ks, nprime, 2);
  const size_t total = (size_t)nprime * cout * (((ks * ks + tps - 1) / tps) * tps);
  const int grid = (int)((total + 255) / 256);
  hipLaunchKernelGGL(pack_conv_weight_bf16s_kernel<true>, dim3(grid > 4096 ? 4096 : grid), dim3(256), 0,
                     (hipStream_t)stream, w_oihw, (_Float16*)w_packed, nprime, cout, ks, 2, tps, ci0, cin, wscale, nullptr, k32);
  return tsr_check_launch();
}

// 1x1 convolution on bf16 CB16 tensors as a STREAMING GEMM (training with bf16 activation storage: the MSRB `confusion`
// forward, 256 -> 64 with a virtual input and a residual, and its two dgrad launches, 64 -> 128 with the ReLU-mask /
// BatchNorm-sum epilogue).  The layer is HBM-bound (0.1 ms of MFMA work against 2-2.5 GB per launch at B = 2048) and the
// tiled kernel above spends its time on barriers: one K = 16 step of 8 MFMAs per (slab staging, LDS round trip,
// __syncthreads) -- 2.6 TB/s.  Here nothing but the weights touches LDS: a CB16 line of bf16 is 32 B, so the A fragment
// of v_mfma_f32_32x32x16_bf16 (lane = (pixel, k half), 8 consecutive channels) IS one 16-B global load per lane, the
// 32 lanes of a patch row pair cover 256-B runs, and a wave keeps all its loads of a 4-block chunk in flight.  Persistent
// workgroups (the packed weight matrix is staged once), wave = one image x 8x8 patch x ALL C_out, so the shared
// epilogue (conv_epilogue<COUT, true, 1, true>) applies unchanged with 4 images per slab group.
template <int COUT>
__global__ __launch_bounds__(256, 2) void conv1x1_b16_ex_kernel(const ConvArgs a) {
  constexpr int NB = COUT / 32;
  extern __shared__ __attribute__((aligned(16))) char lds1[];      // [cin/16][2][COUT][8] bf16, then (scale, shift)[cin]
  const int tid = threadIdx.x, lane = tid & 63, wm = tid >> 6;
  const int h = lane >> 5, li = lane & 31;
  const int nblk = a.cin >> 4;
  const int wbytes = a.cin * COUT * 2;
  for (int i = tid * 16; i < wbytes; i += 256 * 16) *(f32x4*)(lds1 + i) = *(const f32x4*)((const char*)a.wp + i);
  float* tsc = (float*)(lds1 + wbytes);
  if (a.in_scale)
    for (int c = tid; c < a.cin; c += 256) { tsc[c] = a.in_scale[c]; tsc[a.cin + c] = a.in_shift[c]; }
  __syncthreads();

  const int HW = a.H * a.W, tpi = a.tiles_x * a.tiles_y;
  const int in_blocks = a.in_ctot >> 4;
  const int total = ((a.B + 3) >> 2) * tpi;
  const bf16x8* wl = (const bf16x8*)lds1 + (h * COUT + li);          // + blk * 2 * COUT + nb * 32
  for (int t = blockIdx.x; t < total; t += gridDim.x) {
    const int ig = t / tpi, trem = t - ig * tpi;
    const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
    const int y0 = ty * 8, x0 = tx * 8, b0 = ig * 4;
    const int b = b0 + wm < a.B ? b0 + wm : a.B - 1;       // rows of an absent image / pixel compute on valid data and
    int pix[2];                                            // are never stored (the epilogue tests the real coordinates)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      int gy = y0 + 4 * mb + (li >> 3), gx = x0 + (li & 7);
      gy = gy < a.H ? gy : a.H - 1;
      gx = gx < a.W ? gx : a.W - 1;
      pix[mb] = gy * a.W + gx;
    }
    const __bf16* inb = (const __bf16*)a.in + ((size_t)b * in_blocks + (a.in_coff >> 4)) * HW * 16 + h * 8;
    f32x16 acc[2][NB];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;
    for (int c0 = 0; c0 < nblk; c0 += 4) {                 // chunks of 4 channel blocks: 8 loads in flight per lane
      bf16x8 fa[4][2];
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          const int c = c0 + k < nblk ? c0 + k : nblk - 1;
          fa[k][mb] = *(const bf16x8*)(inb + ((size_t)c * HW + pix[mb]) * 16);
        }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (c0 + k < nblk) {
          const int c = c0 + k;
          if (a.in_scale) {      // producer's train-mode BN + ReLU on the stored pre-activation, fp32, back to the bf16 operand
            const f32x4 s0 = *(const f32x4*)(tsc + c * 16 + h * 8), s1 = *(const f32x4*)(tsc + c * 16 + h * 8 + 4);
            const f32x4 t0 = *(const f32x4*)(tsc + a.cin + c * 16 + h * 8), t1 = *(const f32x4*)(tsc + a.cin + c * 16 + h * 8 + 4);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                const float sc = j < 4 ? s0[j & 3] : s1[j & 3], sh = j < 4 ? t0[j & 3] : t1[j & 3];
                fa[k][mb][j] = (__bf16)tsr_relu(fmaf((float)fa[k][mb][j], sc, sh));
              }
          }
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            const bf16x8 fb = wl[(size_t)c * 2 * COUT + nb * 32];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) acc[mb][nb] = Plane<false>::mfma(fa[k][mb], fb, acc[mb][nb]);
          }
        }
      }
    }
    conv_epilogue<COUT, true, 1, true>(a, acc, t, b0, y0, x0, wm, 0, h, li, HW, 1.f);
  }
}

template <int COUT>
static int launch_1x1_b16_ex(const ConvArgs& a, hipStream_t st) {
  const int total = ((a.B + 3) / 4) * a.tiles_x * a.tiles_y;
  const size_t smem = (size_t)a.cin * COUT * 2 + (size_t)a.cin * 8;
  if (smem > 72 * 1024) return TSR_ERR_ARG;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv1x1_b16_ex_kernel<COUT>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv1x1_b16_ex_kernel<COUT>), dim3(total < 512 ? total : 512), dim3(256), smem, st, a);
  return tsr_check_launch();
}

// Training launches with bf16 ACTIVATION STORAGE (tsr_conv_desc.nsplit == -1): in / res / mask / out are bf16 CB16 tensors,
// one bf16 plane, fp32 accumulation, the training epilogues (statistics from the fp32 accumulator, bf16 stores).
template <int KS, int COUT>
static int launch_b16_ex(const ConvArgs& a, hipStream_t st) {
  if constexpr (KS > 1) {          // 4 images per workgroup, wave = image x all C_out (as launch_b16 / the NS = 1 train form)
    const int grid4 = ((a.B + 3) / 4) * a.tiles_x * a.tiles_y;
    hipLaunchKernelGGL((conv_mfma_split16_kernel<KS, COUT, 1, true, false, 1, false, true>), dim3(grid4), dim3(256), 0, st, a);
    return tsr_check_launch();
  }
  // The streaming form runs the `confusion` FORWARD (256 -> 64, plain epilogue: 0.96 -> 0.81 ms at B = 2048); its dgrad
  // launches (64 -> 128 with the mask / BatchNorm-sum epilogue, 8-B epilogue accesses on bf16 tensors) measured slower
  // streamed (0.94 vs 0.77 ms) and stay on the tiled kernel -- which also keeps the 2-image statistics-slab numbering
  // of every 1x1 launch that writes slabs.
  if constexpr (COUT == 64) {
    if (a.epi_mode == 0) return launch_1x1_b16_ex<COUT>(a, st);
  }
  const int grid = ((a.B + 1) / 2) * a.tiles_x * a.tiles_y;
  hipLaunchKernelGGL((conv_mfma_split16_kernel<KS, COUT, 1, true, false, 2, false, true>), dim3(grid), dim3(256), 0, st, a);
  return tsr_check_launch();
}

int tsr_conv_b16k_ex(const ConvArgs& a, int cout, int ks, bool pair, hipStream_t st);       // conv_b16k.hip
static int dispatch_b16_ex(const ConvArgs& a, int cout, int ks, hipStream_t st) {
  if (cout == 64) {
    if (ks == 1) return launch_b16_ex<1, 64>(a, st);
    if (ks == 3) return launch_b16_ex<3, 64>(a, st);
    if (ks == 5) return launch_b16_ex<5, 64>(a, st);
  } else if (cout == 128) {
    if (ks == 1) return launch_b16_ex<1, 128>(a, st);
    if (ks == 3) return launch_b16_ex<3, 128>(a, st);
    if (ks == 5) return launch_b16_ex<5, 128>(a, st);
  }
  return TSR_ERR_ARG;
}

// tsr_conv2d_ex with nsplit != 0 lands here (argument checks were done by the caller)
int tsr_conv2d_ex_bf16s(const ConvArgs& a, int cout, int ks, int nsplit, hipStream_t st) {
  if (nsplit == -3 || nsplit == -4) return tsr_conv_b16k_ex(a, cout, ks, nsplit == -4, st);   // bf16 storage, conv_b16k.hip
  if (nsplit == -1) return dispatch_b16_ex(a, cout, ks, st);                    // bf16 storage + bf16 operands
  if (nsplit == -2) return dispatch_bf16s<2, true, true>(a, cout, ks, st);      // fp16x3
  if (nsplit == 3) return dispatch_bf16s<3, true>(a, cout, ks, st);
  if (nsplit == 2) return dispatch_bf16s<2, true>(a, cout, ks, st);
  if (nsplit == 1) return dispatch_bf16s<1, true>(a, cout, ks, st);
  return TSR_ERR_ARG;
}

extern "C" int tsr_conv2d_fwd_bf16s(const float* in, int in_ctot, int in_coff, int cin,
                                    const void* w_packed, int cout, int ks, int nsplit,
                                    const float* scale, const float* shift,
                                    const float* res, int res_ctot, int res_coff,
                                    float* out, int out_ctot, int out_coff, int relu,
                                    int B, int H, int W, void* stream) {
  if (!in || !w_packed || !out || B <= 0 || H <= 0 || W <= 0 || nsplit < 1 || nsplit > 3) return TSR_ERR_ARG;
  if ((cin & 15) || (in_ctot & 15) || (in_coff & 15) || (out_ctot & 15) || (out_coff & 15) || cin <= 0 ||
      in_coff + cin > in_ctot || out_coff + cout > out_ctot)
    return TSR_ERR_ARG;
  if (res && ((res_ctot & 15) || (res_coff & 15) || res_coff + cout > res_ctot)) return TSR_ERR_ARG;
  ConvArgs a = {};
  a.in = in; a.in_ctot = in_ctot; a.in_coff = in_coff; a.cin = cin;
  a.wp = (const float*)w_packed; a.scale = scale; a.shift = shift;
  a.res = res; a.res_ctot = res_ctot; a.res_coff = res_coff;
  a.out = out; a.out_ctot = out_ctot; a.out_coff = out_coff; a.relu = relu;
  a.B = B; a.H = H; a.W = W;
  a.tiles_x = (W + 7) / 8; a.tiles_y = (H + 7) / 8;
  hipStream_t st = (hipStream_t)stream;
  if (nsplit == 3) return dispatch_bf16s<3, false>(a, cout, ks, st);
  if (nsplit == 2) return dispatch_bf16s<2, false>(a, cout, ks, st);
  return dispatch_bf16s<1, false>(a, cout, ks, st);
}

// Stage-2 convolution of an MSRB with its half of the 1x1 `confusion` fused (conv_fuse1x1.h): fp16x3 arithmetic,
// C_out = 128, ks = 3 or 5; out / res describe the 64-channel result of the fused second GEMM.
extern "C" int tsr_conv2d_fwd_f16s_fuse1x1(const float* in, int in_ctot, int in_coff, int cin,
                                           const void* w_packed, int ks, float w_inv_scale,
                                           const float* in_amax, float* out_amax,
                                           const float* scale, const float* shift, int relu,
                                           const void* w2_packed, float w2_inv_scale, const float* shift2,
                                           const float* res, int res_ctot, int res_coff,
                                           float* out, int out_ctot, int out_coff, int relu2,
                                           int B, int H, int W, void* stream) {
  if (!in || !w_packed || !w2_packed || !out || !in_amax || B <= 0 || H <= 0 || W <= 0 || !(w_inv_scale > 0.f) ||
      !(w2_inv_scale > 0.f) || (ks != 3 && ks != 5))
    return TSR_ERR_ARG;
  if ((cin & 15) || (in_ctot & 15) || (in_coff & 15) || (out_ctot & 15) || (out_coff & 15) || cin <= 0 ||
      in_coff + cin > in_ctot || out_coff + 64 > out_ctot)
    return TSR_ERR_ARG;
  if (res && ((res_ctot & 15) || (res_coff & 15) || res_coff + 64 > res_ctot)) return TSR_ERR_ARG;
  ConvArgs a = {};
  a.in = in; a.in_ctot = in_ctot; a.in_coff = in_coff; a.cin = cin;
  a.wp = (const float*)w_packed; a.scale = scale; a.shift = shift; a.relu = relu;
  a.res = res; a.res_ctot = res_ctot; a.res_coff = res_coff;
  a.out = out; a.out_ctot = out_ctot; a.out_coff = out_coff;
  a.B = B; a.H = H; a.W = W;
  a.tiles_x = (W + 7) / 8; a.tiles_y = (H + 7) / 8;
  a.in_amax = in_amax; a.w_inv_scale = w_inv_scale; a.out_amax = out_amax;
  a.w2 = w2_packed; a.w2_inv_scale = w2_inv_scale; a.shift2 = shift2; a.relu2 = relu2;
  const int grid = ((B + 1) / 2) * a.tiles_x * a.tiles_y;
  hipStream_t st = (hipStream_t)stream;
  if (use_k32(ks, cin, 128)) return tsr_conv_k32_fuse1x1(a, ks, st);
  if (ks == 5) {
    hipLaunchKernelGGL((conv_mfma_split16_kernel<5, 128, 2, false, true, 2, false, false, true>), dim3(grid), dim3(256), 0, st, a);
  } else if (((cin >> 4) & 1) == 0) {
    hipLaunchKernelGGL((conv_mfma_split16_kernel<3, 128, 2, false, true, 2, true, false, true>), dim3(grid), dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL((conv_mfma_split16_kernel<3, 128, 2, false, true, 2, false, false, true>), dim3(grid), dim3(256), 0, st, a);
  }
  return tsr_check_launch();
}

// bf16 activation storage (BASELINE's "bf16" configurations; reduced precision, never the parity path): `in`, `res`
// and `out` are bf16 CB16 tensors, weights the one-plane pack of tsr_pack_conv_weight_bf16s(nsplit = 1), fp32
// accumulation and epilogue arithmetic.
template <int KS, int COUT>
static int launch_b16(const ConvArgs& a, hipStream_t st) {
  if constexpr (KS > 1) {
    // 4 images per workgroup, wave = image x all C_out: the weight slab of a step serves 256 pixels instead of 128 --
    // with one product per MAC the weight stream through L2 / LDS is what bounds the 2-image form (measured at
    // B = 4096: 5x5 128->128 31.7 -> 25.1 ms per 6 launches, 3x3 64->64 9.0 -> 6.9 ms; whole forward 53.9 k -> 64.2 k/s)
    const int grid4 = ((a.B + 3) / 4) * a.tiles_x * a.tiles_y;
    // (a double-buffered halo for the 3x3 form of this variant measured slower: 3x3 64->64 6.9 -> 8.5 ms)
    hipLaunchKernelGGL((conv_mfma_split16_kernel<KS, COUT, 1, false, false, 1, false, true>), dim3(grid4), dim3(256), 0, st, a);
    return tsr_check_launch();
  } else {        // 1x1: two images per workgroup
    const int grid = ((a.B + 1) / 2) * a.tiles_x * a.tiles_y;
    hipLaunchKernelGGL((conv_mfma_split16_kernel<KS, COUT, 1, false, false, 2, false, true>), dim3(grid), dim3(256), 0, st, a);
    return tsr_check_launch();
  }
}

extern "C" int tsr_conv2d_fwd_b16(const void* in, int in_ctot, int in_coff, int cin,
                                  const void* w_packed, int cout, int ks,
                                  const float* scale, const float* shift,
                                  const void* res, int res_ctot, int res_coff,
                                  void* out, int out_ctot, int out_coff, int relu,
                                  int B, int H, int W, void* stream) {
  if (!in || !w_packed || !out || B <= 0 || H <= 0 || W <= 0) return TSR_ERR_ARG;
  if ((cin & 15) || (in_ctot & 15) || (in_coff & 15) || (out_ctot & 15) || (out_coff & 15) || cin <= 0 ||
      in_coff + cin > in_ctot || out_coff + cout > out_ctot)
    return TSR_ERR_ARG;
  if (res && ((res_ctot & 15) || (res_coff & 15) || res_coff + cout > res_ctot)) return TSR_ERR_ARG;
  ConvArgs a = {};
  a.in = (const float*)in; a.in_ctot = in_ctot; a.in_coff = in_coff; a.cin = cin;
  a.wp = (const float*)w_packed; a.scale = scale; a.shift = shift;
  a.res = (const float*)res; a.res_ctot = res_ctot; a.res_coff = res_coff;
  a.out = (float*)out; a.out_ctot = out_ctot; a.out_coff = out_coff; a.relu = relu;
  a.B = B; a.H = H; a.W = W;
  a.tiles_x = (W + 7) / 8; a.tiles_y = (H + 7) / 8;
  hipStream_t st = (hipStream_t)stream;
  if (cout == 64) {
    if (ks == 1) return launch_b16<1, 64>(a, st);
    if (ks == 3) return launch_b16<3, 64>(a, st);
    if (ks == 5) return launch_b16<5, 64>(a, st);
  } else if (cout == 128) {
    if (ks == 1) return launch_b16<1, 128>(a, st);
    if (ks == 3) return launch_b16<3, 128>(a, st);
    if (ks == 5) return launch_b16<5, 128>(a, st);
  }
  return TSR_ERR_ARG;
}

// fp16 two-plane variant ("fp16x3": x*sx = h1+h2, w*sw = g1+g2, products h1g1 + h1g2 + h2g1, fp32 accumulate).
// in_amax: device scalar holding max|x| of the input tensor (written by its producer through out_amax);
// out_amax: device scalar that receives (atomic max) max|y| of this launch's output, for the consumer.
extern "C" int tsr_conv2d_fwd_f16s(const float* in, int in_ctot, int in_coff, int cin,
                                   const void* w_packed, int cout, int ks, float w_inv_scale,
                                   const float* in_amax, float* out_amax,
                                   const float* scale, const float* shift,
                                   const float* res, int res_ctot, int res_coff,
                                   float* out, int out_ctot, int out_coff, int relu,
                                   int B, int H, int W, void* stream) {
  if (!in || !w_packed || !out || !in_amax || B <= 0 || H <= 0 || W <= 0 || !(w_inv_scale > 0.f)) return TSR_ERR_ARG;
  if ((cin & 15) || (in_ctot & 15) || (in_coff & 15) || (out_ctot & 15) || (out_coff & 15) || cin <= 0 ||
      in_coff + cin > in_ctot || out_coff + cout > out_ctot)
    return TSR_ERR_ARG;
  if (res && ((res_ctot & 15) || (res_coff & 15) || res_coff + cout > res_ctot)) return TSR_ERR_ARG;
  ConvArgs a = {};
  a.in = in; a.in_ctot = in_ctot; a.in_coff = in_coff; a.cin = cin;
  a.wp = (const float*)w_packed; a.scale = scale; a.shift = shift;
  a.res = res; a.res_ctot = res_ctot; a.res_coff = res_coff;
  a.out = out; a.out_ctot = out_ctot; a.out_coff = out_coff; a.relu = relu;
  a.B = B; a.H = H; a.W = W;
  a.tiles_x = (W + 7) / 8; a.tiles_y = (H + 7) / 8;
  a.in_amax = in_amax; a.w_inv_scale = w_inv_scale; a.out_amax = out_amax;
  return dispatch_bf16s<2, false, true>(a, cout, ks, (hipStream_t)stream);
}

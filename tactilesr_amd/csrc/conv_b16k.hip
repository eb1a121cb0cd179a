// Inference convolutions of the bf16 ACTIVATION-STORAGE path (BASELINE's "bf16" configurations: bf16 CB16 tensors in HBM,
// plain bf16 MFMA operands, fp32 accumulation) on v_mfma_f32_16x16x32_bf16, CHANNELS AS ROWS:
//
//     D[co][pixel] += W[co][ci] . X[ci][pixel]          A operand = weights (M = C_out), B operand = halo pixels (N)
//
//   * Why this orientation.  The accumulator of a 16x16 tile then holds, per lane, FOUR CONSECUTIVE CHANNELS of ONE
//     pixel (lane = (pixel n = lane & 15, channel quad g = lane >> 4)) -- which is (a) the CB16 storage order, so the
//     output stage is one 8-B bf16 store per tile with no transpose, and (b), for two channel tiles side by side, exactly
//     the B-operand register layout of the NEXT matrix product over those channels: the MSRB's fused 1x1 `confusion`
//     half (model/tactileSR_model.py:196-206) runs on the converted accumulators directly -- no LDS round trip, no
//     barrier, no 4x4 transposes (the 32x32x16 pixel-row kernel parked the tile in LDS: its epilogue was 26-28 k cycles,
//     40 % of the 3x3 launch's workgroup time).  The 1x1 weight is packed in the matching K order (tsr_pack_w2_b16k).
//   * Why this instruction.  Under an MFMA-dense loop the chip holds a higher clock on the 16x16x32 shape than on
//     32x32x16 at equal cycles per FLOP (MI355X_MICROARCH.md, DVFS give-back item 7).
//   * K = 32 = the 32 input channels of TWO CB16 blocks at ONE tap (the tensors are bf16: a halo pixel of a 32-channel
//     block is 64 B): no tap pairing, no odd-tap cross step.
//
// Workgroup = 256 threads = 4 images x one 8x8 patch x all C_out; wave = image; per wave C_out/16 x 4 accumulator tiles.
// LDS: halo slab [image][row][pixel][32 ch] (row stride == 2 (mod 4) 16-B slots: conflict-free ds_read_b128 B fragments)
// + a 3-slot ring of one-tap weight slabs [k group 4][C_out][8] fed by LDS-DMA.  Slab s is read (A fragments, one step
// ahead, refilled in place as the rows of the step retire) during step s-1 only, so slab s+3 is requested at the START of
// step s into the slot slab s just left and has two whole steps to land.  Per step and wave: 32 MFMAs, 12 ds_read_b128.
// Forms: plain / FUSED (stage 2 of an MSRB + its 1x1 half) / PAIR (stage 1: conv_3_1 || conv_5_1 on one halo; the 3x3
// half's MFMAs and fragment reads are skipped on the 16 outer taps).  The bf16-storage TRAIN step runs its 128-channel
// launches on the same loop (tsr_conv2d_ex, nsplit = -3 / -4): DGRAD (ReLU-mask + BatchNorm-backward-sum epilogue), TRAIN and
// PAIR_TRAIN (raw output + Welford partials) -- their inputs are stored gradients, stored block outputs or materialised
// activations (tsr_bn_relu_b16), i.e. nothing needs transforming while staging.
#include "tsr_common.h"
#include "conv_args.h"
#include "tactilesr_hip.h"
#include <type_traits>

typedef __bf16 kb16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 kb16x4 __attribute__((ext_vector_type(4)));
typedef unsigned ku32x4 __attribute__((ext_vector_type(4)));
typedef int ki32x4 __attribute__((ext_vector_type(4)));

// LDS-DMA (buffer_load_dwordx4 ... lds: 16 B per lane, global -> LDS at m0 + 16 * lane, no VGPR hop) as inline assembly: the
// builtin form needs an exec-masked BRANCH around a request that only part of the wave takes part in, and a basic-block
// boundary inside a step costs hipcc its register allocation (it starts spilling accumulators).  `rs` = raw buffer
// descriptor (base, 0 stride, num_records, flags), `vo` = per-lane byte offset, `so` = scalar byte offset, `m0v` = LDS byte
// address of lane 0's 16 B.  The hardware counts these like any vector-memory load (vmcnt, in issue order).
__device__ __forceinline__ void b16k_dma(ki32x4 rs, int vo, int so, unsigned m0v) {
  unsigned keep;       // (M0 is compiler-reserved: saved and restored inside the statement)
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(m0v), "v"(vo), "s"(rs), "s"(so));
}
// one request of the lanes in `mask` only; `m0b` != 0: a second one with the same source bytes to that LDS row (the branch
// lives inside the statement: no basic-block boundary for the compiler)
__device__ __forceinline__ void b16k_dma_row(ki32x4 rs, int vo, int so, unsigned m0a, unsigned m0b, unsigned long long mask) {
  unsigned long long sv;
  unsigned keep;
  asm volatile("s_mov_b32 %1, m0\n\ts_mov_b64 %0, exec\n\ts_mov_b64 exec, %7\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
               "buffer_load_dwordx4 %4, %5, %6 offen lds\n\ts_cmp_eq_u32 %3, 0\n\ts_cbranch_scc1 .Lb16k_skip%=\n\t"
               "s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %5, %6 offen lds\n"
               ".Lb16k_skip%=:\n\ts_mov_b64 exec, %0\n\ts_mov_b32 m0, %1"
               : "=&s"(sv), "=&s"(keep) : "s"(m0a), "s"(m0b), "v"(vo), "s"(rs), "s"(so), "s"(mask) : "scc");
}
// one request of the lanes in `mask` only (mask 0: the wave takes no part)
__device__ __forceinline__ void b16k_dma_masked(ki32x4 rs, int vo, int so, unsigned m0a, bool on) {
  unsigned long long sv;
  unsigned keep;
  const unsigned m32 = (unsigned)__builtin_amdgcn_readfirstlane(-(int)on);      // (wave-uniform by construction)
  const unsigned long long mask = ((unsigned long long)m32 << 32) | m32;
  asm volatile("s_mov_b32 %1, m0\n\ts_mov_b64 %0, exec\n\ts_mov_b64 exec, %6\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
               "buffer_load_dwordx4 %3, %4, %5 offen lds\n\ts_mov_b64 exec, %0\n\ts_mov_b32 m0, %1"
               : "=&s"(sv), "=&s"(keep) : "s"(m0a), "v"(vo), "s"(rs), "s"(so), "s"(mask));
}
__device__ __forceinline__ unsigned b16k_lds_addr(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}

constexpr int b16k_row_slots(int px) {
  int rs = px * 4;
  while ((rs & 3) != 2) ++rs;
  return rs;
}

template <int KS, int COUT, int NW> struct B16KGeom {
  static constexpr int HH = 8 + KS - 1;
  static constexpr int T = KS * KS;
  static constexpr int PIXB = 64;
  static constexpr int ROWB = b16k_row_slots(HH) * 16;
  static constexpr int NROW = 16;                    // physical halo rows per image: a circular buffer (see below)
  static constexpr int IMGB = (NROW + 1) * ROWB;     // + row 16, a copy of row 0 (a row pair may start at row 15)
  static constexpr int HALO_B = NW * IMGB;           // one image per wave
  static constexpr int WTAP_B = 64 * COUT;           // 32 channels x C_out bf16
  static constexpr int RING = 3;
  static constexpr int LDS_B = HALO_B + RING * WTAP_B;
};

enum { B16K_PLAIN = 0, B16K_FUSED = 1, B16K_PAIR = 2, B16K_DGRAD = 3, B16K_TRAIN = 4, B16K_PAIR_TRAIN = 5 };
constexpr bool b16k_pair(int mode) { return mode == B16K_PAIR || mode == B16K_PAIR_TRAIN; }


// compile-time loop: f(std::integral_constant<int, I>) for I = 0 .. N-1 (the scheduling hints need constant operands)
template <int I, int N, class F> __device__ __forceinline__ void b16k_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>());
    b16k_static_for<I + 1, N>(f);
  }
}
// pair form: first C_out tile with work at tap t (the 3x3 conv, tiles 0..3, has no weight outside the inner 3x3 taps)
template <int KS, int MODE, int MT> constexpr int b16k_tap_rows(int t) {
  const int kh = t / KS, kw = t - kh * KS;
  return (!b16k_pair(MODE) || (kh >= 1 && kh <= 3 && kw >= 1 && kw <= 3)) ? 0 : MT / 2;
}

typedef float kf32x8 __attribute__((ext_vector_type(8)));

// Epilogue.  Lane = (pixel m of the 16-pixel row pair nt, channel quad g): acc[mt][nt] holds channels 16 mt + 4 g .. + 3 of that
// pixel, i.e. 8 contiguous bytes of the bf16 CB16 tensor.
//   * ReLU is t + |t| on HALVED operands (one instruction, NaN-propagating like torch's; v_max would drop a NaN and a
//     compare + select is two): the BatchNorm vectors are halved once per channel (exact), in the fused form the 1x1 weight
//     is packed halved (tsr_pack_w2_b16k) and bias / residual enter through an fma with 1/2; without ReLU the same h is
//     doubled (h + h).
//   * Fused form: the rounded stage-1 tile IS the B operand of the second product -- K slot (g, j) of K step kk = channel
//     32 kk + 4 g + j (j < 4) or 32 kk + 16 + 4 g + j - 4 -- so the 1x1 runs straight from the accumulator registers.
//   * Every global read is requested before the first store (behind a store the compiler may not hoist a load: the
//     pointers could alias) and addresses are a uniform base + a 32-bit lane offset.
template <int MT, int MODE>
__device__ __forceinline__ void b16k_epilogue(const ConvArgs& a, f32x4 (&acc)[MT][4], int b, int y0, int x0, int m, int g, int HW,
                                              const char* w2h0, const char* w2h1) {
  constexpr int NT = 4;
  const bool img_ok = b < a.B;
  const int bsafe = img_ok ? b : 0;
  bool ok[NT];
  unsigned po[NT];                      // byte offset of this lane's channel quad inside a 16-channel plane
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int gy = y0 + 2 * nt + (m >> 3), gx = x0 + (m & 7);
    ok[nt] = img_ok && gy < a.H && gx < a.W;
    po[nt] = (ok[nt] ? (unsigned)(gy * a.W + gx) * 32u : 0u) + 8u * g;
  }
  const unsigned plane = (unsigned)HW * 32u;                 // bytes per (image, 16-channel block)
  const char* rb = a.res ? (const char*)a.res + ((size_t)bsafe * (a.res_ctot >> 4) + (a.res_coff >> 4)) * plane : nullptr;
  char* ob = (char*)a.out + ((size_t)bsafe * (a.out_ctot >> 4) + (a.out_coff >> 4)) * plane;
  const f32x4 one4 = {1.f, 1.f, 1.f, 1.f}, zero4 = {0.f, 0.f, 0.f, 0.f};
  const kb16x4 zero4h = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
  auto relu_x2 = [](f32x4 t) __attribute__((always_inline)) {
    return (f32x4){tsr_relu_x2(t[0]), tsr_relu_x2(t[1]), tsr_relu_x2(t[2]), tsr_relu_x2(t[3])};
  };

  // BatchNorm fold of all C_out tiles, in place (halved where a ReLU follows)
  {
    const float hs = a.relu ? 0.5f : 1.f;
    f32x4 sc[MT], sh[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      sc[mt] = a.scale ? *(const f32x4*)(a.scale + mt * 16 + 4 * g) : one4;
      sh[mt] = a.shift ? *(const f32x4*)(a.shift + mt * 16 + 4 * g) : zero4;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      sc[mt] *= hs;
      sh[mt] *= hs;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = acc[mt][nt] * sc[mt] + sh[mt];
    }
  }

  if constexpr (MODE == B16K_FUSED) {
    // 1. the residual tile and the bias are requested HERE -- the BatchNorm vectors' registers are free again -- and travel
    //    under the conversion
    __builtin_amdgcn_sched_barrier(0);
    kb16x8 wa[4][4];
    f32x4 sh2[4];
    kb16x4 rv[4][NT];
#pragma unroll
    for (int m2 = 0; m2 < 4; ++m2)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) rv[m2][nt] = zero4h;
    if (rb) {
#pragma unroll
      for (int m2 = 0; m2 < 4; ++m2)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) rv[m2][nt] = *(const kb16x4*)(rb + (m2 * plane + po[nt]));
    }
#pragma unroll
    for (int m2 = 0; m2 < 4; ++m2) sh2[m2] = a.shift2 ? *(const f32x4*)(a.shift2 + m2 * 16 + 4 * g) * 0.5f : zero4;
    __builtin_amdgcn_sched_barrier(0);
    // 2. stage-1 activation, rounded to bf16: the B fragments of the second product (the accumulators die here)
    kb16x8 bq[4][NT];
    if (a.relu) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          bq[kk][nt] = __builtin_convertvector(__builtin_shufflevector(relu_x2(acc[2 * kk][nt]), relu_x2(acc[2 * kk + 1][nt]),
                                                                       0, 1, 2, 3, 4, 5, 6, 7), kb16x8);
    } else {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          bq[kk][nt] = __builtin_convertvector(__builtin_shufflevector(acc[2 * kk][nt], acc[2 * kk + 1][nt], 0, 1, 2, 3, 4, 5, 6, 7), kb16x8);
    }
    // the W2 fragments: LDS, [kk & 1][g][64][8] inside the half kk >> 1 (landed and published by the main loop's last barriers)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int m2 = 0; m2 < 4; ++m2)
        wa[kk][m2] = *(const kb16x8*)((kk < 2 ? w2h0 : w2h1) + (((kk & 1) * 4 + g) * 64 + m2 * 16 + m) * 16);
    // 3. per pair of output tiles: acc2 = (W2 / 2) . bq; h = acc2 + (residual / 2 + shift2 / 2); out = h + |h| (ReLU) or h + h,
    //    bf16, 8-B stores
#define B16K_OUT2(EXPR_)                                                                  \
  _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                     \
    kb16x4 o[2];                                                                          \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                       \
      kb16x4 r_ = rv[2 * mh + u][nt];                                                     \
      asm volatile("" : "+v"(r_));       /* (hipcc would convert all 16 quads ahead of the product and spill them) */ \
      const f32x4 h = acc2[u][nt] + (__builtin_convertvector(r_, f32x4) * 0.5f + sh2[2 * mh + u]); \
      o[u] = __builtin_convertvector(EXPR_, kb16x4);                                      \
    }                                                                                     \
    if (ok[nt]) {                                                                         \
      _Pragma("unroll") for (int u = 0; u < 2; ++u) *(kb16x4*)(ob + ((2 * mh + u) * plane + po[nt])) = o[u]; \
    }                                                                                     \
  }
#pragma unroll
    for (int mh = 0; mh < 2; ++mh) {
      f32x4 acc2[2][NT];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc2[u][nt] = zero4;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc2[u][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[kk][2 * mh + u], bq[kk][nt], acc2[u][nt], 0, 0, 0);
      if (a.relu2) { B16K_OUT2(relu_x2(h)) } else { B16K_OUT2(h + h) }
    }
#undef B16K_OUT2
  } else {
    // plain form: h = bn(acc) [+ residual] (halved under a ReLU), nt-outer: one exec mask per row pair
    const float hs = a.relu ? 0.5f : 1.f;
#define B16K_OUT1(H_, EXPR_)                                                              \
  _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                     \
    kb16x4 o[MT];                                                                         \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                   \
      const f32x4 h = H_;                                                                 \
      o[mt] = __builtin_convertvector(EXPR_, kb16x4);                                     \
    }                                                                                     \
    if (ok[nt]) {                                                                         \
      _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) *(kb16x4*)(ob + (mt * plane + po[nt])) = o[mt]; \
    }                                                                                     \
  }
    if (rb) {      // (a residual: the ResBlock's second conv -- all quads of a row pair requested together, one pair ahead)
      kb16x4 rv[2][MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) rv[0][mt] = *(const kb16x4*)(rb + (mt * plane + po[0]));
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if (nt + 1 < NT) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) rv[(nt + 1) & 1][mt] = *(const kb16x4*)(rb + (mt * plane + po[nt + 1]));
        }
        kb16x4 o[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const f32x4 h = __builtin_convertvector(rv[nt & 1][mt], f32x4) * hs + acc[mt][nt];
          o[mt] = __builtin_convertvector(a.relu ? relu_x2(h) : h, kb16x4);
        }
        if (ok[nt]) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) *(kb16x4*)(ob + (mt * plane + po[nt])) = o[mt];
        }
      }
    } else if (a.relu) {
      B16K_OUT1(acc[mt][nt], relu_x2(h))
    } else {
      B16K_OUT1(acc[mt][nt], h)
    }
#undef B16K_OUT1
  }
}

// sum over the 16 lanes of a DPP row (the 16 pixels that hold the same channel quad); every lane ends up with the total
__device__ __forceinline__ float b16k_row_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // lane ^ 1
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // lane ^ 2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));   // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));   // row_ror:8
  return v;
}

// Training with bf16 activation storage, the DGRAD launches of the 128-channel 3x3 / 5x5 layers (tsr_conv2d_ex, epi_mode 2; the
// input is the stored gradient dz: nothing to transform while staging, so the inference loop applies unchanged):
//   x = acc * scale + res, zeroed where the stored activation's BatchNorm + ReLU was off (mask * mask_scale + mask_shift <= 0),
//   out = bf16(x); BatchNorm-backward partials sum(x), sum(x * xhat), xhat = mask * bn_a + bn_b, per (workgroup, image, channel)
//   into the slab (same entry numbering as the 32x32x16 kernel's 4-image form: entry = logical workgroup * 4 + image).
template <int MT>
__device__ __forceinline__ void b16k_epilogue_dgrad(const ConvArgs& a, f32x4 (&acc)[MT][4], int bid, int wm, int b, int y0, int x0,
                                                    int m, int g, int HW) {
  constexpr int NT = 4;
  const bool img_ok = b < a.B;
  const int bsafe = img_ok ? b : 0;
  bool ok[NT];
  unsigned po[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int gy = y0 + 2 * nt + (m >> 3), gx = x0 + (m & 7);
    ok[nt] = img_ok && gy < a.H && gx < a.W;
    po[nt] = (ok[nt] ? (unsigned)(gy * a.W + gx) * 32u : 0u) + 8u * g;
  }
  const unsigned plane = (unsigned)HW * 32u;
  const char* rb = a.res ? (const char*)a.res + ((size_t)bsafe * (a.res_ctot >> 4) + (a.res_coff >> 4)) * plane : nullptr;
  const char* mb = (const char*)a.mask + ((size_t)bsafe * (a.mask_ctot >> 4) + (a.mask_coff >> 4)) * plane;
  char* ob = (char*)a.out + ((size_t)bsafe * (a.out_ctot >> 4) + (a.out_coff >> 4)) * plane;
  const f32x4 one4 = {1.f, 1.f, 1.f, 1.f}, zero4 = {0.f, 0.f, 0.f, 0.f};
  const kb16x4 zero4h = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
  float* sl = a.slab + (((size_t)bid * 4 + wm) * (MT * 16) + 4 * g) * 2;
  // the stored activation and the partial gradient of a C_out tile are requested one tile ahead of its stores
  kb16x4 mv[2][NT], rv[2][NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    mv[0][nt] = *(const kb16x4*)(mb + po[nt]);
    rv[0][nt] = rb ? *(const kb16x4*)(rb + po[nt]) : zero4h;
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int nq = mt * 16 + 4 * g;
    const f32x4 sc = a.scale ? *(const f32x4*)(a.scale + nq) : one4;
    const f32x4 msc = a.mask_scale ? *(const f32x4*)(a.mask_scale + nq) : one4;
    const f32x4 msh = a.mask_scale ? *(const f32x4*)(a.mask_shift + nq) : zero4;
    const f32x4 ba = a.bn_a ? *(const f32x4*)(a.bn_a + nq) : zero4;
    const f32x4 bb = a.bn_a ? *(const f32x4*)(a.bn_b + nq) : zero4;
    if (mt + 1 < MT) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        mv[(mt + 1) & 1][nt] = *(const kb16x4*)(mb + ((mt + 1) * plane + po[nt]));
        rv[(mt + 1) & 1][nt] = rb ? *(const kb16x4*)(rb + ((mt + 1) * plane + po[nt])) : zero4h;
      }
    }
    f32x4 s1 = zero4, s2 = zero4;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const f32x4 mk = __builtin_convertvector(mv[mt & 1][nt], f32x4), r = __builtin_convertvector(rv[mt & 1][nt], f32x4);
      f32x4 x;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float v = acc[mt][nt][c] * sc[c] + r[c];
        if (!(fmaf(mk[c], msc[c], msh[c]) > 0.f)) v = 0.f;
        if (!ok[nt]) v = 0.f;            // (a dropped slot reads the image's first pixel: it must not feed the sums)
        x[c] = v;
        s1[c] += v;
        s2[c] = fmaf(v, fmaf(ok[nt] ? mk[c] : 0.f, ba[c], bb[c]), s2[c]);
      }
      if (ok[nt]) *(kb16x4*)(ob + (mt * plane + po[nt])) = __builtin_convertvector(x, kb16x4);
    }
    if (a.bn_a) {
#pragma unroll
      for (int c = 0; c < 4; ++c) { s1[c] = b16k_row_sum(s1[c]); s2[c] = b16k_row_sum(s2[c]); }
      if (m == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { sl[(mt * 16 + c) * 2] = s1[c]; sl[(mt * 16 + c) * 2 + 1] = s2[c]; }
      }
    }
  }
}

// Training with bf16 activation storage, the FORWARD launches of the 128-channel 3x3 / 5x5 layers that feed a BatchNorm
// (tsr_conv2d_ex, epi_mode 1; the input is a materialised activation -- tsr_bn_relu_b16 -- or a stored block output): out =
// bf16(acc), the raw bias-free conv output, and the Welford partial (mean, M2) of the fp32 accumulators over this image's valid
// pixels of the patch, per (workgroup, image, channel) into the slab, the count into slab_cnt (entry numbering as above).
template <int MT>
__device__ __forceinline__ void b16k_epilogue_train(const ConvArgs& a, f32x4 (&acc)[MT][4], int bid, int wm, int b, int y0, int x0,
                                                    int m, int g, int HW) {
  constexpr int NT = 4;
  const bool img_ok = b < a.B;
  const int bsafe = img_ok ? b : 0;
  bool ok[NT];
  unsigned po[NT];
  float cnt = 0.f;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int gy = y0 + 2 * nt + (m >> 3), gx = x0 + (m & 7);
    ok[nt] = img_ok && gy < a.H && gx < a.W;
    po[nt] = (ok[nt] ? (unsigned)(gy * a.W + gx) * 32u : 0u) + 8u * g;
    cnt += ok[nt] ? 1.f : 0.f;
  }
  cnt = b16k_row_sum(cnt);
  const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
  const unsigned plane = (unsigned)HW * 32u;
  char* ob = (char*)a.out + ((size_t)bsafe * (a.out_ctot >> 4) + (a.out_coff >> 4)) * plane;
  const size_t e = (size_t)bid * 4 + wm;
  float* sl = a.slab + (e * (MT * 16) + 4 * g) * 2;
  if (m == 0 && g == 0) a.slab_cnt[e] = cnt;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    f32x4 mean = {0.f, 0.f, 0.f, 0.f}, m2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int c = 0; c < 4; ++c) mean[c] += ok[nt] ? acc[mt][nt][c] : 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) mean[c] = b16k_row_sum(mean[c]) * inv;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float d = ok[nt] ? acc[mt][nt][c] - mean[c] : 0.f;
        m2[c] = fmaf(d, d, m2[c]);
      }
      if (ok[nt]) *(kb16x4*)(ob + (mt * plane + po[nt])) = __builtin_convertvector(acc[mt][nt], kb16x4);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) m2[c] = b16k_row_sum(m2[c]);
    if (m == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c) { sl[(mt * 16 + c) * 2] = mean[c]; sl[(mt * 16 + c) * 2 + 1] = m2[c]; }
    }
  }
}

// Halo-row requests of step t of a block (two slots; -1 = none): row r of the CURRENT block = r, of the NEXT block = 16 + r.
// Constraints (checked by hand against the circular-buffer rule in the kernel): a slot is requested only after the step
// whose tap last read its old row (old row q < KS-1 dies with kernel row q, the others with the block), at least two
// steps before the first tap that reads the new row (row r is first needed at kernel row max(0, r - 7)); next-block rows
// 0..7 no later than step T-2.
template <int KS> struct B16KJobs;
template <> struct B16KJobs<5> {       // HH = 12: next rows 0..3 -> never-used slots, 4..7 -> this block's rows 0..3, 8..11 -> its rows 4..7
  static constexpr int job(int t, int i) {
    if (i) return -1;
    if (t < 4) return 8 + t;                    // this block's rows 8..11 (first needed at taps 5, 10, 15, 20)
    if (t >= 6 && t < 10) return 16 + (t - 6);
    if (t == 11) return 16 + 4;                 // slot = row 0, dead after tap 4
    if (t == 13) return 16 + 5;                 // row 1, dead after tap 9
    if (t == 16) return 16 + 6;                 // row 2, dead after tap 14
    if (t == 21) return 16 + 7;                 // row 3, dead after tap 19
    return -1;
  }
};
template <> struct B16KJobs<3> {       // HH = 10: next rows 0..5 -> never-used slots, 6, 7 -> this block's rows 0, 1, 8, 9 -> its rows 2, 3
  static constexpr int job(int t, int i) {
    if (t == 0) return i ? -1 : 8;              // this block's rows 8, 9 (first needed at taps 3, 6)
    if (t == 1) return i ? -1 : 9;
    if (t >= 2 && t <= 4) return 16 + 2 * (t - 2) + i;
    if (t == 5) return i ? -1 : 16 + 6;         // slot = row 0, dead after tap 2
    if (t == 6) return i ? -1 : 16 + 7;         // row 1, dead after tap 5
    return -1;
  }
};

// NW = waves = images per workgroup (4: two workgroups per CU; 8: one 512-thread workgroup per CU whose weight slab serves 512
// pixels -- see B16K_LAUNCH).
// Barrier steps of a channel block.  Plain / fused: one tap per step.  Pair form: the 16 outer taps carry half the MFMAs (the 3x3
// conv has no weight there), so two CONSECUTIVE outer taps share a step -- 8 double steps + the 9 inner taps = 17 steps of 32
// MFMAs per wave instead of 25 -- and a weight slab: [k group][64 channels of the 5x5 conv at the second tap | at the first
// tap][8] (tsr_pack_conv_weight_b16k_pair), which the A-fragment reads address exactly like a full slab.
template <int KS, int MODE> struct B16KSteps {
  static constexpr int T = KS * KS;
  struct Tab { int n = 0; int first[T] = {}; int ntap[T] = {}; int step_of[T] = {}; };
  static constexpr bool outer(int t) { return b16k_pair(MODE) && b16k_tap_rows<KS, MODE, 8>(t) != 0; }
  static constexpr Tab make() {
    Tab tb;
    int t = 0;
    while (t < T) {
      const int n = (outer(t) && t + 1 < T && outer(t + 1)) ? 2 : 1;
      tb.first[tb.n] = t; tb.ntap[tb.n] = n;
      for (int i = 0; i < n; ++i) tb.step_of[t + i] = tb.n;
      ++tb.n; t += n;
    }
    return tb;
  }
  static constexpr Tab tab = make();
};

template <int KS, int COUT, int MODE, int NW>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 1 : 2) void conv_b16k_kernel(const ConvArgs a) {
  static_assert(MODE == B16K_PLAIN || MODE == B16K_DGRAD || MODE == B16K_TRAIN ||
                (COUT == 128 && (MODE == B16K_FUSED || (b16k_pair(MODE) && KS == 5))), "fused / pair: 128 channels");
  typedef B16KGeom<KS, COUT, NW> G;
  constexpr int P = KS / 2, HH = G::HH, T = G::T, MT = COUT / 16, NT = 4;
  constexpr int PIXB = G::PIXB, ROWB = G::ROWB, IMGB = G::IMGB, HALO_B = G::HALO_B, WTAP_B = G::WTAP_B;
  constexpr int WCH = WTAP_B / 1024;                         // 1-KB LDS-DMA requests per slab: wave w issues w, w + NW, ...
  constexpr int WV = (WCH + NW - 1) / NW;
  __shared__ __attribute__((aligned(16))) char lds[G::LDS_B];
  char* halo = lds;
  char* wbuf = lds + HALO_B;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, g = lane >> 4;

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
  const int y0 = ty * 8, x0 = tx * 8, b0 = ig * NW;
  const int HW = a.H * a.W;
  const int in_blocks = a.in_ctot >> 4;

  // ---- halo staging: LDS-DMA, one ROW of one image per wave instruction (wave w stages image w), no registers.
  // The slab of a 32-channel block is HH rows; the LDS holds NROW = 16 physical rows per image as a CIRCULAR buffer: row r of
  // block c lives in physical row (c * HH + r) & 15.  Tap (kh, kw) reads rows kh .. kh+7, so with kh ascending the rows of a
  // block die one kernel row at a time, and the next block's rows are needed one kernel row at a time: its rows 0 .. NROW-HH-1
  // go into slots the current block never used, the following ones into the current block's rows 0, 1, .. as they die, and
  // its last rows (first needed at kernel row 1, 2, ..) are requested during its own first steps into the slots that were
  // live until the previous block's last tap.  A row requested at step s is awaited before the barrier that ends step s+1
  // and read from step s+2 on: no block boundary, no staging registers, no conversion -- a step is a step.
  const int hpx = lane >> 2, hqd = lane & 3;
  const int hgx = x0 - P + hpx;
  const bool col_ok = hpx < HH && hgx >= 0 && hgx < a.W && b0 + wm < a.B;
  const unsigned img_stride = (unsigned)in_blocks * HW * 32;                      // bytes per image
  const int lane_base = (int)(wm * img_stride) + hgx * 32 + (hqd >> 1) * HW * 32 + (hqd & 1) * 16;
  const unsigned long long in_grp = (unsigned long long)((const char*)a.in + ((size_t)b0 * in_blocks + (a.in_coff >> 4)) * HW * 32);
  const unsigned long long w_base = (unsigned long long)a.wp, w2_base = (unsigned long long)a.w2;
  // An offset beyond the descriptor's range makes a buffer load return zeros: the zero padding (a lane whose column lies
  // outside the image carries such an offset), and -- through a descriptor of range ZERO (a scalar select on its
  // num_records word) -- a row outside the image / a request that has nothing to fetch.  Such requests are issued all the
  // same: every step then issues a compile-time number of operations, which is what its counted vmcnt wait relies on.
  // The row part of the address goes through the SCALAR offset, the lane part (column, channel quarter, image) is one
  // loop-invariant register.  (`opaque`: hipcc would otherwise precompute the per-row / per-tap address registers of a
  // whole block -- 30 VGPRs the accumulator tile does not leave.)
  const int vo_lane = col_ok ? lane_base : (int)0x80000000;
  auto opaque = [](int v) __attribute__((always_inline)) { asm volatile("" : "+s"(v)); return v; };
  const unsigned halo_a = b16k_lds_addr(halo), wbuf_a = b16k_lds_addr(wbuf);
  auto dma_row = [&](int cblk, int r, bool real) __attribute__((always_inline)) {
    const int gy = opaque(y0 - P + r);
    const bool row_ok = real && gy >= 0 && gy < a.H;
    const int prow = (cblk * HH + r) & 15;
    const unsigned dst = halo_a + wm * IMGB + prow * ROWB;
    // physical row 0 has a copy behind row 15: lanes m >= 8 of a fragment read the row AFTER the (uniform) first row
    const unsigned dst2 = (dst + 16 * ROWB) & -(unsigned)(prow == 0);  // (an LDS address of a halo row is never 0: row 16)
    // (masks, not selects: hipcc turns a select between an expression and 0 into a BRANCH, and a basic-block boundary inside a
    //  step ends its MFMA / LDS-read interleave)
    const ki32x4 rs = {(int)in_grp, (int)(in_grp >> 32) & 0xffff, 0x7fffffff & -(int)row_ok, 0x00020000};
    b16k_dma_row(rs, vo_lane, cblk * HW * 64 + gy * a.W * 32, dst, dst2, (1ull << (4 * HH)) - 1);
  };
  const int wvo = tid * 16;

  const int laneA = (g * COUT + m) * 16;                                          // + mt * 256 (+ slot)
  const int laneB = wm * IMGB + (m >> 3) * ROWB + (m & 7) * PIXB + g * 16;         // + first physical row * ROWB + kw * PIXB

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  typedef B16KSteps<KS, MODE> ST;
  constexpr int NSTEP = ST::tab.n;                           // barrier steps (= weight slabs) per channel block
  const int nblk = a.cin >> 5;
  const int S = nblk * NSTEP;

  // (fused form: the two requests past the end of the conv's stream -- their slots are dead -- fetch the two 8-KB halves of the
  //  packed 1x1 weight instead of nothing: the epilogue reads its W2 fragments from LDS)
#define DMA_W(sidx, slot_)                                                                \
  {                                                                                      \
    const int real_ = -(int)((sidx) < S);                 /* all ones / zero: masks, not selects (see dma_row) */ \
    const int w2_ = MODE == B16K_FUSED ? ~real_ & -(int)((sidx) < S + 2) : 0;            \
    const int blo_ = (int)w2_base ^ (((int)w_base ^ (int)w2_base) & real_);              \
    const int bhi_ = (int)(w2_base >> 32) ^ (((int)(w_base >> 32) ^ (int)(w2_base >> 32)) & real_); \
    const ki32x4 rs_ = {blo_, bhi_ & 0xffff, 0x7fffffff & (real_ | w2_), 0x00020000};    \
    const int so_ = (((sidx) - S) * 8192) ^ (((((sidx) - S) * 8192) ^ ((sidx) * WTAP_B)) & real_); \
    const unsigned dst_ = wbuf_a + (slot_) * WTAP_B + wm * 1024;                         \
    _Pragma("unroll") for (int v = 0; v < WV; ++v) {                                     \
      if ((v + 1) * NW <= WCH) b16k_dma(rs_, wvo, so_ + v * NW * 1024, dst_ + v * NW * 1024); \
      else b16k_dma_masked(rs_, wvo, so_ + v * NW * 1024, dst_ + v * NW * 1024, wm + v * NW < WCH); \
    }                                                                                    \
  }
  // vmcnt wait that leaves the n_ youngest vector-memory operations in flight (they count in issue order)
#define VM_WAIT(n_) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n_) & 15) | (((n_) >> 4) << 14))
  // end of a step: everything requested before this step has landed (the n_ operations of this step stay in flight), this
  // wave's LDS reads are done (the slot of the slab they read is re-requested right after the barrier), then a RAW
  // s_barrier -- __syncthreads()'s fence would drain the LDS-DMA queue (vmcnt(0)) at every step
  // (the empty asm statements are compiler-only fences: no LDS access may be moved across the wait / barrier pair)
#define STEP_END(n_)                                                                     \
  {                                                                                      \
    asm volatile("" ::: "memory");                                                       \
    __builtin_amdgcn_s_waitcnt(0x0F70 | ((n_) & 15) | (((n_) >> 4) << 14));              \
    __builtin_amdgcn_s_barrier();                                                        \
    asm volatile("" ::: "memory");                                                       \
  }
#define LOAD_A(mt_, slot_) A[mt_] = *(const kb16x8*)(wbuf + (slot_) * WTAP_B + laneA + (mt_) * 256)
  // B fragment of row pair nt_ at tap (kh_, kw_) of the block whose row 0 is physical row rb_
#define LOAD_B(nt_, rb_, kh_, kw_)                                                        \
  Bf[nt_] = *(const kb16x8*)(halo + laneB + (((rb_) + 2 * (nt_) + (kh_)) & 15) * ROWB + (kw_) * PIXB)
#define MFMA(am_, mt_, nt_) acc[am_][nt_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[mt_], Bf[nt_], acc[am_][nt_], 0, 0, 0)
#define SGB(mask_, n_) __builtin_amdgcn_sched_group_barrier(mask_, n_, 0)

  // ---- prologue: rows 0..7 of block 0, W(0), W(1) landed; W(2) in flight
#pragma unroll
  for (int r = 0; r < 8; ++r) dma_row(0, r, true);
  DMA_W(0, 0);
  DMA_W(1, 1);
  STEP_END(0);

  kb16x8 A[MT], Bf[NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) LOAD_A(mt, 0);
  LOAD_B(0, 0, 0, 0);
  LOAD_B(1, 0, 0, 0);

  int s = 0, slot = 0;
  for (int c = 0; c < nblk; ++c) {
    const bool more = c + 1 < nblk;
    const int rb0 = (c * HH) & 15;
    b16k_static_for<0, NSTEP>([&](auto sc) __attribute__((always_inline)) {
      constexpr int st = decltype(sc)::value;
      constexpr int t0 = ST::tab.first[st], NTAP = ST::tab.ntap[st];
      const int slot1 = slot == 2 ? 0 : slot + 1;
      const int slot2 = slot1 == 2 ? 0 : slot1 + 1;
      const int rb = opaque(rb0), rbn = (rb + HH) & 15;
      // slab s+2 -> the slot of slab s-1: its fragments were read during step s-2 and consumed by step s-1's MFMAs, and every
      // wave has passed the barrier that ended step s-1
      DMA_W(s + 2, slot2);
      // one tap: C_out tiles [ML, MH) of the A fragments accumulate into rows AO + mt (pair form, second tap of a double step:
      // the 5x5 conv's channels sit in A rows 0..3)
      auto tap = [&](auto tc, auto mlc, auto mhc, auto aoc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value, ML = decltype(mlc)::value, MH = decltype(mhc)::value, AO = decltype(aoc)::value;
        constexpr int kh = t / KS, kw = t - kh * KS;
        constexpr int tn = t + 1 < T ? t + 1 : 0;
        constexpr int nkh = tn / KS, nkw = tn - nkh * KS;
        constexpr int j0 = B16KJobs<KS>::job(t, 0), j1 = B16KJobs<KS>::job(t, 1);
        LOAD_B(2, rb, kh, kw);
        LOAD_B(3, rb, kh, kw);
#pragma unroll
        for (int mt = ML; mt < MH; ++mt) { MFMA(AO + mt, mt, 0); MFMA(AO + mt, mt, 1); }
        SGB(0x100, 2);
        SGB(0x008, 2 * (MH - ML));
        __builtin_amdgcn_sched_barrier(0);
        // this tap's halo requests, issued behind the first half's MFMAs (right after the barrier they would delay them)
        if constexpr (j0 >= 0) dma_row(c + (j0 >> 4), j0 & 15, (j0 >> 4) == 0 || more);
        if constexpr (j1 >= 0) dma_row(c + (j1 >> 4), j1 & 15, (j1 >> 4) == 0 || more);
        LOAD_B(0, (t + 1 < T ? rb : rbn), nkh, nkw);
        LOAD_B(1, (t + 1 < T ? rb : rbn), nkh, nkw);
#pragma unroll
        for (int mt = ML; mt < MH; ++mt) {
          MFMA(AO + mt, mt, 2);
          MFMA(AO + mt, mt, 3);
          LOAD_A(mt, slot1);              // the NEXT step's fragment of this row, refilled in place as the row retires
        }
        // order: B2 B3 | first half | requests | B0' B1' | second half with the A rows refilled as they retire
        SGB(0x100, 2);
#pragma unroll
        for (int mt = ML; mt < MH; ++mt) { SGB(0x008, 2); SGB(0x100, 1); }
      };
      constexpr int NJ = (B16KJobs<KS>::job(t0, 0) >= 0) + (B16KJobs<KS>::job(t0, 1) >= 0) +
                         (NTAP == 2 ? (B16KJobs<KS>::job(t0 + 1, 0) >= 0) + (B16KJobs<KS>::job(t0 + 1, 1) >= 0) : 0);
      typedef std::integral_constant<int, 0> I0;
      typedef std::integral_constant<int, MT / 2> IH;
      typedef std::integral_constant<int, MT> IM;
      if constexpr (NTAP == 2) {
        tap(std::integral_constant<int, t0>(), IH(), IM(), I0());           // first outer tap: A rows 4..7 -> C_out tiles 4..7
        __builtin_amdgcn_sched_barrier(0);
        tap(std::integral_constant<int, t0 + 1>(), I0(), IH(), IH());       // second: A rows 0..3 -> C_out tiles 4..7
      } else {
        tap(std::integral_constant<int, t0>(), I0(), IM(), I0());
      }
      // everything requested BEFORE this step has landed (this step's own halo requests stay in flight): slab s+2, the halo
      // rows of step s-1
      STEP_END(NJ);
      ++s;
      slot = slot1;
    });
  }
  // (the trailing requests fetch nothing but still write LDS: none may be in flight when the workgroup's LDS is released)
  VM_WAIT(0);
#undef DMA_W
#undef VM_WAIT
#undef STEP_END
#undef LOAD_A
#undef LOAD_B
#undef MFMA
#undef SGB

  if constexpr (MODE == B16K_DGRAD) {
    b16k_epilogue_dgrad<MT>(a, acc, bid, wm, b0 + wm, y0, x0, m, g, HW);
    return;
  }
  if constexpr (MODE == B16K_TRAIN || MODE == B16K_PAIR_TRAIN) {      // (pair: tsr_conv2d_ex, nsplit = -4)
    b16k_epilogue_train<MT>(a, acc, bid, wm, b0 + wm, y0, x0, m, g, HW);
    return;
  }
  // (fused form: the 1x1 weight's halves sit in the ring slots of the two requests past the stream's end, slabs S and S+1)
  b16k_epilogue<MT, MODE>(a, acc, b0 + wm, y0, x0, m, g, HW, wbuf + slot * WTAP_B, wbuf + (slot == 2 ? 0 : slot + 1) * WTAP_B);
}

// ---- weight packs ------------------------------------------------------------------------------------------------------
// OIHW fp32 -> [C_in/32][tap][k group 4][C_out][8] bf16: the A fragment of lane (m, g) for C_out tile mt is the 16 B at
// ((g * C_out + mt * 16 + m) * 8) of a tap slab; slabs follow one another in step order (LDS-DMA copies them verbatim).
__global__ void pack_b16k_kernel(const float* __restrict__ w, __bf16* __restrict__ wp, int cout, int cin, int T) {
  const size_t total = (size_t)cout * cin * T;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int j = i & 7;
    size_t r = i >> 3;
    const int co = r % cout; r /= cout;
    const int g = r & 3; r >>= 2;
    const int tap = r % T;
    const int cb = r / T;
    const int ci = cb * 32 + g * 8 + j;
    wp[i] = (__bf16)w[((size_t)co * cin + ci) * T + tap];
  }
}

// [64][128] fp32 (one half of an MSRB's `confusion` weight) -> HALVED, [kk 4][g 4][64][8] bf16 in the K order in which the fused
// epilogue's accumulators ARE the B operand: slot (g, j) of K step kk = channel 32 kk + 4 g + j (j < 4) or
// 32 kk + 16 + 4 g + (j - 4) (j >= 4)
__global__ void pack_w2_b16k_kernel(const float* __restrict__ w2, __bf16* __restrict__ wp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 64 * 128) return;
  const int j = i & 7, co = (i >> 3) & 63, g = (i >> 9) & 3, kk = i >> 11;
  const int ch = 32 * kk + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4));
  wp[i] = (__bf16)(0.5f * w2[co * 128 + ch]);       // halved (exact): the fused epilogue's ReLU is h + |h| on halves
}

// Stage-1 pair: W = cat([3x3 weight zero-padded to 5x5, 5x5 weight]) along C_out, [128][C_in][5][5] fp32 -> one 8-KB slab per
// barrier step (B16KSteps): an inner tap's slab holds all 128 channels; a double step of two outer taps (u, u+1) holds the 5x5
// conv's 64 channels twice -- rows 0..63 at tap u+1, rows 64..127 at tap u -- so that the kernel's A-fragment reads need no
// second address map
__global__ void pack_b16k_pair_kernel(const float* __restrict__ w, __bf16* __restrict__ wp, int cin) {
  typedef B16KSteps<5, B16K_PAIR> ST;
  constexpr int NSTEP = ST::tab.n;
  const size_t total = (size_t)(cin >> 5) * NSTEP * 4096;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int j = i & 7, row = (i >> 3) & 127, g = (i >> 10) & 3;
    const int st = (i >> 12) % NSTEP, cb = (i >> 12) / NSTEP;
    const int ci = cb * 32 + g * 8 + j;
    const int t0 = ST::tab.first[st];
    int co = row, tap = t0;
    if (ST::tab.ntap[st] == 2) {
      co = 64 + (row & 63);
      tap = row < 64 ? t0 + 1 : t0;
    }
    wp[i] = (__bf16)w[((size_t)co * cin + ci) * 25 + tap];
  }
}

// dgrad weight of a conv with OIHW weight w[cout_f][cin_f][ks][ks]: the packed conv is W'[n][k = co][kh][kw] =
// W[co][ci0 + n][K-1-kh][K-1-kw] (n = 0..127: the input-channel slice whose gradient the launch produces, reduction over the
// forward conv's C_out), in conv_b16k's slab layout [co/32][tap][k group][128][8]
__global__ void pack_b16k_dgrad_kernel(const float* __restrict__ w, __bf16* __restrict__ wp, int cout_f, int cin_f, int T, int ci0,
                                       int np) {
  const size_t total = (size_t)np * cout_f * T;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int j = i & 7;
    size_t r = i >> 3;
    const int n = r % np; r /= np;
    const int g = r & 3; r >>= 2;
    const int tap = r % T, kb = r / T;
    const int co = kb * 32 + g * 8 + j;
    wp[i] = (__bf16)w[((size_t)co * cin_f + ci0 + n) * T + (T - 1 - tap)];
  }
}

extern "C" long long tsr_conv_weight_b16k_elems(int cout, int cin, int ks) { return (long long)cout * cin * ks * ks; }
extern "C" long long tsr_conv_weight_b16k_pair_elems(int cin) { return (long long)(cin >> 5) * B16KSteps<5, B16K_PAIR>::tab.n * 4096; }

// NW = 4 is what ships.  NW = 8 (one 512-thread workgroup per CU: half the L2 -> LDS weight traffic, the epilogue no longer under
// another workgroup's loop) passes the same tests and measures the SAME launch times at B = 4096 (5x5 fused 3.63 vs 3.61 ms,
// 3x3 fused 1.77 vs 1.74, pair 1.63 vs 1.62, same box) -- as did halving the barriers: at ~1.5 PFLOP/s on real activations the
// 5x5 launch sits where the chip holds ~1.9 GHz under bf16 MFMA load (MI355X_MICROARCH.md, DVFS give-back: 1.25 PFLOP/s for a
// bare GEMM loop on random data), and a saved stall comes back as a lower clock.  -DTSR_B16K_NW=8 builds that form.
#ifndef TSR_B16K_NW
#define TSR_B16K_NW 4
#endif
#define B16K_LAUNCH(KS_, COUT_, MODE_)                                                                         \
  hipLaunchKernelGGL((conv_b16k_kernel<KS_, COUT_, MODE_, TSR_B16K_NW>),                                       \
                     dim3(((a.B + TSR_B16K_NW - 1) / TSR_B16K_NW) * a.tiles_x * a.tiles_y), dim3(TSR_B16K_NW * 64), 0, st, a);
static bool b16k_shape_ok(int cout, int cin, int ks) {
  return (cout == 64 || cout == 128) && cin > 0 && (cin & 31) == 0 && (ks == 3 || ks == 5);
}

extern "C" int tsr_pack_conv_weight_b16k(const float* w_oihw, void* w_packed, int cout, int cin, int ks, void* stream) {
  // (ks = 1: the forward of a 1x1 conv with 64 output channels, conv1x1_b16k.hip -- the same slab layout with one tap)
  if (!w_oihw || !w_packed || !(b16k_shape_ok(cout, cin, ks) || (ks == 1 && cout == 64 && cin > 0 && (cin & 31) == 0))) return TSR_ERR_ARG;
  const size_t total = (size_t)cout * cin * ks * ks;
  const int grid = (int)((total + 255) / 256);
  hipLaunchKernelGGL(pack_b16k_kernel, dim3(grid > 4096 ? 4096 : grid), dim3(256), 0, (hipStream_t)stream, w_oihw,
                     (__bf16*)w_packed, cout, cin, ks * ks);
  return tsr_check_launch();
}

extern "C" int tsr_pack_conv_weight_b16k_pair(const float* w128_oihw5, void* w_packed, int cin, void* stream) {
  if (!w128_oihw5 || !w_packed || !b16k_shape_ok(128, cin, 5)) return TSR_ERR_ARG;
  const size_t total = (size_t)tsr_conv_weight_b16k_pair_elems(cin);
  const int grid = (int)((total + 255) / 256);
  hipLaunchKernelGGL(pack_b16k_pair_kernel, dim3(grid > 4096 ? 4096 : grid), dim3(256), 0, (hipStream_t)stream, w128_oihw5,
                     (__bf16*)w_packed, cin);
  return tsr_check_launch();
}

// 1 if tsr_conv2d_ex accepts nsplit = -3 for a dgrad launch of this shape (bf16 tensors, conv_b16k, weights from
// tsr_pack_conv_weight_dgrad_b16k): 128 input channels per launch, 3x3 / 5x5, the forward conv's C_out a multiple of 32;
// or 1x1 with the forward conv's C_out = 64 (masked form only: epi_mode 2, no partial gradient)
extern "C" int tsr_conv2d_ex_dgrad_b16k(int nprime, int cout_f, int ks) {
  if (ks == 1) return nprime == 128 && cout_f == 64;        // the masked dgrad of a 1x1 conv (conv1x1_b16k.hip)
  return (nprime == 128 || nprime == 64) && (ks == 3 || ks == 5) && cout_f > 0 && (cout_f & 31) == 0;
}

// 1 if tsr_conv2d_ex accepts nsplit = -3 for the FORWARD of a 1x1 conv of this shape on a virtual input (in_scale / in_shift
// set; epi_mode 0: bias, residual, ReLU; weights from tsr_pack_conv_weight_b16k): conv1x1_b16k.hip
extern "C" int tsr_conv2d_ex_fwd1x1_b16k(int cout, int cin) { return cout == 64 && (cin == 256 || cin == 128); }

extern "C" int tsr_pack_conv_weight_dgrad_b16k(const float* w_oihw, void* w_packed, int cout, int cin, int ks, int ci0, int nprime,
                                               void* stream) {
  if (!w_oihw || !w_packed || !tsr_conv2d_ex_dgrad_b16k(nprime, cout, ks) || ci0 < 0 || ci0 + nprime > cin) return TSR_ERR_ARG;
  const size_t total = (size_t)nprime * cout * ks * ks;
  const int grid = (int)((total + 255) / 256);
  hipLaunchKernelGGL(pack_b16k_dgrad_kernel, dim3(grid > 4096 ? 4096 : grid), dim3(256), 0, (hipStream_t)stream, w_oihw,
                     (__bf16*)w_packed, cout, cin, ks * ks, ci0, nprime);
  return tsr_check_launch();
}

// tsr_conv2d_ex with nsplit = -3: a training launch on bf16 tensors whose input needs no transform (a stored gradient, a
// stored block output, a materialised activation): epi_mode 0 = out = act(acc * scale + shift + res) (the forward weight
// pack, or the dgrad pack for an unmasked partial gradient), 1 = raw output + Welford partials (forward pack), 2 = the masked
// dgrad (dgrad pack).  nsplit = -4 (`pair`): epi_mode 1 of the stage-1 pair (one 128-channel output, two convs).
int tsr_dgrad1x1_b16k(const ConvArgs& a, hipStream_t st);       // conv1x1_b16k.hip
int tsr_fwd1x1_b16k(const ConvArgs& a, int cout, hipStream_t st);
int tsr_conv_b16k_ex(const ConvArgs& a, int cout, int ks, bool pair, hipStream_t st) {
  // the forward of a 1x1 conv with 64 output channels on a VIRTUAL input (the only form here that takes an input transform)
  if (ks == 1 && cout == 64 && a.epi_mode == 0 && !pair) return tsr_fwd1x1_b16k(a, cout, st);
  if (!tsr_conv2d_ex_dgrad_b16k(cout, a.cin, ks) || a.in_scale || a.res_scale ||
      (long long)4 * a.in_ctot * a.H * a.W * 2 >= 0x7fffffffLL)
    return TSR_ERR_ARG;
  if (ks == 1) return pair ? TSR_ERR_ARG : tsr_dgrad1x1_b16k(a, st);
  if (pair) {      // nsplit = -4: conv_3_1 || conv_5_1 of an MSRB in train mode (weights: tsr_pack_conv_weight_b16k_pair)
    if (ks != 5 || cout != 128 || a.epi_mode != 1 || !a.slab || !a.slab_cnt) return TSR_ERR_ARG;
    B16K_LAUNCH(5, 128, B16K_PAIR_TRAIN)
    return tsr_check_launch();
  }
#define B16K_LAUNCH_C(MODE_)                                   \
  {                                                            \
    if (cout == 128) {                                         \
      if (ks == 3) B16K_LAUNCH(3, 128, MODE_)                  \
      else B16K_LAUNCH(5, 128, MODE_)                          \
    } else {                                                   \
      if (ks == 3) B16K_LAUNCH(3, 64, MODE_)                   \
      else B16K_LAUNCH(5, 64, MODE_)                           \
    }                                                          \
  }
  if (a.epi_mode == 2) {
    B16K_LAUNCH_C(B16K_DGRAD)
  } else if (a.epi_mode == 1) {
    if (!a.slab || !a.slab_cnt) return TSR_ERR_ARG;
    B16K_LAUNCH_C(B16K_TRAIN)
  } else {
    B16K_LAUNCH_C(B16K_PLAIN)
  }
#undef B16K_LAUNCH_C
  return tsr_check_launch();
}

extern "C" int tsr_pack_w2_b16k(const float* w2_64x128, void* w_packed, void* stream) {
  if (!w2_64x128 || !w_packed) return TSR_ERR_ARG;
  hipLaunchKernelGGL(pack_w2_b16k_kernel, dim3(32), dim3(256), 0, (hipStream_t)stream, w2_64x128, (__bf16*)w_packed);
  return tsr_check_launch();
}

// ---- launchers ---------------------------------------------------------------------------------------------------------
static int b16k_fill(ConvArgs& a, const void* in, int in_ctot, int in_coff, int cin, const void* w_packed, int cout,
                     const float* scale, const float* shift, const void* res, int res_ctot, int res_coff, void* out,
                     int out_ctot, int out_coff, int out_ch, int relu, int B, int H, int W) {
  if (!in || !w_packed || !out || B <= 0 || H <= 0 || W <= 0) return TSR_ERR_ARG;
  if ((in_ctot & 15) || (in_coff & 15) || (out_ctot & 15) || (out_coff & 15) || in_coff + cin > in_ctot ||
      out_coff + out_ch > out_ctot)
    return TSR_ERR_ARG;
  if (res && ((res_ctot & 15) || (res_coff & 15) || res_coff + out_ch > res_ctot)) return TSR_ERR_ARG;
  if ((long long)4 * in_ctot * H * W * 2 >= 0x7fffffffLL) return TSR_ERR_ARG;      // 32-bit halo offsets inside a 4-image group
  a = ConvArgs{};
  a.in = (const float*)in; a.in_ctot = in_ctot; a.in_coff = in_coff; a.cin = cin;
  a.wp = (const float*)w_packed; a.scale = scale; a.shift = shift;
  a.res = (const float*)res; a.res_ctot = res_ctot; a.res_coff = res_coff;
  a.out = (float*)out; a.out_ctot = out_ctot; a.out_coff = out_coff; a.relu = relu;
  a.B = B; a.H = H; a.W = W;
  a.tiles_x = (W + 7) / 8; a.tiles_y = (H + 7) / 8;
  (void)cout;
  return TSR_OK;
}

extern "C" int tsr_conv2d_fwd_b16k(const void* in, int in_ctot, int in_coff, int cin, const void* w_packed, int cout, int ks,
                                   const float* scale, const float* shift, const void* res, int res_ctot, int res_coff,
                                   void* out, int out_ctot, int out_coff, int relu, int B, int H, int W, void* stream) {
  if (!b16k_shape_ok(cout, cin, ks)) return TSR_ERR_ARG;
  ConvArgs a;
  const int rc = b16k_fill(a, in, in_ctot, in_coff, cin, w_packed, cout, scale, shift, res, res_ctot, res_coff, out, out_ctot,
                           out_coff, cout, relu, B, H, W);
  if (rc != TSR_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (cout == 64 && ks == 3) B16K_LAUNCH(3, 64, B16K_PLAIN)
  else if (cout == 64) B16K_LAUNCH(5, 64, B16K_PLAIN)
  else if (ks == 3) B16K_LAUNCH(3, 128, B16K_PLAIN)
  else B16K_LAUNCH(5, 128, B16K_PLAIN)
  return tsr_check_launch();
}

// Stage-2 convolution of an MSRB (128 -> 128, BatchNorm + ReLU) with its half of the 1x1 `confusion` fused: out / res
// describe the 64-channel result; w2_packed from tsr_pack_w2_b16k.
extern "C" int tsr_conv2d_fwd_b16k_fuse1x1(const void* in, int in_ctot, int in_coff, int cin, const void* w_packed, int ks,
                                           const float* scale, const float* shift, int relu, const void* w2_packed,
                                           const float* shift2, const void* res, int res_ctot, int res_coff, void* out,
                                           int out_ctot, int out_coff, int relu2, int B, int H, int W, void* stream) {
  if (!b16k_shape_ok(128, cin, ks) || !w2_packed) return TSR_ERR_ARG;
  ConvArgs a;
  const int rc = b16k_fill(a, in, in_ctot, in_coff, cin, w_packed, 128, scale, shift, res, res_ctot, res_coff, out, out_ctot,
                           out_coff, 64, relu, B, H, W);
  if (rc != TSR_OK) return rc;
  a.w2 = w2_packed; a.w2_inv_scale = 1.f; a.shift2 = shift2; a.relu2 = relu2;
  hipStream_t st = (hipStream_t)stream;
  if (ks == 3) B16K_LAUNCH(3, 128, B16K_FUSED)
  else B16K_LAUNCH(5, 128, B16K_FUSED)
  return tsr_check_launch();
}

// Stage-1 pair of an MSRB: w_packed = tsr_pack_conv_weight_b16k_pair(cat([zero-pad(w3 -> 5x5), w5]), cin); scale / shift
// = the two convs' folded BatchNorm vectors concatenated; out = 128 channels in torch.cat order.
extern "C" int tsr_conv2d_fwd_b16k_pair(const void* in, int in_ctot, int in_coff, int cin, const void* w_packed,
                                        const float* scale, const float* shift, void* out, int out_ctot, int out_coff,
                                        int relu, int B, int H, int W, void* stream) {
  if (!b16k_shape_ok(128, cin, 5)) return TSR_ERR_ARG;
  ConvArgs a;
  const int rc = b16k_fill(a, in, in_ctot, in_coff, cin, w_packed, 128, scale, shift, nullptr, 0, 0, out, out_ctot, out_coff,
                           128, relu, B, H, W);
  if (rc != TSR_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  B16K_LAUNCH(5, 128, B16K_PAIR)
  return tsr_check_launch();
}

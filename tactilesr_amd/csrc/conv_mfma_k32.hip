// fp16x3 implicit-GEMM convolution on v_mfma_f32_16x16x32_f16 (K = 32 per instruction), the default form of the
// 3x3 / 5x5 convolutions.  Same arithmetic as conv_mfma_split16.hip (fp32 operands as two power-of-two-scaled fp16
// planes, products h2g1 + h1g1 + h1g2 accumulated in fp32), same halo staging (slab split on the fly), same 3-slot weight
// ring (fed by LDS-DMA in the inference instantiations, through registers in the training ones), same epilogues -- a
// different matrix instruction:
//
//   * Why: under an MFMA-dense loop the MI355X lowers its clock, and it holds a HIGHER clock on the 16x16x32 shape than
//     on 32x32x16 at equal cycles per FLOP (MI355X_MICROARCH.md, DVFS give-back item 7).  Measured on this kernel's
//     predecessor with the instruction swapped in place (wrong results, same FLOPs, same LDS traffic): 1.77 -> 2.02 GHz
//     (GRBM_GUI_ACTIVE / 8 / wall), 5x5 128->128 at B = 4096 12.8 -> 11.7 ms.
//   * K = 32 = two taps x 16 channels.  The LDS images are the 32x32x16 kernel's ([pixel][plane][16 ch] halo,
//     [tap][plane][k half][C_out][8] weights); only the lane -> address maps differ: lane = (m = lane & 15, g = lane >> 4),
//     g = (tap of the pair, 8-channel half).  A rows: 16 pixels = 2 rows x 8 columns of the wave's 8x8 patch; the
//     two taps of a pair differ by a per-lane byte delta that takes two values (next column / next row), so two
//     precomputed address registers and compile-time immediates cover every step.
//   * 9 and 25 are odd: the last tap of an even channel block is paired with the last tap of the following odd block
//     (zero padding would cost 4 % / 11 % of the MFMA work).  The even block's 8x8 window of that tap is copied
//     LDS -> LDS into a small side buffer while its slab is still resident; the odd block ends with the cross step
//     [side buffer | halo].  tsr_pack_conv_weight_f16s lays the weight stream out in exactly that step order.
//   * Fragment registers are single-buffered and refilled one product phase ahead: per step the phases are
//     P0 = A1 x B0 (h2 g1), P1 = A0 x B0 (h1 g1), P2 = A0 x B1 (h1 g2); A1 is dead after P0 and reloaded (next step) during
//     P1, B0 after P1 during P2, A0 / B1 of the step itself arrive under P0.  64 accumulator + 64 fragment registers.
//
// Workgroup = 256 threads, 8x8 patch: WN = 2 -> 2 images x 2 C_out halves (C_out = 128), WN = 1 -> 4 images x all
// C_out (C_out = 64); 512 threads (training, C_out = 128): 4 images x 2 halves; every wave owns 64 pixels x 64 channels =
// 4 x 4 accumulator tiles.  Forms: plain / EXT (training epilogues) / FUSE2 (MSRB 1x1 fused, conv_fuse1x1_16.h) / DBH
// (3x3: double-buffered halo) / PAIR (MSRB stage 1: 3x3 || 5x5 on one halo).
#include "tsr_common.h"
#include "conv_args.h"
#include "conv_epilogue16.h"
#include "conv_fuse1x1_16.h"
#include "tactilesr_hip.h"
#include <type_traits>

typedef _Float16 kf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 kf16x4 __attribute__((ext_vector_type(4)));

// halo / side-buffer row stride in 16-B slots: == 2 (mod 4), which makes the ds_read_b128 A fragments of this lane map
// bank-conflict free (lane groups {0-3,12-15,20-27}, ...: pixel slots 4*col + {0, stride} + k-half {0, 1} all distinct)
constexpr int k32_row_slots(int px) {
  int rs = px * 4;
  while ((rs & 3) != 2) ++rs;
  return rs;
}

template <int KS, int COUT, int WN, bool DBH = false, int NTHR = 256> struct K32Geom {
  static constexpr int IMG = (NTHR / 64) / WN;
  static constexpr int HH = 8 + KS - 1;
  static constexpr int T = KS * KS;
  static constexpr int HS = (T - 1) / 2;                       // tap pairs per block; + the cross step per block pair
  static constexpr int PIXB = 64;
  static constexpr int ROWB = k32_row_slots(HH) * 16;
  static constexpr int IMGB = HH * ROWB;
  static constexpr int HALO_B = IMG * IMGB;
  static constexpr int SROWB = k32_row_slots(8) * 16;
  static constexpr int SIMGB = 8 * SROWB;
  static constexpr int SIDE_B = DBH ? 0 : IMG * SIMGB;         // (two resident slabs need no side buffer)
  static constexpr int WTAP_B = 2 * 16 * COUT * 2;             // [plane 2][k half 2][C_out][8] fp16
  static constexpr int WSLAB_B = 2 * WTAP_B;                   // one step = one pair of taps
  static constexpr int NHB = DBH ? 2 : 1;
  static constexpr int MAIN_LDS = NHB * HALO_B + SIDE_B + 3 * WSLAB_B;
};

// DBH (3x3): double-buffered halo.  Both blocks of a pair are resident, so the cross step reads its two taps straight
// from the two slabs (no side buffer) and sits BETWEEN the blocks: even block taps 0..7, cross (tap 8 | tap 8), odd block
// taps 0..7.  The odd block's slab is staged under the even block's steps and the next even block's under the odd
// block's: no block boundary is left in the loop (a 3x3 block is only 4.5 steps long; the boundary's barrier pair,
// conversion burst and exposed first fragment read cost 15 % there).
// NTHR = 512 (C_out = 128): 8 waves = 4 images x 2 C_out halves, one workgroup per CU.  The weight slab of a step then
// serves 256 pixels instead of 128: the weight stream (1.6 MB per workgroup at 5x5 128->128, 7 TB/s through L2 over the
// launch, i.e. the ~13 B/clk/CU a CU can pull) and its LDS writes are what the 256-thread form spends 20 % of its time
// on (ablation without weight staging: 11.8 -> 9.5 ms).
// PAIR (5x5 geometry, 128 output channels, inference): the two stage-1 convolutions of an MSRB -- 3x3 64->64 and
// 5x5 64->64 on the SAME input (model/tactileSR_model.py:167-175,198-200) -- as one launch on one staged halo.  The
// launch is a 5x5 conv to 128 channels whose 3x3 half has zero weights on the 16 outer taps; those MFMAs are not issued:
// every wave owns two n-tiles of each conv (host-side channel permutation, tsr_pair_channel_perm), tap pairs are ordered
// outer ring first (8 steps of half work), then the inner 3x3 (4 steps) with its centre-right tap (3,3) as the cross
// tap, and all pairs are horizontal or vertical neighbours (lane deltas: one pixel / one row).
constexpr int PAIR_KH[12] = {0, 0, 0, 2, 1, 3, 4, 4, 1, 1, 2, 3};
constexpr int PAIR_KW[12] = {0, 2, 4, 4, 0, 0, 1, 3, 1, 3, 1, 1};
constexpr int PAIR_V[12] = {0, 0, 1, 1, 1, 1, 0, 0, 0, 1, 0, 0};      // second tap: 0 = next column, 1 = next row
constexpr int PAIR_OUTER = 8;                                        // steps 0..7 touch only the 5x5 conv's channels

template <int KS, int COUT, bool EXT, int WN, bool FUSE2 = false, bool DBH = false, int NTHR = 256, bool PAIR = false>
__global__ __launch_bounds__(NTHR, NTHR == 512 ? 1 : 2) void conv_k32_kernel(const ConvArgs a) {
  static_assert(!PAIR || (KS == 5 && COUT == 128 && !EXT && !FUSE2 && !DBH), "pair form: 5x5 geometry, inference");
  static_assert(COUT == 64 * WN, "every wave owns 64 channels");
  static_assert(!FUSE2 || (!EXT && COUT == 128), "fused 1x1: 128 channels, inference");
  static_assert(NTHR == 256 || (NTHR == 512 && WN == 2), "512 threads: 4 images x 2 C_out halves");
  typedef K32Geom<KS, COUT, WN, DBH, NTHR> G;
  constexpr int NW = NTHR / 64;
  constexpr bool WDMA = !EXT;             // weight slabs by LDS-DMA (inference) or through registers (training)
  constexpr int IMG = G::IMG, P = KS / 2, HH = G::HH, T = G::T, HS = G::HS, NT = COUT / (16 * WN);
  constexpr int PIXB = G::PIXB, ROWB = G::ROWB, IMGB = G::IMGB, HALO_B = G::HALO_B;
  constexpr int SROWB = G::SROWB, SIMGB = G::SIMGB, SIDE_B = G::SIDE_B;
  constexpr int WTAP_B = G::WTAP_B, WSLAB_B = G::WSLAB_B;
  constexpr int WITEMS = WSLAB_B / 16, WV = (WITEMS + NTHR - 1) / NTHR;
  constexpr int NITEM = IMG * HH * HH * 4, NIT = (NITEM + NTHR - 1) / NTHR;
  constexpr int FUSE_LDS = FUSE2 ? Fuse1x1Geom16<IMG>::BYTES + 64 : 0;
  __shared__ __attribute__((aligned(16))) char lds[G::MAIN_LDS > FUSE_LDS ? G::MAIN_LDS : FUSE_LDS];
  char* halo = lds;
  char* wbuf = lds + G::NHB * HALO_B + SIDE_B;     // 3-slot ring: slab s lives in slot s % 3

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);       // wave-uniform copy (LDS-DMA base goes to M0)
  const int wm = WN == 2 ? wave >> 1 : wave, wn = WN == 2 ? wave & 1 : 0;
  const int m = lane & 15, g = lane >> 4, khalf = g & 1, tsel = g >> 1;

  float sx = 1.f, accmul = 1.f, sxh = 1.f;      // power-of-two operand scales: set in the prologue (operand_scales)
  int bid;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tpi = a.tiles_x * a.tiles_y;
  const int ig = bid / tpi;
  const int trem = bid - ig * tpi;
  const int ebase = bid * IMG;       // first statistics-slab entry of this workgroup (train epilogues)
  const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
  const int y0 = ty * 8, x0 = tx * 8, b0 = ig * IMG;
  const int HW = a.H * a.W;
  const int in_blocks = a.in_ctot >> 4;

  int st_src[NIT], st_dst[NIT];
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    const int it = tid + k * NTHR;
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
  const float* in_base = a.in + ((size_t)b0 * in_blocks + (a.in_coff >> 4)) * HW * 16;

  // A-fragment lane bases (bytes from `lds`): lx = pair in one row (second tap one pixel to the right), lw = pair that
  // wraps to the next row, lc / lcs = cross step (first tap from the side buffer, second from the halo)
  const int laneA0 = wm * IMGB + (m >> 3) * ROWB + (m & 7) * PIXB + khalf * 16;
  const int lx = laneA0 + tsel * PIXB;
  constexpr int CK = PAIR ? 3 : KS - 1;                 // cross tap (CK, CK): the last tap, or the pair form's (3,3)
  const int lw = PAIR ? laneA0 + tsel * ROWB            // pair form: vertical neighbour
                      : laneA0 + tsel * (ROWB - (KS - 1) * PIXB);
  const int lc = tsel ? laneA0 + CK * ROWB + CK * PIXB
                      : HALO_B + wm * SIMGB + (m >> 3) * SROWB + (m & 7) * PIXB + khalf * 16;
  const int lcs = tsel ? 2 * ROWB : 2 * SROWB;
  const int lcd = laneA0 + (KS - 1) * (ROWB + PIXB) + tsel * HALO_B;      // DBH cross step: last tap of slab 0 | of slab 1
  const int laneB = tsel * WTAP_B + (khalf * COUT + wn * (COUT / WN) + m) * 16;
  const int laneBh = tsel * (WTAP_B / 2) + (khalf * 64 + wn * 32 + m) * 16;      // pair form, half slabs: [plane][k half][64][8]

  f32x4 acc[4][NT];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nreal = a.cin >> 4;
  const int nchunk = nreal + (nreal & 1);     // blocks come in pairs
  const int S = (nchunk >> 1) * T;            // steps: HS per even block, HS + 1 per odd block
  const char* wsrc = (const char*)a.wp;
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, 0x7fffffff, 0x00020000);

  // BN scale / shift of this thread's channel quad of the block in flight (every staging item of a thread carries the
  // same quad): requested WITH the slab, not at the block boundary where the conversion would wait for them
  f32x4 hsc = {1.f, 1.f, 1.f, 1.f}, hsh = {0.f, 0.f, 0.f, 0.f};
  auto load_halo = [&](int c, f32x4* hv) __attribute__((always_inline)) {
    // (an odd block count is padded to even: the phantom block re-reads the last real one against zero weights)
    const float* inc = in_base + (size_t)(c < nreal ? c : nreal - 1) * HW * 16;
    if (EXT && a.in_scale) {
      const int cq = (c < nreal ? c : nreal - 1) * 16 + (tid & 3) * 4;
      hsc = *(const f32x4*)(a.in_scale + cq);
      hsh = *(const f32x4*)(a.in_shift + cq);
    }
    // branch-free: out-of-image items read a valid dummy address (offset 0) and are zeroed by their scale in store_halo
    // (hipcc otherwise wraps every load in its own exec-masked branch with a wait in front)
#pragma unroll
    for (int k = 0; k < NIT; ++k) hv[k] = *(const f32x4*)(inc + (unsigned)(st_src[k] < 0 ? 0 : st_src[k]));
  };
  auto store_halo = [&](const f32x4* hv, int c, int hb = 0) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      if (st_dst[k] >= 0) {
        f32x4 v = hv[k];
        if (EXT && a.in_scale) {   // producer's train-mode BN+ReLU, fused into the load
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) v[jj] = tsr_relu_x2(fmaf(v[jj], hsc[jj], hsh[jj]));     // 2 relu(bn(z)); sxh = sx / 2
        }
        // zero padding outside the image: a SELECT, not a multiply by 0 -- the dummy element that was loaded for a padded
        // slot may be NaN / Inf (another image's pixel) and must not leak into this image's border
        const bool inside = st_src[k] >= 0 && c < nreal;      // (the phantom block of an odd block count is staged as zeros)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) v[jj] = inside ? v[jj] * sxh : 0.f;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          kf16x4 bq;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            bq[jj] = (_Float16)v[jj];
            v[jj] -= (float)bq[jj];
          }
          *(kf16x4*)(halo + hb * HALO_B + st_dst[k] + p * 32) = bq;
        }
      }
    }
  };
  // Weight slabs travel global memory -> LDS by LDS-DMA (global_load_lds_dwordx4: 1 KB per wave instruction, no VGPR hop,
  // no ds_write): slab s+2 is requested at the START of step s into ring slot (s+2)%3 -- free since the barrier that ended
  // step s-1 -- and awaited (vmcnt 0) before the barrier that ends step s, which publishes it.  The instruction is the
  // MUBUF form (buffer_load_dwordx4 ... lds), not global_load_lds: hipcc models the FLAT-encoded one as a possible LDS
  // access through FLAT and then degrades every counted lgkmcnt wait of the step to lgkmcnt(0) -- a full LDS drain in the
  // middle of every step (whole eval forward 133.3 -> 131.9 ms with the MUBUF form).  Against register staging
  // (4 global loads + 4 ds_write_b128 per thread and step, 16 VGPRs): 5x5 128->128 11.57 -> 10.75 ms, eval forward
  // 141.4 -> 137.0 ms.
#define DMA_BYTES(goff_, slot_, nv_)                                                     \
  {                                                                                      \
    const int vo_ = (int)(goff_) + tid * 16;                                             \
    char* dst_ = wbuf + (slot_) * WSLAB_B + wave_s * 1024;                               \
    _Pragma("unroll") for (int v = 0; v < (nv_); ++v)                                    \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)(dst_ + v * NTHR * 16), 16, \
                                               vo_ + v * NTHR * 16, 0, 0, 0);             \
  }
  // Training launches (EXT) keep REGISTER staging -- slab s+3 loaded into VGPRs at the end of step s, written to LDS at
  // the end of step s+1 -- and 512-thread workgroups: same-box A/B of the train step 277.2 ms against 281.7 (LDS-DMA, 256
  // threads) / 285 (LDS-DMA, 512 threads); the inference launches gain 4 % from LDS-DMA at 256 threads (138.5 -> 132.8 ms).
#define LOAD_W(sidx)                                                                     \
  {                                                                                      \
    const f32x4* src_ = (const f32x4*)(wsrc + (size_t)(sidx) * WSLAB_B);                  \
    _Pragma("unroll") for (int v = 0; v < WV; ++v) wreg[v] = src_[tid + v * NTHR];        \
  }
#define STORE_W(slot)                                                                    \
  {                                                                                      \
    char* wb_ = wbuf + (slot) * WSLAB_B;                                                 \
    _Pragma("unroll") for (int v = 0; v < WV; ++v) ((f32x4*)wb_)[tid + v * NTHR] = wreg[v]; \
  }
#define DMA_W(sidx, slot) DMA_BYTES((size_t)(sidx) * WSLAB_B, slot, WV)
  // wait for the slab only: the n_ YOUNGEST vector-memory operations (the next block's halo loads, issued after the
  // slab request of the same step) may stay in flight -- vmcnt counts in issue order
#define DMA_WAIT_N(n_) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n_) & 15) | (((n_) >> 4) << 14))
#define DMA_WAIT() DMA_WAIT_N(0)
  // pair form: the slabs of the outer-ring steps hold the 5x5 conv's 64 channels only (half size); the stream is walked
  // with a running offset.  pq_ = index of the step within its block pair (0..24), compile time.
#define PAIR_HALF(pq_) ((pq_) < 24 && ((pq_) % 12) < PAIR_OUTER)
#define DMA_WP(slot, pq_)                                                                \
  {                                                                                      \
    if (PAIR_HALF(pq_)) { DMA_BYTES(woff, slot, WV / 2); woff += WSLAB_B / 2; }          \
    else { DMA_BYTES(woff, slot, WV); woff += WSLAB_B; }                                 \
  }
#define STEP_BARRIER() __syncthreads()
  // fragments of pair step st_ (taps 2 st_, 2 st_ + 1 of the resident block), plane p_
#define LOAD_A(dst, p_, st_, hb_)                                                        \
  {                                                                                      \
    const int kh_ = PAIR ? PAIR_KH[(st_)] : (2 * (st_)) / KS, kw_ = PAIR ? PAIR_KW[(st_)] : (2 * (st_)) - kh_ * KS; \
    const bool x_ = PAIR ? PAIR_V[(st_)] == 0 : kw_ < KS - 1;                            \
    const char* ab_ = lds + (x_ ? lx : lw) + (hb_) * HALO_B + kh_ * ROWB + kw_ * PIXB + (p_) * 32; \
    _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) dst[mt] = *(const kf16x8*)(ab_ + mt * 2 * ROWB); \
  }
#define LOAD_A_CROSS(dst, p_)                                                            \
  {                                                                                      \
    _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) dst[mt] = *(const kf16x8*)(lds + lc + mt * lcs + (p_) * 32); \
  }
#define LOAD_A_CROSSD(dst, p_)                                                           \
  {                                                                                      \
    _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) dst[mt] = *(const kf16x8*)(lds + lcd + mt * 2 * ROWB + (p_) * 32); \
  }
#define LOAD_B(dst, p_, slot_)                                                           \
  {                                                                                      \
    const char* wb_ = wbuf + (slot_) * WSLAB_B + laneB + (p_) * (2 * COUT * 16);         \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) dst[nt] = *(const kf16x8*)(wb_ + nt * 256); \
  }
  // pair form, outer-ring steps: only n-tiles 2, 3 (the 5x5 conv's channels of this wave)
#define LOAD_B_HI(dst, p_, slot_)                                                        \
  {                                                                                      \
    const char* wb_ = wbuf + (slot_) * WSLAB_B + laneBh + (p_) * (2 * 64 * 16);          \
    _Pragma("unroll") for (int nt = 2; nt < NT; ++nt) dst[nt] = *(const kf16x8*)(wb_ + (nt - 2) * 256); \
  }
#define MFMA_PHASE_HI(A_, B_)                                                            \
  _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                       \
    _Pragma("unroll") for (int nt = 2; nt < NT; ++nt)                                    \
      acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A_[mt], B_[nt], acc[mt][nt], 0, 0, 0);
#define MFMA_PHASE(A_, B_)                                                               \
  _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                       \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                    \
      acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A_[mt], B_[nt], acc[mt][nt], 0, 0, 0);
  // interleave hint: n_ groups of (per_ MFMAs, 1 LDS read)
#define INTERLEAVE(n_, per_)                                                             \
  _Pragma("unroll") for (int i_ = 0; i_ < (n_); ++i_) {                                  \
    __builtin_amdgcn_sched_group_barrier(0x008, (per_), 0);                              \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                   \
  }

  // ---- prologue: halo(0), W(0), W(1) in LDS; W(2) in flight
  // power-of-two operand scales (see conv_mfma_split16.hip).  Called AFTER the first slab's global loads have been
  // issued: its own dependent loads (amax scalars, the BN vectors' bound) and barrier run under their latency -- with one
  // 512-thread workgroup per CU nothing else hides a serial prologue.
  auto operand_scales = [&]() __attribute__((always_inline)) {
    float mx = a.in_amax ? *a.in_amax : 0.f;
    if (EXT && a.in_scale) {
      __shared__ float bnd[2 * NW];
      float ms = 0.f, mt = 0.f;
      for (int c = threadIdx.x; c < a.cin; c += NTHR) {
        ms = fmaxf(ms, fabsf(a.in_scale[c]));
        mt = fmaxf(mt, fabsf(a.in_shift[c]));
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        ms = fmaxf(ms, __shfl_xor(ms, o));
        mt = fmaxf(mt, __shfl_xor(mt, o));
      }
      if (lane == 0) { bnd[wave * 2] = ms; bnd[wave * 2 + 1] = mt; }
      __syncthreads();
      ms = 0.f; mt = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) { ms = fmaxf(ms, bnd[2 * w]); mt = fmaxf(mt, bnd[2 * w + 1]); }
      mx = mx * ms + mt;
    }
    if (mx > 0.f) {
      int e = (int)((__float_as_uint(mx) >> 23) & 0xFF) - 127;
      int be = 13 - e + 127;
      be = be < 1 ? 1 : (be > 254 ? 254 : be);
      sx = __uint_as_float((unsigned)be << 23);
    }
    float w_inv = a.w_inv_scale;
    if (a.w_amax) {
      const float wmx = *a.w_amax;
      w_inv = 1.f;
      if (wmx > 0.f && wmx < 3.0e38f) {
        int be = 127 - (13 - ((int)((__float_as_uint(wmx) >> 23) & 0xFF) - 127));
        be = be < 1 ? 1 : (be > 254 ? 254 : be);
        w_inv = __uint_as_float((unsigned)be << 23);
      }
    }
    accmul = w_inv / sx;
    sxh = (EXT && a.in_scale) ? 0.5f * sx : sx;     // staging scale (virtual inputs arrive as 2 relu(.))
  };

  f32x4 hv[NIT], wreg[WDMA ? 1 : WV];
  size_t woff = 0;                        // pair form: byte offset of the next slab to request
  load_halo(0, hv);
  if constexpr (PAIR) {                   // S >= 25; steps 0, 1 are outer-ring steps
    DMA_WP(0, 0);
    DMA_WP(1, 1);
    operand_scales();
  } else if constexpr (WDMA) {
    DMA_W(0, 0);
    if (S > 1) DMA_W(1, 1);
    operand_scales();
  } else {
    LOAD_W(0);
    operand_scales();
    STORE_W(0);
    if (S > 1) { LOAD_W(1); STORE_W(1); }
  }
  store_halo(hv, 0);
  if constexpr (WDMA) { DMA_WAIT(); } else { if (S > 2) LOAD_W(2); }
  __syncthreads();

  kf16x8 A0[4], A1[4], B0[NT] = {}, B1[NT] = {};
  LOAD_A(A1, 1, 0, 0);
  if constexpr (PAIR) { LOAD_B_HI(B0, 0, 0); } else { LOAD_B(B0, 0, 0); }

  // side-buffer copy of the last tap's 8x8 window (even blocks): 256 16-B items per image, SCI images per thread
  constexpr int SCI = IMG * 256 / NTHR, SCS = NTHR / 256;       // thread t: images (t >> 8) + SCS * k
  const int sc_t = tid & 255, sc_i0 = tid >> 8;
  const int sc_src = sc_i0 * IMGB + ((sc_t >> 5) + CK) * ROWB + (((sc_t >> 2) & 7) + CK) * PIXB + (sc_t & 3) * 16;
  const int sc_dst = HALO_B + sc_i0 * SIMGB + (sc_t >> 5) * SROWB + ((sc_t >> 2) & 7) * PIXB + (sc_t & 3) * 16;

  int s = 0;
  int slot = 0;                            // s % 3, kept incrementally
  auto block = [&](int c, auto oddc) __attribute__((always_inline)) {
    constexpr int ODD = decltype(oddc)::value;
    constexpr int NST = HS + ODD;
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      const int slot1 = slot == 2 ? 0 : slot + 1;
      const int slot2 = slot1 == 2 ? 0 : slot1 + 1;
      const bool cross = ODD && st == HS;
      f32x4 sidev[SCI];
      if (WDMA && s + 2 < S) {                                     // slab s+2 -> the slot last read a barrier ago
        if constexpr (PAIR) { DMA_WP(slot2, (ODD * HS + st + 2) % 25); } else { DMA_W(s + 2, slot2); }
      }
      const bool outer = PAIR && st < PAIR_OUTER;                  // half-work step: the 5x5 conv's n-tiles only
      const bool next_outer = PAIR && (st + 1 < PAIR_OUTER || st + 1 == NST);   // (a block starts on the outer ring)
      // this step's second-phase operands
      if (cross) { LOAD_A_CROSS(A0, 0); } else { LOAD_A(A0, 0, st, 0); }
      if (outer) { LOAD_B_HI(B1, 1, slot); } else { LOAD_B(B1, 1, slot); }
      if (!ODD && st == 0) {
#pragma unroll
        for (int k = 0; k < SCI; ++k) sidev[k] = *(const f32x4*)(lds + sc_src + k * SCS * IMGB);
      }
      if (st == NST - 2 && c + 1 < nchunk) load_halo(c + 1, hv);   // next block's slab: in flight for a step and a half
      if (outer) {
        MFMA_PHASE_HI(A1, B0);
        if (!ODD && st == 0) { INTERLEAVE(6 + SCI, 1); } else { INTERLEAVE(6, 1); }
      } else {
        MFMA_PHASE(A1, B0);                                        // P0: h2 g1
        if (!ODD && st == 0) { INTERLEAVE(8 + SCI, 1); } else { INTERLEAVE(8, 2); }
      }
      if (st + 1 < NST) {                                          // next step's A1 (same block: the slab is resident)
        if (ODD && st + 1 == HS) { LOAD_A_CROSS(A1, 1); } else { LOAD_A(A1, 1, st + 1, 0); }
      }
      if (outer) {
        MFMA_PHASE_HI(A0, B0);                                     // P1: h1 g1
        if (st + 1 < NST) { INTERLEAVE(4, 2); }
      } else {
        MFMA_PHASE(A0, B0);
        if (st + 1 < NST) { INTERLEAVE(4, 4); }
      }
      // next step's B0: published one barrier ago (after the last step: stale, unused)
      if (next_outer) { LOAD_B_HI(B0, 0, slot1); } else { LOAD_B(B0, 0, slot1); }
      if (outer) {
        MFMA_PHASE_HI(A0, B1);                                     // P2: h1 g2
        if (next_outer) { INTERLEAVE(2, 4); } else { INTERLEAVE(4, 2); }
      } else {
        MFMA_PHASE(A0, B1);
        if (next_outer) { INTERLEAVE(2, 8); } else { INTERLEAVE(4, 4); }
      }
      if (!ODD && st == 0) {
#pragma unroll
        for (int k = 0; k < SCI; ++k) *(f32x4*)(lds + sc_dst + k * SCS * SIMGB) = sidev[k];
      }
      if constexpr (WDMA) {
        if (st == NST - 2 && c + 1 < nchunk) { DMA_WAIT_N(NIT); } else { DMA_WAIT(); }
      } else {
        if (s + 2 < S) STORE_W(slot2);           // slot (s+2)%3 was last read one barrier ago
        if (s + 3 < S) LOAD_W(s + 3);
      }
      STEP_BARRIER();
      if (st + 1 == NST && c + 1 < nchunk) {
        store_halo(hv, c + 1);     // every wave is past its last read of the old slab (barrier above)
        __syncthreads();
        LOAD_A(A1, 1, 0, 0);
      }
      ++s;
      slot = slot1;
    }
  };
  if constexpr (!DBH) {
    for (int c = 0; c < nchunk; c += 2) {
      block(c, std::integral_constant<int, 0>());
      block(c + 1, std::integral_constant<int, 1>());
    }
  } else {
    for (int c = 0; c < nchunk; c += 2) {
#pragma unroll
      for (int p = 0; p < T; ++p) {          // even block: p < HS (slab 0); cross: p == HS; odd block: p > HS (slab 1)
        const int slot1 = slot == 2 ? 0 : slot + 1;
        const int slot2 = slot1 == 2 ? 0 : slot1 + 1;
        if (WDMA && s + 2 < S) DMA_W(s + 2, slot2);
        if (p == HS) { LOAD_A_CROSSD(A0, 0); }
        else if (p < HS) { LOAD_A(A0, 0, p, 0); }
        else { LOAD_A(A0, 0, p - HS - 1, 1); }
        LOAD_B(B1, 1, slot);
        if (p == 0) load_halo(c + 1, hv);                          // the odd block's slab (the block count is even)
        if (p == HS + 1 && c + 2 < nchunk) load_halo(c + 2, hv);   // the next pair's even block
        MFMA_PHASE(A1, B0);
        INTERLEAVE(8, 2);
        if (p + 1 == HS) { LOAD_A_CROSSD(A1, 1); }
        else if (p + 1 < HS) { LOAD_A(A1, 1, p + 1, 0); }
        else if (p + 1 < T) { LOAD_A(A1, 1, p - HS, 1); }
        else { LOAD_A(A1, 1, 0, 0); }                              // next pair's first step (after the last pair: unused)
        MFMA_PHASE(A0, B0);
        INTERLEAVE(4, 4);
        LOAD_B(B0, 0, slot1);
        MFMA_PHASE(A0, B1);
        INTERLEAVE(4, 4);
        if (p == HS - 2) store_halo(hv, c + 1, 1);                 // slab 1: last read in the previous pair
        if (p == T - 2 && c + 2 < nchunk) store_halo(hv, c + 2, 0);  // slab 0: last read by the cross step
        if constexpr (WDMA) {
          if (p == 0 || (p == HS + 1 && c + 2 < nchunk)) { DMA_WAIT_N(NIT); } else { DMA_WAIT(); }
        } else {
          if (s + 2 < S) STORE_W(slot2);
          if (s + 3 < S) LOAD_W(s + 3);
        }
        STEP_BARRIER();
        ++s;
        slot = slot1;
      }
    }
  }
#undef DMA_W
#undef LOAD_W
#undef STORE_W
#undef DMA_WP
#undef DMA_BYTES
#undef DMA_WAIT
#undef DMA_WAIT_N
#undef PAIR_HALF
#undef STEP_BARRIER
#undef LOAD_A
#undef LOAD_A_CROSS
#undef LOAD_A_CROSSD
#undef LOAD_B
#undef LOAD_B_HI
#undef MFMA_PHASE_HI
#undef MFMA_PHASE
#undef INTERLEAVE

  if constexpr (FUSE2) conv_fuse1x1_epilogue16<IMG>(a, acc, lds, b0, y0, x0, wm, wn, lane, HW, accmul);
  else conv_epilogue16<COUT, EXT, WN>(a, acc, ebase, b0, y0, x0, wm, wn, lane, HW, accmul);
}

// ---- launchers (called from conv_mfma_split16.hip's dispatchers; argument checks were done there) -------------------
// Images per workgroup of the plain / training launches (= statistics-slab entries per workgroup: the slab numbering of
// the train epilogues follows it, tsr_conv2d_slab_entries_ex asks here): 4 for both widths -- C_out = 128 training
// launches are 512-thread workgroups (4 images x 2 C_out halves), C_out = 64 ones 256 threads (4 images x 64 channels).
int tsr_conv_k32_images(int cout) { (void)cout; return 4; }

template <int KS, int COUT, bool EXT>
static int launch_k32(const ConvArgs& a, hipStream_t st) {
  // (C_out = 64 as 2 images x 2 halves of 32 channels, 3 workgroups per CU, measured no better than the 4-image form:
  // 5x5 3.41 vs 3.39 ms, 3x3 1.60 vs 1.68 ms at B = 4096; the 32x32x16 kernel does 3.23 / 1.61 ms)
  constexpr int WN = COUT / 64;
  if constexpr (COUT == 128 && EXT) {      // training: 512 threads (a weight slab serves 256 pixels) + register staging
    const int grid = ((a.B + 3) / 4) * a.tiles_x * a.tiles_y;
    hipLaunchKernelGGL((conv_k32_kernel<KS, COUT, EXT, WN, false, KS == 3, 512>), dim3(grid), dim3(512), 0, st, a);
    return tsr_check_launch();
  }
  constexpr int IMG = 4 / WN;              // inference: 256 threads + LDS-DMA weight ring
  const int grid = ((a.B + IMG - 1) / IMG) * a.tiles_x * a.tiles_y;
  hipLaunchKernelGGL((conv_k32_kernel<KS, COUT, EXT, WN, false, KS == 3>), dim3(grid), dim3(256), 0, st, a);
  return tsr_check_launch();
}

int tsr_conv_k32(const ConvArgs& a, int cout, int ks, bool ext, hipStream_t st) {
  if (ext) {
    if (cout == 64 && ks == 3) return launch_k32<3, 64, true>(a, st);
    if (cout == 64 && ks == 5) return launch_k32<5, 64, true>(a, st);
    if (cout == 128 && ks == 3) return launch_k32<3, 128, true>(a, st);
    if (cout == 128 && ks == 5) return launch_k32<5, 128, true>(a, st);
  } else {
    if (cout == 64 && ks == 3) return launch_k32<3, 64, false>(a, st);
    if (cout == 64 && ks == 5) return launch_k32<5, 64, false>(a, st);
    if (cout == 128 && ks == 3) return launch_k32<3, 128, false>(a, st);
    if (cout == 128 && ks == 5) return launch_k32<5, 128, false>(a, st);
  }
  return TSR_ERR_ARG;
}

// ---- stage-1 pair: channel order and weight pack ---------------------------------------------------------------------
// Kernel channel k = wn*64 + nt*16 + c (wave half wn, n-tile nt): n-tiles 0,1 carry the 3x3 conv's channels
// wn*32 + nt*16 + c, n-tiles 2,3 the 5x5 conv's channels wn*32 + (nt-2)*16 + c.  perm[k] = the channel of
// torch.cat([conv3, conv5], 1) (model/tactileSR_model.py:200) that kernel channel k holds.
__host__ __device__ inline int pair_logical_channel(int k) {
  const int wn = k >> 6, nt = (k >> 4) & 3, c = k & 15;
  return (nt < 2 ? 0 : 64) + wn * 32 + (nt & 1) * 16 + c;
}
extern "C" int tsr_pair_channel_perm(int* perm128) {      // host array
  if (!perm128) return TSR_ERR_ARG;
  for (int k = 0; k < 128; ++k) perm128[k] = pair_logical_channel(k);
  return TSR_OK;
}

// w3: [64][cin][3][3], w5: [64][cin][5][5] fp32 -> the K = 32 kernel's stream for the pair form: per pair of channel
// blocks 12 tap pairs of the even block, 12 of the odd block (PAIR_KH / PAIR_KW / PAIR_V order), the cross pair (3,3);
// each tap slab [plane 2][k half 2][128 kernel channels][8] fp16, scaled by wscale (or by the power of two derived from
// *w_amax); the 3x3 conv's outer-ring entries are zero.
__global__ void pack_pair_kernel(const float* __restrict__ w3, const float* __restrict__ w5, _Float16* __restrict__ wp,
                                 int cin, float wscale, const float* __restrict__ w_amax) {
  if (w_amax) {
    const float wm = *w_amax;
    wscale = 1.f;
    if (wm > 0.f && wm < 3.0e38f) {
      int be = 127 + 13 - ((int)((__float_as_uint(wm) >> 23) & 0xFF) - 127);
      be = be < 1 ? 1 : (be > 254 ? 254 : be);
      wscale = __uint_as_float((unsigned)be << 23);
    }
  }
  const int cin_p = (cin + 31) & ~31;
  const size_t total = (size_t)128 * cin_p * 25;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int j = i & 7;
    size_t r = i >> 3;
    const int k = r & 127; r >>= 7;
    const int kh2 = r & 1; r >>= 1;
    const int tap = r % 25;
    const int chunk = r / 25;
    const int ci = chunk * 16 + kh2 * 8 + j;
    const int kh = tap / 5, kw = tap - kh * 5;
    const int nt = (k >> 4) & 3, ch = (k >> 6) * 32 + (nt & 1) * 16 + (k & 15);
    float v = 0.f;
    if (ci < cin) {
      if (nt >= 2) v = w5[((size_t)ch * cin + ci) * 25 + tap];
      else if (kh >= 1 && kh <= 3 && kw >= 1 && kw <= 3) v = w3[((size_t)ch * cin + ci) * 9 + (kh - 1) * 3 + (kw - 1)];
    }
    v *= wscale;
    // position of the tap in the block's step order
    int pos = 24;                                   // (3,3): the cross tap
#pragma unroll
    for (int st = 0; st < 12; ++st) {
      if (kh == PAIR_KH[st] && kw == PAIR_KW[st]) pos = 2 * st;
      if (kh == PAIR_KH[st] + PAIR_V[st] && kw == PAIR_KW[st] + 1 - PAIR_V[st]) pos = 2 * st + 1;
    }
    // element offset of the tap slab: per block pair [even block | odd block | cross step]; a block = 8 outer-ring steps
    // of 2 x 2048 elements (the 5x5 conv's 64 channels only) + 4 inner steps of 2 x 4096; the cross step 2 x 4096
    const int odd = chunk & 1;
    const bool half = pos < 2 * PAIR_OUTER;
    if (half && nt < 2) continue;                 // the 3x3 conv has no weight on the outer ring
    size_t base = (size_t)(chunk >> 1) * (2 * 65536 + 8192);
    if (pos >= 24) base += 2 * 65536 + (size_t)odd * 4096;
    else if (half) base += (size_t)odd * 65536 + (size_t)(pos >> 1) * 4096 + (size_t)(pos & 1) * 2048;
    else base += (size_t)odd * 65536 + 32768 + (size_t)((pos - 16) >> 1) * 8192 + (size_t)(pos & 1) * 4096;
    const int nch = half ? 64 : 128;
    const int kk = half ? (k >> 6) * 32 + (nt - 2) * 16 + (k & 15) : k;
    for (int p = 0; p < 2; ++p) {
      const _Float16 q = (_Float16)v;
      v -= (float)q;
      wp[base + ((size_t)(p * 2 + kh2) * nch + kk) * 8 + j] = q;
    }
  }
}

extern "C" long long tsr_conv_weight_pair_elems(int cin) { return (long long)(((cin + 31) & ~31) / 32) * (2 * 65536 + 8192); }

extern "C" int tsr_pack_conv_weight_pair_f16s(const float* w3_oihw, const float* w5_oihw, void* w_packed, int cin,
                                              float wscale, const float* w_amax, void* stream) {
  if (!w3_oihw || !w5_oihw || !w_packed || cin <= 0 || (cin & 15) || (!w_amax && !(wscale > 0.f))) return TSR_ERR_ARG;
  const size_t total = (size_t)128 * ((cin + 31) & ~31) * 25;
  const int grid = (int)((total + 255) / 256);
  hipLaunchKernelGGL(pack_pair_kernel, dim3(grid > 4096 ? 4096 : grid), dim3(256), 0, (hipStream_t)stream, w3_oihw,
                     w5_oihw, (_Float16*)w_packed, cin, wscale, w_amax);
  return tsr_check_launch();
}

extern "C" int tsr_conv2d_fwd_f16s_pair(const float* in, int in_ctot, int in_coff, int cin, const void* w_packed,
                                        float w_inv_scale, const float* in_amax, float* out_amax,
                                        const float* scale, const float* shift, float* out, int out_ctot, int out_coff,
                                        int relu, int B, int H, int W, void* stream) {
  if (!in || !w_packed || !out || !in_amax || B <= 0 || H <= 0 || W <= 0 || !(w_inv_scale > 0.f)) return TSR_ERR_ARG;
  if ((cin & 15) || (in_ctot & 15) || (in_coff & 15) || (out_ctot & 15) || (out_coff & 15) || cin <= 0 ||
      in_coff + cin > in_ctot || out_coff + 128 > out_ctot)
    return TSR_ERR_ARG;
  ConvArgs a = {};
  a.in = in; a.in_ctot = in_ctot; a.in_coff = in_coff; a.cin = cin;
  a.wp = (const float*)w_packed; a.scale = scale; a.shift = shift;
  a.out = out; a.out_ctot = out_ctot; a.out_coff = out_coff; a.relu = relu;
  a.B = B; a.H = H; a.W = W;
  a.tiles_x = (W + 7) / 8; a.tiles_y = (H + 7) / 8;
  a.in_amax = in_amax; a.w_inv_scale = w_inv_scale; a.out_amax = out_amax;
  const int grid = ((B + 1) / 2) * a.tiles_x * a.tiles_y;
  hipLaunchKernelGGL((conv_k32_kernel<5, 128, false, 2, false, false, 256, true>), dim3(grid), dim3(256), 0,
                     (hipStream_t)stream, a);
  return tsr_check_launch();
}

int tsr_conv_k32_fuse1x1(const ConvArgs& a, int ks, hipStream_t st) {
  // The fused forms stay at 256 threads (2 workgroups per CU): with 512 threads the parked tile needs 139 KB of LDS, one
  // workgroup per CU, and the fused epilogue is no longer hidden behind a second workgroup's main loop -- same-box A/B of
  // the whole eval forward: 145.2 ms (512) vs 144.6 ms (256).
  const int grid = ((a.B + 1) / 2) * a.tiles_x * a.tiles_y;
  if (ks == 5) hipLaunchKernelGGL((conv_k32_kernel<5, 128, false, 2, true>), dim3(grid), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((conv_k32_kernel<3, 128, false, 2, true, true>), dim3(grid), dim3(256), 0, st, a);
  return tsr_check_launch();
}

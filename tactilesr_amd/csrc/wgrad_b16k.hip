// Weight gradient of the bf16 ACTIVATION-STORAGE training path (BASELINE's "bf16" configurations; reference: autograd's
// conv2d weight / bias gradients behind train/tactileSR_train.py:215-228) on v_mfma_f32_16x16x32_bf16, staged by LDS-DMA:
//
//     dW[co][ci][kh][kw] = sum over (image, y, x) of dz[co][y][x] * a[ci][y+kh-P][x+kw-P]        db[co] = sum of dz[co]
//
// GEMM per tap: M = co (A operand = dz), N = ci (B operand = a), K = pixels; K = 32 = one 4-row x 8-column work item, as
// in wgrad_mfma_tr16.hip, whose tiles (C_out x C_in x kernel rows per workgroup), batch splits and slab layout this kernel
// keeps -- it replaces wgrad_tr16_kernel<.., IO16 = true> for launches WITHOUT a fused input transform.  What differs:
//
//   * Both tensors are bf16 CB16 already, i.e. a pixel of a 16-channel block is 32 contiguous bytes in HBM and is wanted
//     in exactly that form in LDS (the MFMA fragments come out of ds_read_b64_tr_b16 transposed): staging is
//     buffer_load_dwordx4 ... lds -- no staging registers, no VALU, no ds_write.  A work item's slot is filled by NV
//     wave-wide requests per wave (1 KB each, 16 B per lane): [dz: C_out/16 blocks x 32 pixels] [a: C_in/16 blocks x
//     (4 + KHW - 1) rows x PITCH pixels].  Pixels outside the image (zero padding, ragged right / bottom patches) carry a
//     lane offset beyond the descriptor's range and arrive as zeros; requests past the split's last item use a descriptor
//     of range 0 (every step issues the same number of operations: its vmcnt wait counts them).
//   * Ring of 3 slots, two items of flight: item s+2 is requested at the top of step s into the slot item s-1 left (every
//     wave has passed the barrier behind step s-1), item s+1 is awaited (vmcnt(NV): all but this step's requests) in front of
//     the raw s_barrier that ends step s.  One barrier per item, none of them drains the request queue.
//   * K order: K slot (lane group g, read j, element e) is pixel 16 j + 4 g + e of the item (row-major 4 x 8), so the
//     two 16-lane groups of a half-wave read 256 contiguous bytes of a dz block and, for `a` (rows pitched PITCH = 12
//     pixels = 384 B), two 128-B runs 128 B apart (same row) -- conflict-free for every tap; a tap (kh, kw) is an
//     immediate offset of the B fragment's address.
//   * The bias gradient (workgroups cib = 0, kg = 0) is summed from the A fragments: wave (wm, wn) sums C_out tile wn of its
//     wm, 16 VALU operations per item.
//   * XF instantiations (the 1x1 `confusion`: one workgroup = 64 x 256 outputs) DO take a fused input transform: the raw z
//     tile lands in LDS like any other and every thread rewrites its share in place -- bf16(relu(z * scale + shift)), the value
//     the register-staging kernel formed -- one step AHEAD of the step that multiplies the item (4-slot ring).  The 3x3 / 5x5
//     launches of a virtual input read its materialised form instead (tsr_bn_relu_b16, at the end of this file): that tensor is
//     shared with the forward convolutions of the same stage.
#include "tsr_common.h"
#include "tactilesr_hip.h"
#include <type_traits>

typedef __bf16 gb16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gb16x4 __attribute__((ext_vector_type(4)));
typedef short gv4i16 __attribute__((__vector_size__(4 * sizeof(short))));
typedef int gi32x4 __attribute__((ext_vector_type(4)));

struct WgradBArgs {
  const char* a;  int a_ctot; int a_coff; int cin;
  const float* a_scale; const float* a_shift;  // XF kernels: the input is relu(a * scale + shift), formed in LDS
  const char* dz; int dz_ctot; int dz_coff; int cout;
  float* slab; float* bslab;
  int B, H, W, nsplit;
  int tiles_x, tiles_y;                        // 8-column x 4-row patches
};

// LDS-DMA request (see conv_b16k.hip): 16 B per lane from rs.base + vo to LDS m0v + 16 * lane; M0 saved / restored
__device__ __forceinline__ void wgb_dma(gi32x4 rs, int vo, unsigned m0v) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(m0v), "v"(vo), "s"(rs));
}

#ifndef TSR_WGB_PF
#define TSR_WGB_PF 1
#endif

template <int KS, int KHW, int CO, int CI, int WM, bool XF = false>
struct WgradBGeom {
  static constexpr int NWM = CO / WM, NWN = CI / 32, NW = NWM * NWN, NT = 64 * NW;
  static constexpr int MT = WM / 16;
  static constexpr int NKG = (KS + KHW - 1) / KHW;          // kernel-row groups (one workgroup each)
  static constexpr int NTAP = KHW * KS;
  static constexpr int AROWS = 4 + KHW - 1, ACOLS = 8 + KS - 1, PITCH = KS == 1 ? 8 : 12;
  // 16-B units of a slot: dz blocks (64 each), then `a` blocks (UPB each)
  static constexpr int UDZ = (CO / 16) * 64, UPB = AROWS * PITCH * 2, UA = (CI / 16) * UPB, U = UDZ + UA;
  static constexpr int NV = ((U + 63) / 64 + NW - 1) / NW;  // requests per wave and item
  // XF: one more slot -- an item is transformed in place during the step BEFORE the one that multiplies it
  static constexpr int SLOTB = NV * NW * 1024, RING = XF ? 4 : 3, LA = RING - 1, LDSB = RING * SLOTB;
  static constexpr int NXF = (UA + NT - 1) / NT;              // 16-B units of the input tile a thread transforms per item
  static constexpr bool DZ0 = CO / 16 == NW;                  // request 0 of EVERY wave is its dz block (no run-time kinds)
  static_assert(MT <= NWN, "bias sums: one C_out tile per wave");
  static_assert(ACOLS <= PITCH && LDSB <= 160 * 1024, "slot");
};

template <int I, int N, class F> __device__ __forceinline__ void wgb_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>());
    wgb_static_for<I + 1, N>(f);
  }
}

template <int KS, int KHW, int CO, int CI, int WM, bool XF = false>
__global__ __launch_bounds__((CO / WM) * (CI / 32) * 64, (CO / WM) * (CI / 32) >= 8 ? 1 : 2)
void wgrad_b16k_kernel(const WgradBArgs g) {
  typedef WgradBGeom<KS, KHW, CO, CI, WM, XF> G;
  typedef __attribute__((address_space(3))) gv4i16* lds_v4;
  constexpr int P = KS / 2, MT = G::MT, NTAP = G::NTAP, NW = G::NW, NV = G::NV, PITCH = G::PITCH;
  constexpr int SLOTB = G::SLOTB;

  __shared__ __attribute__((aligned(1024))) char lds[G::LDSB];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / G::NWN, wn = wave - wm * G::NWN;

  const int nci = g.cin / CI, nco = g.cout / CO;
  int bid;
  {   // XCD-aware block order: the kernel-row workgroups / channel tiles of one batch split share an L2
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

  // work items of this split: a contiguous range of (image, patch row, patch column)
  const int tpi = g.tiles_x * g.tiles_y;
  const int total_items = g.B * tpi;
  const int per = (total_items + g.nsplit - 1) / g.nsplit;
  const int it0 = sp * per;
  const int it1 = it0 + per < total_items ? it0 + per : total_items;
  const int nitem = it1 > it0 ? it1 - it0 : 0;

  // ---- request constants of this lane: request v of the wave is 1-KB chunk k = wave + v * NW of the slot; the dz blocks
  // are chunks 0 .. C_out/16 - 1.  In most tile shapes C_out/16 = NW, i.e. request 0 of every wave is its dz block and
  // requests 1.. fetch `a` (G::DZ0: the kind is a compile-time fact); otherwise it is a wave-uniform run-time flag.
  int lc[NV], ry[NV], cx[NV];
  bool isdz[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int k = wave + v * NW;
    isdz[v] = G::DZ0 ? v == 0 : k < CO / 16;
    if (isdz[v]) {
      const int px = lane >> 1;
      ry[v] = px >> 3; cx[v] = px & 7;
      lc[v] = k * HW * 32 + (ry[v] * g.W + cx[v]) * 32 + (lane & 1) * 16;
    } else {
      const int ua = 64 * (k - CO / 16) + lane;
      const int blk = ua / G::UPB, rem = ua - blk * G::UPB;
      const int ra = rem / (2 * PITCH), t = rem - ra * 2 * PITCH, ca = t >> 1;
      const bool bad = blk >= CI / 16 || ca >= G::ACOLS;      // slot padding: never fetched
      ry[v] = bad ? 0x40000000 : ra; cx[v] = ca;
      lc[v] = blk * HW * 32 + (ra * g.W + ca) * 32 + (t & 1) * 16;
    }
  }
  const unsigned lds_a = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)lds;

  // next item to REQUEST: running 64-bit addresses of its corner in both tensors (the input's shifted by the kernel-row
  // group's first tap), advanced by one of three precomputed steps (next patch / next patch row / next image)
  int left = nitem;
  int ntx, nty;
  unsigned long long pd, pa;
  {
    const int b = it0 / tpi;
    nty = (it0 - b * tpi) / g.tiles_x; ntx = it0 - b * tpi - nty * g.tiles_x;
    pd = (unsigned long long)g.dz + (((unsigned long long)b * dz_blocks + (dz_c0 >> 4)) * HW + nty * 4 * g.W + ntx * 8) * 32;
    pa = (unsigned long long)g.a +
         (unsigned long long)((((long long)b * a_blocks + (a_c0 >> 4)) * HW + (nty * 4 + kh0 - P) * g.W + (ntx * 8 - P)) * 32);
  }
  const long long step_row = ((long long)4 * g.W - (g.tiles_x - 1) * 8) * 32;
  const long long last_off = ((long long)(g.tiles_y - 1) * 4 * g.W + (g.tiles_x - 1) * 8) * 32;
  const long long step_img_d = (long long)dz_blocks * HW * 32 - last_off, step_img_a = (long long)a_blocks * HW * 32 - last_off;
  unsigned long long q_pd = 0, q_pa = 0;      // the item being requested
  int q_y = 0, q_x = 0, q_nr = 0;
  auto item_next = [&]() __attribute__((always_inline)) {
    q_pd = pd; q_pa = pa; q_y = nty * 4; q_x = ntx * 8;
    q_nr = 0x7fffffff & -(int)(left > 0);
    --left;
    const bool wx = ntx + 1 == g.tiles_x, wy = wx && nty + 1 == g.tiles_y;
    pd += wx ? (wy ? step_img_d : step_row) : 256;
    pa += wx ? (wy ? step_img_a : step_row) : 256;
    nty = wx ? (wy ? 0 : nty + 1) : nty;
    ntx = wx ? 0 : ntx + 1;
  };
  auto dma = [&](auto vc, int slot) __attribute__((always_inline)) {
    constexpr int v = decltype(vc)::value;
    const bool dk = G::DZ0 ? v == 0 : isdz[v];
    const unsigned long long bs = dk ? q_pd : q_pa;
    const int Y = dk ? q_y : q_y + kh0 - P, X = dk ? q_x : q_x - P;
    const bool ok = ((unsigned)(Y + ry[v]) < (unsigned)g.H) & ((unsigned)(X + cx[v]) < (unsigned)g.W);
    const int vo = ok ? lc[v] : (int)0x80000000;
    const gi32x4 rs = {(int)bs, (int)(bs >> 32) & 0xffff, q_nr, 0x00020000};
    wgb_dma(rs, vo, lds_a + slot * SLOTB + 1024 * (wave + v * NW));
  };
  auto request = [&](int slot) __attribute__((always_inline)) {
    item_next();
    wgb_static_for<0, NV>([&](auto vc) __attribute__((always_inline)) { dma(vc, slot); });
  };
#define WGB_VM_WAIT(n_) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n_) & 15) | (((n_) >> 4) << 14))
// (XF: also lgkmcnt(0) -- this thread's transformed units are in LDS before the barrier)
#define WGB_STEP_END_X(n_)                                                                \
  {                                                                                      \
    asm volatile("" ::: "memory");                                                       \
    __builtin_amdgcn_s_waitcnt(0x0070 | ((n_) & 15) | (((n_) >> 4) << 14));              \
    __builtin_amdgcn_s_barrier();                                                        \
    asm volatile("" ::: "memory");                                                       \
  }
#define WGB_STEP_END(n_)                      \
  {                                          \
    asm volatile("" ::: "memory");           \
    WGB_VM_WAIT(n_);                         \
    __builtin_amdgcn_s_barrier();            \
    asm volatile("" ::: "memory");           \
  }

  // ---- fragment addressing (ds_read_b64_tr_b16): lane 16 g + 4 q + p supplies the address of pixel row q of its group's
  // 4 pixels, channels 4p..4p+3 (8 B) and receives those 4 pixels of channel (lane & 15)
  const int g4 = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
  const int a_lane = (wm * MT) * 1024 + (4 * g4 + tq) * 32 + tp * 8;                               // + m * 1024, + j * 512
  const int b_lane = G::UDZ * 16 + (wn * 2) * G::UPB * 16 + ((g4 >> 1) * PITCH + 4 * (g4 & 1) + tq) * 32 + tp * 8;

  f32x4 acc[MT][2][NTAP];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int t = 0; t < NTAP; ++t) acc[m][n][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;

  auto frag = [&](const char* p) __attribute__((always_inline)) {
    const gv4i16 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p));
    return __builtin_bit_cast(gb16x4, r0);
  };

  auto run = [&](auto bias_c) __attribute__((always_inline)) {
    constexpr bool BIAS = decltype(bias_c)::value;
    // XF: this thread's 16-B units of the input tile (8 channels of one pixel each) and their BatchNorm scale / shift
    int xoff[XF ? G::NXF : 1];
    f32x4 xsc[XF ? G::NXF : 1][2], xsh[XF ? G::NXF : 1][2];
    if constexpr (XF) {
#pragma unroll
      for (int i = 0; i < G::NXF; ++i) {
        const int u = threadIdx.x + i * G::NT;
        const int uc = u < G::UA ? u : G::UA - 1;
        const int blk = uc / G::UPB, half = uc & 1;
        xoff[i] = u < G::UA ? G::UDZ * 16 + u * 16 : -1;
        const float* sp_ = g.a_scale + cib * CI + blk * 16 + half * 8;
        const float* hp_ = g.a_shift + cib * CI + blk * 16 + half * 8;
        xsc[i][0] = *(const f32x4*)sp_ * 0.5f; xsc[i][1] = *(const f32x4*)(sp_ + 4) * 0.5f;
        xsh[i][0] = *(const f32x4*)hp_ * 0.5f; xsh[i][1] = *(const f32x4*)(hp_ + 4) * 0.5f;
      }
    }
    // in place: bf16(relu(fp32(z) * scale + shift)) -- what wgrad_tr16_kernel<.., IO16> forms while staging (bit for bit: halving
    // the two vectors is exact).  (Pixels outside the
    // image arrive as zeros and become relu(shift): their dz is zero, so they add nothing.)
    // two halves: the unit reads are issued at the top of a step (their latency passes under the step's MFMAs), the arithmetic
    // and the write-back follow the MFMAs
    gb16x8 xz[XF ? G::NXF : 1];
    constexpr bool XALL = G::UA % G::NT == 0;           // every thread has all its NXF units
    auto xf_load = [&](int slot) __attribute__((always_inline)) {
      if constexpr (XF) {
#pragma unroll
        for (int i = 0; i < G::NXF; ++i)
          if (XALL || xoff[i] >= 0) xz[i] = *(const gb16x8*)(lds + slot * SLOTB + xoff[i]);
      }
    };
    auto xf_store = [&](int slot) __attribute__((always_inline)) {
      if constexpr (XF) {
#pragma unroll
        for (int i = 0; i < G::NXF; ++i) {
          if (XALL || xoff[i] >= 0) {
            gb16x8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {      // (scale / shift are HALVED: relu(t) = t/2 + |t/2|, one instruction, NaN-propagating)
              o[e] = (__bf16)tsr_relu_x2(fmaf((float)xz[i][e], xsc[i][0][e], xsh[i][0][e]));
              o[4 + e] = (__bf16)tsr_relu_x2(fmaf((float)xz[i][4 + e], xsc[i][1][e], xsh[i][1][e]));
            }
            *(gb16x8*)(lds + slot * SLOTB + xoff[i]) = o;
          }
        }
      }
    };
    // prologue: the first LA items requested; item 0 (XF: items 0 and 1) landed
#pragma unroll
    for (int i = 0; i < G::LA; ++i) request(i);
    if constexpr (XF) {
      WGB_STEP_END((G::LA - 2) * NV);
      xf_load(0);
      xf_store(0);
      WGB_STEP_END_X((G::LA - 2) * NV);
    } else {
      WGB_STEP_END(0);
    }
    int slot = 0;
    for (int s = 0; s < nitem; ++s) {
      const int slot2 = slot == 0 ? G::RING - 1 : slot - 1;   // (slot + LA) % RING
      const char* sa = lds + slot * SLOTB + a_lane;
      const char* sb = lds + slot * SLOTB + b_lane;
      constexpr int PF = TSR_WGB_PF;             // B fragments are read PF taps ahead of their MFMAs
      gb16x8 af[MT], bf[PF + 1][2];
      auto load_b = [&](gb16x8* dst, int tap) __attribute__((always_inline)) {
        const int khl = tap / KS, kw = tap - khl * KS;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const char* p = sb + n * G::UPB * 16 + khl * PITCH * 32 + kw * 32;
          dst[n] = __builtin_shufflevector(frag(p), frag(p + 2 * PITCH * 32), 0, 1, 2, 3, 4, 5, 6, 7);
        }
      };
#pragma unroll
      for (int m = 0; m < MT; ++m)
        af[m] = __builtin_shufflevector(frag(sa + m * 1024), frag(sa + m * 1024 + 512), 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int t = 0; t < PF; ++t) load_b(bf[t], t);
      item_next();
      xf_load(slot == G::RING - 1 ? 0 : slot + 1);      // XF: item s + 1 (landed before this step began)
      __builtin_amdgcn_sched_barrier(0);
      wgb_static_for<0, NTAP>([&](auto tc) __attribute__((always_inline)) {
        constexpr int u = decltype(tc)::value;
        if constexpr (u + PF < NTAP) load_b(bf[(u + PF) % (PF + 1)], u + PF);
        // item s + LA, one request per tap (the wave's instruction issue is the scarce thing: a request is ~12 scalar /
        // vector instructions, which fit between a tap's MFMAs; all of them in one block at the step's top would stop both
        // waves of the SIMD at the same time); a step with fewer taps than requests issues the rest with its last tap
        if constexpr (u < NV && u < NTAP - 1) dma(std::integral_constant<int, u>(), slot2);
        if constexpr (u == NTAP - 1)
          wgb_static_for<(NTAP - 1 < NV ? NTAP - 1 : NV), NV>([&](auto vc) __attribute__((always_inline)) { dma(vc, slot2); });
        if constexpr (BIAS && u == (NV < NTAP - 1 ? NV : NTAP - 1)) {
          typedef unsigned gu32x4 __attribute__((ext_vector_type(4)));
          gu32x4 sel = {0u, 0u, 0u, 0u};
#pragma unroll
          for (int m = 0; m < MT; ++m) sel = wn == m ? __builtin_bit_cast(gu32x4, af[m]) : sel;
#pragma unroll
          for (int e = 0; e < 4; ++e) bsum += __uint_as_float(sel[e] << 16) + __uint_as_float(sel[e] & 0xffff0000u);
        }
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int m = 0; m < MT; ++m)
            acc[m][n][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m], bf[u % (PF + 1)][n], acc[m][n][u], 0, 0, 0);
        if constexpr (u + PF < NTAP) {
          constexpr int PER = (2 * MT) / 4 > 0 ? (2 * MT) / 4 : 1;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);   // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // DS read
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (XF) {       // item s + 1 (landed before this step began) becomes the activation, for the next step's reads
        xf_store(slot == G::RING - 1 ? 0 : slot + 1);
        WGB_STEP_END_X((G::LA - 2) * NV);
      } else {
        WGB_STEP_END((G::LA - 1) * NV);
      }
      slot = slot == G::RING - 1 ? 0 : slot + 1;
    }
    WGB_VM_WAIT(0);        // (the trailing zero-range requests)
  };
  if (do_bias) run(std::true_type());
  else run(std::false_type());

  // ---- this split's partial dW: slab[sp][co][ci][kh][kw]
  {
    constexpr int T = KS * KS;
    float* sl = g.slab + (size_t)sp * g.cout * g.cin * T;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const int ci = cib * CI + wn * 32 + n * 16 + (lane & 15);
#pragma unroll
        for (int tap = 0; tap < NTAP; ++tap) {
          if (kh0 * KS + tap < T) {          // a short last row group holds rows beyond the kernel: not written
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int co = cob * CO + wm * WM + m * 16 + 4 * g4 + r;
              sl[((size_t)co * g.cin + ci) * T + kh0 * KS + tap] = acc[m][n][tap][r];
            }
          }
        }
      }
  }
  if (do_bias) {
    bsum += __shfl_xor(bsum, 16);
    bsum += __shfl_xor(bsum, 32);
    if (lane < 16 && wn < MT) g.bslab[(size_t)sp * g.cout + cob * CO + wm * WM + wn * 16 + lane] = bsum;
  }
}

template <int KS, int KHW, int CO, int CI, int WM, bool XF = false>
static int wgb_launch(const WgradBArgs& g, hipStream_t st) {
  typedef WgradBGeom<KS, KHW, CO, CI, WM, XF> G;
  const int grid = g.nsplit * G::NKG * (g.cout / CO) * (g.cin / CI);
  hipLaunchKernelGGL((wgrad_b16k_kernel<KS, KHW, CO, CI, WM, XF>), dim3(grid), dim3(G::NT), 0, st, g);
  return tsr_check_launch();
}

// 1 if tsr_conv2d_wgrad_bf16s (planes = -1, no input transform) runs this shape here: 3x3 / 5x5, the tiles of
// wgrad_mfma_tr16.hip's WgradTCfg (so tsr_conv2d_wgrad_splits and the slab layout are the same for both kernels)
bool tsr_wgrad_b16k_ok(int cout, int cin, int ks, int H, int W, int a_ctot, int dz_ctot) {
  if ((long long)(a_ctot > dz_ctot ? a_ctot : dz_ctot) * H * W * 2 >= 0x7fffffffLL) return false;   // 32-bit offsets inside an image
  if (ks == 1) return cout == 64 && (cin % 256) == 0;       // the MSRB `confusion`: one workgroup = 64 x 256, input transform in LDS
  if (ks != 3 && ks != 5) return false;
  return (cout % 64) == 0 && (cin % 64) == 0;
}
// 1x1 (cout = 64, cin a multiple of 256): workgroups per batch split
int tsr_wgrad_b16k_1x1_wgs(int cout, int cin) { return cout == 64 && (cin % 256) == 0 ? cin / 256 : 0; }

// `a_scale` / `a_shift` (the input is relu(a * scale + shift)): 1x1 only -- the 3x3 / 5x5 launches take a plain input
int tsr_wgrad_b16k(const void* a, int a_ctot, int a_coff, int cin, const float* a_scale, const float* a_shift, const void* dz,
                   int dz_ctot, int dz_coff, int cout, int ks, float* slab, float* bias_slab, int nsplit, int B, int H, int W,
                   hipStream_t st) {
  WgradBArgs g;
  g.a = (const char*)a; g.a_ctot = a_ctot; g.a_coff = a_coff; g.cin = cin; g.a_scale = a_scale; g.a_shift = a_shift;
  g.dz = (const char*)dz; g.dz_ctot = dz_ctot; g.dz_coff = dz_coff; g.cout = cout;
  g.slab = slab; g.bslab = bias_slab; g.B = B; g.H = H; g.W = W; g.nsplit = nsplit;
  g.tiles_x = (W + 7) / 8; g.tiles_y = (H + 3) / 4;
  // (a 16-wave form of the 1x1 tile -- WM = 32 -- and ring depths 5 / 6 measured the same as this one)
  if (ks == 1) return a_scale ? wgb_launch<1, 1, 64, 256, 64, true>(g, st) : wgb_launch<1, 1, 64, 256, 64, false>(g, st);
  if (a_scale) return TSR_ERR_ARG;
  const bool big5 = (cout % 128) == 0 && (cin % 128) == 0, big3 = (cout % 128) == 0 && (cin % 64) == 0;
  if (ks == 5) return big5 ? wgb_launch<5, 1, 128, 128, 64>(g, st) : wgb_launch<5, 2, 64, 64, 32>(g, st);
  return big3 ? wgb_launch<3, 3, 128, 64, 32>(g, st) : wgb_launch<3, 3, 64, 64, 32>(g, st);
}

// The stored input of a conv that follows a train-mode BatchNorm + ReLU is the pre-BatchNorm z (`virtual` activation:
// relu(z * scale + shift), the transform fused into its consumers' staging).  The LDS-DMA kernels cannot transform while
// staging, so their launches read this tensor's MATERIALISED form: out = bf16(relu(fp32(z) * scale + shift)), dense bf16
// CB16 with C channels -- exactly the value wgrad_tr16_kernel<.., IO16> forms while staging.  One workgroup per (image,
// 16-channel block) plane: a thread keeps one 8-channel half (its scale / shift in registers) and walks the plane's pixels.
__global__ __launch_bounds__(256) void bn_relu_b16_kernel(const char* z, int z_blocks, int z_blk0, int cblocks, const float* scale,
                                                          const float* shift, char* out, int HW) {
  const int blk = blockIdx.x % cblocks, b = blockIdx.x / cblocks;
  const int half = threadIdx.x & 1;
  const f32x4 s0 = *(const f32x4*)(scale + blk * 16 + half * 8), s1 = *(const f32x4*)(scale + blk * 16 + half * 8 + 4);
  const f32x4 t0 = *(const f32x4*)(shift + blk * 16 + half * 8), t1 = *(const f32x4*)(shift + blk * 16 + half * 8 + 4);
  const char* zp = z + ((size_t)b * z_blocks + z_blk0 + blk) * HW * 32;
  char* op = out + ((size_t)b * cblocks + blk) * HW * 32;
  for (int c = threadIdx.x; c < 2 * HW; c += 256) {
    const gb16x8 v = *(const gb16x8*)(zp + (size_t)c * 16);
    gb16x8 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[i] = (__bf16)tsr_relu(fmaf((float)v[i], s0[i], t0[i]));
      o[4 + i] = (__bf16)tsr_relu(fmaf((float)v[4 + i], s1[i], t1[i]));
    }
    *(gb16x8*)(op + (size_t)c * 16) = o;
  }
}

extern "C" int tsr_bn_relu_b16(const void* z, int z_ctot, int z_coff, int C, const float* scale, const float* shift, void* out,
                               int B, int HW, void* stream) {
  if (!z || !scale || !shift || !out || B <= 0 || HW <= 0 || C <= 0 || (C & 15) || (z_ctot & 15) || (z_coff & 15) ||
      z_coff + C > z_ctot || (long long)B * (C >> 4) > 0x7fffffffLL)
    return TSR_ERR_ARG;
  hipLaunchKernelGGL(bn_relu_b16_kernel, dim3(B * (C >> 4)), dim3(256), 0, (hipStream_t)stream, (const char*)z, z_ctot >> 4,
                     z_coff >> 4, C >> 4, scale, shift, (char*)out, HW);
  return tsr_check_launch();
}

// 1 if tsr_conv2d_wgrad_bf16s (planes = -1) runs a launch of this shape WITHOUT an input transform on this file's kernel: a
// caller holding a virtual input may then materialise it once (tsr_bn_relu_b16) and pass it plain
extern "C" int tsr_conv2d_wgrad_b16k(int cout, int cin, int ks) { return (ks == 3 || ks == 5) && (cout % 64) == 0 && (cin % 64) == 0; }

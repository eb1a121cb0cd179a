// The two dgrad launches of an MSRB's 1x1 `confusion` in the bf16-storage train step (tsr_conv2d_ex, nsplit = -3, ks = 1,
// epi_mode 2): g[128 channels of cat2's gradient] = W^T . dpre[64], zeroed where the stored activation's BatchNorm + ReLU
// was off, plus the BatchNorm-backward sums.  The layer is HBM-bound (2.1 GB per launch at B = 2048 against 0.05 ms of MFMA
// work); the tiled 32x32x16 kernel it ran on stages through LDS behind barriers (3.4 TB/s).  Here NOTHING touches LDS:
//
//   * channels as rows on v_mfma_f32_16x16x32_bf16 (as conv_b16k.hip): the B operand of a 16-pixel group is, per lane
//     (pixel n, k group g), the 8 consecutive input channels 32 kk + 8 g .. + 7 of its pixel = ONE 16-B global load from the
//     CB16 tensor (block 2 kk + (g >> 1), half g & 1); the accumulator holds 4 consecutive output channels of a pixel per
//     lane = the 8 bytes of the mask read and of the output store: a wave's 64 lanes cover 512 contiguous bytes;
//   * wave w of a workgroup owns output tiles 2w, 2w+1 (32 of the 128 channels): its weights (16 registers), its
//     per-channel vectors (32) and its sums (16) stay in registers for the whole launch, U pixel groups are in flight;
//     the four waves re-read the 64-channel input from L1 / L2 (a fifth of the launch's bytes);
//   * a workgroup walks a contiguous range of pixel groups and writes ONE slab entry (sum v, sum v * xhat per channel)
//     at its end: entries = workgroups (tsr_conv2d_slab_entries_ex, nsplit = -3, ks = 1).
#include "tsr_common.h"
#include "conv_args.h"
#include "tactilesr_hip.h"

typedef __bf16 xb16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 xb16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float x1_row_sum(float v) {      // sum over the 16 lanes of a DPP row (conv_b16k.hip)
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));
  return v;
}

// workgroups (= slab entries) of a launch: pixel groups of 16, at most 8 resident workgroups on each of the 256 CUs
int tsr_dgrad1x1_b16k_grid(int B, int H, int W) {
  const long long groups = (long long)B * ((H * W + 15) / 16);
  return (int)(groups < 2048 ? groups : 2048);
}

template <int U>
__global__ __launch_bounds__(256) void dgrad1x1_b16k_kernel(const ConvArgs a, int gpi /* pixel groups per image */, int per) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int m = lane & 15, g = lane >> 4;
  const int HW = a.H * a.W;
  const size_t plane = (size_t)HW * 32;
  const int in_blocks = a.in_ctot >> 4, mask_blocks = a.mask_ctot >> 4, out_blocks = a.out_ctot >> 4;
  // weights: tsr_pack_conv_weight_dgrad_b16k's slab layout [k step][k group][128][8]
  xb16x8 A[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
      A[t][kk] = *(const xb16x8*)((const char*)a.wp + kk * 8192 + ((g * 128 + (2 * wv + t) * 16 + m) * 16));
  const f32x4 one4 = {1.f, 1.f, 1.f, 1.f}, zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 msc[2], msh[2], ba[2], bb[2], s1[2], s2[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int nq = (2 * wv + t) * 16 + 4 * g;
    msc[t] = a.mask_scale ? *(const f32x4*)(a.mask_scale + nq) : one4;
    msh[t] = a.mask_scale ? *(const f32x4*)(a.mask_shift + nq) : zero4;
    ba[t] = a.bn_a ? *(const f32x4*)(a.bn_a + nq) : zero4;
    bb[t] = a.bn_a ? *(const f32x4*)(a.bn_b + nq) : zero4;
    s1[t] = zero4; s2[t] = zero4;
  }
  const int total = a.B * gpi;
  const int i0 = blockIdx.x * per, i1 = i0 + per < total ? i0 + per : total;
  const char* inp = (const char*)a.in + (size_t)(a.in_coff >> 4) * plane + (g >> 1) * plane + (g & 1) * 16;
  const char* mkp = (const char*)a.mask + (size_t)((a.mask_coff >> 4) + 2 * wv) * plane + 8 * g;
  char* outp = (char*)a.out + (size_t)((a.out_coff >> 4) + 2 * wv) * plane + 8 * g;
  for (int i = i0; i < i1; i += U) {
    xb16x8 Bf[U][2];
    xb16x4 mk[U][2];
    bool ok[U];
    unsigned po[U];
    size_t ib[U], mb[U], ob[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ii = i + u < i1 ? i + u : i1 - 1;
      const int b = ii / gpi, q = ii - b * gpi;
      const int p = q * 16 + m;
      ok[u] = i + u < i1 && p < HW;
      po[u] = (unsigned)(p < HW ? p : HW - 1) * 32u;
      ib[u] = (size_t)b * in_blocks * plane; mb[u] = (size_t)b * mask_blocks * plane; ob[u] = (size_t)b * out_blocks * plane;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) Bf[u][kk] = *(const xb16x8*)(inp + ib[u] + (size_t)(2 * kk) * plane + po[u]);
#pragma unroll
      for (int t = 0; t < 2; ++t) mk[u][t] = *(const xb16x4*)(mkp + mb[u] + (size_t)t * plane + po[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x4 acc = zero4;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[t][kk], Bf[u][kk], acc, 0, 0, 0);
        const f32x4 mf = __builtin_convertvector(mk[u][t], f32x4);
        f32x4 x;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float v = acc[c];
          if (!(fmaf(mf[c], msc[t][c], msh[t][c]) > 0.f)) v = 0.f;
          if (!ok[u]) v = 0.f;
          x[c] = v;
          s1[t][c] += v;
          s2[t][c] = fmaf(v, fmaf(mf[c], ba[t][c], bb[t][c]), s2[t][c]);
        }
        if (ok[u]) *(xb16x4*)(outp + ob[u] + (size_t)t * plane + po[u]) = __builtin_convertvector(x, xb16x4);
      }
    }
  }
  if (a.bn_a) {
    float* sl = a.slab + ((size_t)blockIdx.x * 128 + (2 * wv) * 16 + 4 * g) * 2;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float r1 = x1_row_sum(s1[t][c]), r2 = x1_row_sum(s2[t][c]);
        if (m == 0) { sl[(t * 16 + c) * 2] = r1; sl[(t * 16 + c) * 2 + 1] = r2; }
      }
  }
}

// called by tsr_conv_b16k_ex (conv_b16k.hip) for ks = 1: C_in (the forward conv's C_out) = 64, 128 output channels, epi_mode 2
// without a partial gradient / output scale
int tsr_dgrad1x1_b16k(const ConvArgs& a, hipStream_t st) {
  if (a.cin != 64 || a.epi_mode != 2 || !a.mask || a.res || a.scale || a.shift || a.in_scale || (a.bn_a && !a.slab))
    return TSR_ERR_ARG;
  const int gpi = (a.H * a.W + 15) / 16;
  const int grid = tsr_dgrad1x1_b16k_grid(a.B, a.H, a.W);
  const int per = (int)(((long long)a.B * gpi + grid - 1) / grid);
  hipLaunchKernelGGL(dgrad1x1_b16k_kernel<4>, dim3(grid), dim3(256), 0, st, a, gpi, per);
  return tsr_check_launch();
}

// ---------------------------------------------------------------------------------------------------------------------------
// The FORWARD of the 1x1 `confusion` in the bf16-storage train step (tsr_conv2d_ex, nsplit = -3, ks = 1, epi_mode 0):
//   out[64] = act(W[64 x C_in] . relu(z * in_scale + in_shift) + shift + res),      C_in = 256 (a multiple of 32, <= 256)
// on a VIRTUAL input (the stored pre-BatchNorm cat2).  2.5 GB per launch at B = 2048; the register-streaming kernel
// (conv1x1_b16_ex_kernel) moves 3.6 TB/s.  Here the raw z tile of an item (32 consecutive pixels x C_in channels: one
// contiguous KB per 16-channel block) arrives by LDS-DMA, is transformed IN LDS one step ahead of its use (as the 1x1
// weight gradient does, wgrad_b16k.hip), and feeds v_mfma_f32_16x16x32_bf16 with channels as rows: the B operand of lane
// (pixel n, k group g) is one ds_read_b128 (8 consecutive channels of its pixel), wave (tile mt, pixel half ph) keeps the
// 16 x C_in weight tile in registers, and its accumulator (4 channels of a pixel per lane) is the 8 bytes of the residual
// read and of the store.  4-slot ring, counted vmcnt + raw barrier per item; persistent workgroups over contiguous item ranges.
typedef int xi32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void x1_dma(xi32x4 rs, int vo, unsigned m0v) {      // LDS-DMA request (conv_b16k.hip)
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(m0v), "v"(vo), "s"(rs));
}

template <int NBLK>        // 16-channel blocks of the input (C_in / 16)
__global__ __launch_bounds__(512) void fwd1x1_b16k_kernel(const ConvArgs a, int ipi /* items per image */, int per) {
  constexpr int NK = NBLK / 2;                       // K steps of 32 channels
  constexpr int SLOTB = NBLK * 1024, RING = 4, LA = 3;
  constexpr int NV = (NBLK + 7) / 8;                 // requests per wave and item (8 waves)
  constexpr int NXF = (NBLK * 64 + 511) / 512;       // 16-B units a thread transforms per item
  __shared__ __attribute__((aligned(1024))) char lds[RING * SLOTB];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int mt = wave & 3, ph = wave >> 2;
  const int HW = a.H * a.W;
  const unsigned plane = (unsigned)HW * 32u;
  const int in_blocks = a.in_ctot >> 4;
  // weights (tsr_pack_conv_weight_b16k, ks = 1): [k step][k group][64][8]
  xb16x8 A[NK];
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) A[kk] = *(const xb16x8*)((const char*)a.wp + ((size_t)(kk * 4 + g) * 64 + mt * 16 + n) * 16);
  const f32x4 shv = a.shift ? *(const f32x4*)(a.shift + mt * 16 + 4 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
  // (the residual may be virtual too: the first MSRB's input is the stored pre-BatchNorm tensor of the contact layer)
  const f32x4 rsc = a.res_scale ? *(const f32x4*)(a.res_scale + mt * 16 + 4 * g) : (f32x4){1.f, 1.f, 1.f, 1.f};
  const f32x4 rsh = a.res_scale ? *(const f32x4*)(a.res_shift + mt * 16 + 4 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool rvirt = a.res_scale != nullptr;
  // this thread's transform units and their (halved) BatchNorm vectors: relu(t) = t/2 + |t/2|
  int xoff[NXF];
  f32x4 xsc[NXF][2], xsh[NXF][2];
#pragma unroll
  for (int i = 0; i < NXF; ++i) {
    const int u = threadIdx.x + i * 512;             // unit = (block, pixel, half)
    const int uc = u < NBLK * 64 ? u : NBLK * 64 - 1;
    const int blk = uc >> 6, half = uc & 1;
    xoff[i] = u < NBLK * 64 ? u * 16 : -1;
    const float* sp_ = a.in_scale + blk * 16 + half * 8;
    const float* hp_ = a.in_shift + blk * 16 + half * 8;
    xsc[i][0] = *(const f32x4*)sp_ * 0.5f; xsc[i][1] = *(const f32x4*)(sp_ + 4) * 0.5f;
    xsh[i][0] = *(const f32x4*)hp_ * 0.5f; xsh[i][1] = *(const f32x4*)(hp_ + 4) * 0.5f;
  }
  const int total = a.B * ipi;
  const int i0 = blockIdx.x * per, i1 = i0 + per < total ? i0 + per : total;
  const int nitem = i1 > i0 ? i1 - i0 : 0;
  const unsigned lds_a = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)lds;
  const unsigned long long in_base = (unsigned long long)a.in + (unsigned long long)(a.in_coff >> 4) * plane;

  int rq = i0;                                       // next item to request
  auto request = [&](int slot) __attribute__((always_inline)) {
    const int real = -(int)(rq < i1);
    const int ii = rq < total ? rq : total - 1;
    ++rq;
    const int b = ii / ipi, q = ii - b * ipi;
    const unsigned long long bs = in_base + (unsigned long long)b * in_blocks * plane;
    const int p = q * 32 + (lane >> 1);
    const xi32x4 rs = {(int)bs, (int)(bs >> 32) & 0xffff, 0x7fffffff & real, 0x00020000};
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int blk = wave + v * 8;
      const int vo = (p < HW && blk < NBLK) ? (int)(blk * plane + p * 32 + (lane & 1) * 16) : (int)0x80000000;
      if (NBLK % 8 == 0 || blk < NBLK) x1_dma(rs, vo, lds_a + slot * SLOTB + 1024 * blk);
    }
  };
  xb16x8 xz[NXF];
  auto xf_load = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NXF; ++i)
      if (xoff[i] >= 0) xz[i] = *(const xb16x8*)(lds + slot * SLOTB + xoff[i]);
  };
  auto xf_store = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NXF; ++i)
      if (xoff[i] >= 0) {
        xb16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = (__bf16)tsr_relu_x2(fmaf((float)xz[i][e], xsc[i][0][e], xsh[i][0][e]));
          o[4 + e] = (__bf16)tsr_relu_x2(fmaf((float)xz[i][4 + e], xsc[i][1][e], xsh[i][1][e]));
        }
        *(xb16x8*)(lds + slot * SLOTB + xoff[i]) = o;
      }
  };
#define X1_STEP_END(n_)                                                                   \
  {                                                                                      \
    asm volatile("" ::: "memory");                                                       \
    __builtin_amdgcn_s_waitcnt(0x0070 | ((n_) & 15) | (((n_) >> 4) << 14));              \
    __builtin_amdgcn_s_barrier();                                                        \
    asm volatile("" ::: "memory");                                                       \
  }
  // prologue: items 0 .. 2 requested, 0 and 1 landed, 0 transformed
#pragma unroll
  for (int i = 0; i < LA; ++i) request(i);
  X1_STEP_END(NV);
  xf_load(0);
  xf_store(0);
  X1_STEP_END(NV);
  const char* bl = lds + ((g >> 1) * 1024) + (ph * 16 + n) * 32 + (g & 1) * 16;      // + slot, + 2 kk blocks
  const char* resp = a.res ? (const char*)a.res + (size_t)((a.res_coff >> 4) + mt) * plane + 8 * g : nullptr;
  char* outp = (char*)a.out + (size_t)((a.out_coff >> 4) + mt) * plane + 8 * g;
  // The result of step s is STORED at the top of step s + 1: stores share the vmcnt counter with the requests (and may retire
  // out of order with respect to loads), so a store issued right in front of the step's counted wait would make that wait
  // include its whole round trip.
  xb16x4 pend = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
  char* pend_p = nullptr;
  int slot = 0;
  for (int s = 0; s < nitem; ++s) {
    if (pend_p) *(xb16x4*)pend_p = pend;
    const int ii = i0 + s;
    const int b = ii / ipi, q = ii - b * ipi;
    const int p = q * 32 + ph * 16 + n;
    const bool ok = p < HW;
    const unsigned po = (unsigned)(ok ? p : 0) * 32u;
    xb16x4 rv = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
    if (resp) rv = *(const xb16x4*)(resp + (size_t)b * (a.res_ctot >> 4) * plane + po);
    const int nslot = slot == RING - 1 ? 0 : slot + 1;
    xb16x8 Bf[NK];
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) Bf[kk] = *(const xb16x8*)(bl + slot * SLOTB + kk * 2048);
    xf_load(nslot);                                   // item s + 1 (landed before this step began)
    request(slot == 0 ? RING - 1 : slot - 1);         // item s + 3 into the slot item s - 1 left
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[kk], Bf[kk], acc, 0, 0, 0);
    xf_store(nslot);
    f32x4 rf = __builtin_convertvector(rv, f32x4);
    if (rvirt) {
#pragma unroll
      for (int c = 0; c < 4; ++c) rf[c] = tsr_relu(fmaf(rf[c], rsc[c], rsh[c]));
    }
    f32x4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float t = acc[c] + shv[c] + rf[c];
      o[c] = a.relu ? tsr_relu(t) : t;
    }
    pend = __builtin_convertvector(o, xb16x4);
    pend_p = ok ? outp + (size_t)b * (a.out_ctot >> 4) * plane + po : nullptr;
    X1_STEP_END(NV);
    slot = nslot;
  }
  if (pend_p) *(xb16x4*)pend_p = pend;
  __builtin_amdgcn_s_waitcnt(0x0F70);      // the trailing zero-range requests
#undef X1_STEP_END
}

// called by tsr_conv_b16k_ex for ks = 1, epi_mode 0: 64 output channels, a virtual input of 64 .. 256 channels
int tsr_fwd1x1_b16k(const ConvArgs& a, int cout, hipStream_t st) {
  if (cout != 64 || a.epi_mode != 0 || !a.in_scale || !a.in_shift || a.scale || (a.res_scale && !a.res) || (a.cin != 256 && a.cin != 128))
    return TSR_ERR_ARG;
  const int HW = a.H * a.W, ipi = (HW + 31) / 32;
  if ((long long)a.in_ctot * HW * 2 >= 0x7fffffffLL) return TSR_ERR_ARG;      // 32-bit offsets inside an image
  const long long total = (long long)a.B * ipi;
  const int grid = (int)(total < 512 ? total : 512);
  const int per = (int)((total + grid - 1) / grid);
  if (a.cin == 256) hipLaunchKernelGGL(fwd1x1_b16k_kernel<16>, dim3(grid), dim3(512), 0, st, a, ipi, per);
  else hipLaunchKernelGGL(fwd1x1_b16k_kernel<8>, dim3(grid), dim3(512), 0, st, a, ipi, per);
  return tsr_check_launch();
}

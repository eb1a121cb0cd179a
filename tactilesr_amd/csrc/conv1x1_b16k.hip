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

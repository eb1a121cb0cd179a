// fp32 implicit-GEMM convolution on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces every nn.Conv2d(k in {1,3,5}, stride 1, "same" padding, C_in % 16 == 0,
// C_out in {64,128}) that TactileSR.forward dispatches -- MSRB conv_3_1/conv_5_1/
// conv_3_2/conv_5_2/confusion (model/tactileSR_model.py:167-191,196-206), ResBlock
// conv1/conv2 (:219-225), the stem's second conv and inputContact_layer (:41,47),
// output_layer[0] (:53) -- with the following BatchNorm (eval: folded to a per-channel
// scale/shift), bias, residual add and ReLU fused into the epilogue, and torch.cat
// elided by writing into a channel slice of the consumer's buffer.
//
// GEMM view: M = pixels, N = C_out, K = (C_in block, tap, channel-in-block).
// One workgroup (4 waves) = an 8x8 pixel patch of IMG=2 consecutive images (M = 128)
// x all C_out.  Per C_in block of 16 the (8+k-1)^2 halo slab of both images sits in
// LDS and is reused by all k*k taps; per (block, tap) a [16][C_out] weight slab is
// streamed through a 2-deep LDS ring (prefetched into registers one step ahead).
// Waves tile the output 2(M: image) x 2(N: C_out half); each wave owns 2 x (C_out/64)
// 32x32 accumulator tiles.  f32 MFMA is an exact k-ordered fmaf chain, so results are
// fp32-faithful (1e-5 relative parity with the reference's CPU path).
#include "tsr_common.h"
#include "tactilesr_hip.h"

#include "conv_args.h"
#include "conv_epilogue.h"

// EXT = training-path extensions compiled in (input/residual transforms, epi_mode 1/2);
// PF  = prefetch the next channel block's halo slab into registers during the current block.
template <int KS, int COUT, bool EXT, bool PF>
__global__ __launch_bounds__(256, 3) void conv_mfma_f32_kernel(const ConvArgs a) {
  constexpr int IMG = 2;
  constexpr int P = KS / 2;
  constexpr int HH = 8 + KS - 1;       // halo edge
  constexpr int SP = 20;               // floats per halo pixel (16 + 4 pad: 80-B stride)
  constexpr int T = KS * KS;
  constexpr int NB = COUT / 64;        // 32-wide N blocks per wave
  constexpr int HALO_F = IMG * HH * HH * SP;
  constexpr int WSLAB = 16 * COUT;     // floats per (block, tap) weight slab
  constexpr int WV = WSLAB / 4 / 256;  // float4 per thread per slab
  constexpr int NITEM = IMG * HH * HH * 4;
  constexpr int NIT = (NITEM + 255) / 256;

  __shared__ __attribute__((aligned(16))) float lds[HALO_F + 2 * WSLAB];
  float* halo = lds;
  float* wbuf = lds + HALO_F;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int h = lane >> 5, li = lane & 31;

  // XCD-aware bijective remap: each XCD (round-robin on blockIdx) gets a contiguous
  // range of tiles so neighbouring patches of one image share halo lines in one L2.
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

  // per-thread halo staging descriptors (block-invariant)
  int st_src[NIT];   // float offset inside one (image, block) plane set, or -1
  int st_dst[NIT];
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
      st_dst[k] = px * SP + qd * 4;
      if (b < a.B && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
        st_src[k] = ((img * in_blocks) * HW + gy * a.W + gx) * 16 + qd * 4;
    }
  }
  const float* in_base = a.in + ((size_t)b0 * in_blocks + (a.in_coff >> 4)) * HW * 16;

  const int laneA = wm * (HH * HH * SP) + ((li >> 3) * HH + (li & 7)) * SP + h * 4;
  const int laneB = (h * COUT + wn * (COUT / 2) + li) * 4;

  f32x16 acc[2][NB];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;

  const int nchunk = a.cin >> 4;
  const int S = nchunk * T;
  f32x4 wreg[WV];
  {
    const f32x4* src = (const f32x4*)a.wp;
#pragma unroll
    for (int v = 0; v < WV; ++v) wreg[v] = src[tid + v * 256];
  }

  // halo slab of channel block 0 -> registers; block c+1 is fetched while block c's taps compute
  f32x4 hv[NIT];
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    hv[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (PF && st_src[k] >= 0) hv[k] = *(const f32x4*)(in_base + st_src[k]);
  }

  int s = 0;
  for (int c = 0; c < nchunk; ++c) {
    // ---- stage the halo slab of channel block c (previous block's readers must be done)
    if (!PF) {
      const float* inc = in_base + (size_t)c * HW * 16;
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        hv[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (st_src[k] >= 0) hv[k] = *(const f32x4*)(inc + st_src[k]);
      }
    }
    if (EXT && a.in_scale) {   // wave-uniform branch: train-mode BN+ReLU of the producer, fused into the load
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        if (st_src[k] >= 0) {
          const int cq = c * 16 + ((tid + k * 256) & 3) * 4;
          const f32x4 sc = *(const f32x4*)(a.in_scale + cq);
          const f32x4 sh = *(const f32x4*)(a.in_shift + cq);
#pragma unroll
          for (int j = 0; j < 4; ++j) hv[k][j] = tsr_relu(fmaf(hv[k][j], sc[j], sh[j]));
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NIT; ++k)
      if (st_dst[k] >= 0) *(f32x4*)(halo + st_dst[k]) = hv[k];
    if (PF && c + 1 < nchunk) {   // next block's slab: in flight during this block's k*k tap steps
      const float* inc = in_base + (size_t)(c + 1) * HW * 16;
#pragma unroll
      for (int k = 0; k < NIT; ++k)
        if (st_src[k] >= 0) hv[k] = *(const f32x4*)(inc + st_src[k]);
    }

#pragma unroll
    for (int kh = 0; kh < KS; ++kh) {
#pragma unroll
      for (int kw = 0; kw < KS; ++kw) {
        float* wb = wbuf + (s & 1) * WSLAB;
#pragma unroll
        for (int v = 0; v < WV; ++v) ((f32x4*)wb)[tid + v * 256] = wreg[v];
        __syncthreads();
        if (s + 1 < S) {
          const f32x4* src = (const f32x4*)(a.wp + (size_t)(s + 1) * WSLAB);
#pragma unroll
          for (int v = 0; v < WV; ++v) wreg[v] = src[tid + v * 256];
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          f32x4 av[2], bv[NB];
#pragma unroll
          for (int mb = 0; mb < 2; ++mb)
            av[mb] = *(const f32x4*)(halo + laneA + ((4 * mb + kh) * HH + kw) * SP + g * 8);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            bv[nb] = *(const f32x4*)(wb + laneB + nb * 128 + g * (2 * COUT * 4));
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
              for (int nb = 0; nb < NB; ++nb)
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mb][j], bv[nb][j], acc[mb][nb], 0, 0, 0);
        }
        ++s;
      }
    }
  }

  conv_epilogue<COUT, EXT>(a, acc, bid, b0, y0, x0, wm, wn, h, li, HW);
}

// OIHW -> [C_in/16][tap][4 (channel quad)][C_out][4]: the order the kernel streams.
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, float* __restrict__ wp,
                                        int cout, int cin, int ks) {
  const int T = ks * ks;
  const size_t total = (size_t)cout * cin * T;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int jj = i & 3;
    size_t r = i >> 2;
    const int n = r % cout; r /= cout;
    const int kq = r & 3; r >>= 2;
    const int tap = r % T;
    const int chunk = r / T;
    const int ci = chunk * 16 + kq * 4 + jj;
    wp[i] = w[((size_t)n * cin + ci) * T + tap];
  }
}

template <int KS, int COUT, bool EXT>
static int launch_conv(const ConvArgs& a, hipStream_t st) {
  const int groups = (a.B + 1) / 2;
  const int grid = groups * a.tiles_x * a.tiles_y;
  // (the PF = true form -- halo prefetch into registers -- measured no faster in round 1 and is not instantiated)
  hipLaunchKernelGGL((conv_mfma_f32_kernel<KS, COUT, EXT, false>), dim3(grid), dim3(256), 0, st, a);
  return tsr_check_launch();
}

// Data-gradient weights: the dgrad of conv(W: OIHW) is itself a conv of dz with
// W'[n=ci][k=co][kh][kw] = W[co][ci0+n][KS-1-kh][KS-1-kw]; pack a 64/128-wide slice of ci.
__global__ void pack_conv_weight_dgrad_kernel(const float* __restrict__ w, float* __restrict__ wp, int cout_f,
                                              int cin_f, int ks, int ci0, int nprime) {
  const int T = ks * ks;
  const size_t total = (size_t)nprime * cout_f * T;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int jj = i & 3;
    size_t r = i >> 2;
    const int n = r % nprime; r /= nprime;
    const int kq = r & 3; r >>= 2;
    const int tap = r % T;
    const int chunk = r / T;
    const int co = chunk * 16 + kq * 4 + jj;
    wp[i] = w[((size_t)co * cin_f + ci0 + n) * T + (T - 1 - tap)];
  }
}

extern "C" int tsr_pack_conv_weight_dgrad(const float* w_oihw, float* w_packed, int cout, int cin, int ks,
                                          int ci0, int nprime, void* stream) {
  if (!w_oihw || !w_packed || (cout & 15) || (nprime != 64 && nprime != 128) || ci0 < 0 || ci0 + nprime > cin ||
      (ks != 1 && ks != 3 && ks != 5))
    return TSR_ERR_ARG;
  const size_t total = (size_t)nprime * cout * ks * ks;
  const int grid = (int)((total + 255) / 256);
  hipLaunchKernelGGL(pack_conv_weight_dgrad_kernel, dim3(grid > 4096 ? 4096 : grid), dim3(256), 0,
                     (hipStream_t)stream, w_oihw, w_packed, cout, cin, ks, ci0, nprime);
  return tsr_check_launch();
}

extern "C" int tsr_pack_conv_weight(const float* w_oihw, float* w_packed, int cout, int cin, int ks,
                                    void* stream) {
  if (!w_oihw || !w_packed || (cin & 15) || (cout != 64 && cout != 128) ||
      (ks != 1 && ks != 3 && ks != 5))
    return TSR_ERR_ARG;
  const size_t total = (size_t)cout * cin * ks * ks;
  const int grid = (int)((total + 255) / 256);
  hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(grid > 4096 ? 4096 : grid), dim3(256), 0,
                     (hipStream_t)stream, w_oihw, w_packed, cout, cin, ks);
  return tsr_check_launch();
}

template <bool EXT>
static int dispatch_conv(const ConvArgs& a, int cout, int ks, hipStream_t st) {
  if (cout == 64) {
    if (ks == 1) return launch_conv<1, 64, EXT>(a, st);
    if (ks == 3) return launch_conv<3, 64, EXT>(a, st);
    if (ks == 5) return launch_conv<5, 64, EXT>(a, st);
  } else if (cout == 128) {
    if (ks == 1) return launch_conv<1, 128, EXT>(a, st);
    if (ks == 3) return launch_conv<3, 128, EXT>(a, st);
    if (ks == 5) return launch_conv<5, 128, EXT>(a, st);
  }
  return TSR_ERR_ARG;
}

static int check_slices(int cin, int in_ctot, int in_coff, int cout, int out_ctot, int out_coff) {
  if ((cin & 15) || (in_ctot & 15) || (in_coff & 15) || (out_ctot & 15) || (out_coff & 15) || cin <= 0 ||
      in_coff + cin > in_ctot || out_coff + cout > out_ctot)
    return TSR_ERR_ARG;
  return TSR_OK;
}

extern "C" int tsr_conv2d_fwd(const float* in, int in_ctot, int in_coff, int cin,
                              const float* w_packed, int cout, int ks,
                              const float* scale, const float* shift,
                              const float* res, int res_ctot, int res_coff,
                              float* out, int out_ctot, int out_coff, int relu,
                              int B, int H, int W, void* stream) {
  if (!in || !w_packed || !out || B <= 0 || H <= 0 || W <= 0) return TSR_ERR_ARG;
  if (check_slices(cin, in_ctot, in_coff, cout, out_ctot, out_coff)) return TSR_ERR_ARG;
  if (res && ((res_ctot & 15) || (res_coff & 15) || res_coff + cout > res_ctot)) return TSR_ERR_ARG;
  ConvArgs a = {};
  a.in = in; a.in_ctot = in_ctot; a.in_coff = in_coff; a.cin = cin;
  a.wp = w_packed; a.scale = scale; a.shift = shift;
  a.res = res; a.res_ctot = res_ctot; a.res_coff = res_coff;
  a.out = out; a.out_ctot = out_ctot; a.out_coff = out_coff; a.relu = relu;
  a.B = B; a.H = H; a.W = W;
  a.tiles_x = (W + 7) / 8; a.tiles_y = (H + 7) / 8;
  return dispatch_conv<false>(a, cout, ks, (hipStream_t)stream);
}

int tsr_conv_f16s_images(int cout, int ks);     // conv_mfma_split16.hip: images per workgroup of the fp16x3 kernel in use

// Number of (workgroup, image) slab entries a tsr_conv2d_ex launch of this shape emits.
extern "C" int tsr_conv2d_slab_entries(int B, int H, int W) {
  return ((B + 1) / 2) * ((W + 7) / 8) * ((H + 7) / 8) * 2;
}

// Slab entries a tsr_conv2d_ex launch with these parameters writes: one per (workgroup, image slot).  The fp16-split
// 3x3 / 5x5 kernels (tsr_conv_f16s_images: 4 images, 2 for the 1x1 and TSR_CONV_K32_256) and every one-plane 3x3 / 5x5 kernel put 4 images in a
// workgroup, everything else 2 (must match launch_bf16s in conv_mfma_split16.hip).
int tsr_dgrad1x1_b16k_grid(int B, int H, int W);           // conv1x1_b16k.hip
extern "C" int tsr_conv2d_slab_entries_ex(int B, int H, int W, int cout, int ks, int nsplit) {
  if (nsplit == -3 && ks == 1) return tsr_dgrad1x1_b16k_grid(B, H, W);        // one entry per workgroup of the streaming kernel
  if (nsplit == -3 || nsplit == -4) nsplit = -1;
  const int img = nsplit == -2 ? tsr_conv_f16s_images(cout, ks) : ((ks > 1 && (nsplit == 1 || nsplit == -1)) ? 4 : 2);
  return ((B + img - 1) / img) * ((W + 7) / 8) * ((H + 7) / 8) * img;
}

int tsr_conv2d_ex_bf16s(const ConvArgs& a, int cout, int ks, int nsplit, hipStream_t st);   // conv_mfma_split16.hip

extern "C" int tsr_conv2d_ex(const tsr_conv_desc* d, void* stream) {
  if (!d || !d->in || !d->w_packed || !d->out || d->B <= 0 || d->H <= 0 || d->W <= 0) return TSR_ERR_ARG;
  if (check_slices(d->cin, d->in_ctot, d->in_coff, d->cout, d->out_ctot, d->out_coff)) return TSR_ERR_ARG;
  if (d->res && ((d->res_ctot & 15) || (d->res_coff & 15) || d->res_coff + d->cout > d->res_ctot))
    return TSR_ERR_ARG;
  if (d->epi_mode < 0 || d->epi_mode > 2) return TSR_ERR_ARG;
  if (d->epi_mode == 1 && (!d->slab || !d->slab_cnt)) return TSR_ERR_ARG;
  if (d->epi_mode == 2 && (!d->mask || (d->mask_ctot & 15) || (d->mask_coff & 15) ||
                           d->mask_coff + d->cout > d->mask_ctot || (d->bn_a && (!d->bn_b || !d->slab))))
    return TSR_ERR_ARG;
  if ((d->in_scale != nullptr) != (d->in_shift != nullptr)) return TSR_ERR_ARG;
  if ((d->res_scale != nullptr) != (d->res_shift != nullptr)) return TSR_ERR_ARG;
  ConvArgs a = {};
  a.in = d->in; a.in_ctot = d->in_ctot; a.in_coff = d->in_coff; a.cin = d->cin;
  a.wp = d->w_packed; a.scale = d->scale; a.shift = d->shift;
  a.res = d->res; a.res_ctot = d->res_ctot; a.res_coff = d->res_coff;
  a.out = d->out; a.out_ctot = d->out_ctot; a.out_coff = d->out_coff; a.relu = d->relu;
  a.B = d->B; a.H = d->H; a.W = d->W;
  a.tiles_x = (d->W + 7) / 8; a.tiles_y = (d->H + 7) / 8;
  a.in_scale = d->in_scale; a.in_shift = d->in_shift;
  a.res_scale = d->res_scale; a.res_shift = d->res_shift;
  a.epi_mode = d->epi_mode;
  a.mask = d->mask; a.mask_ctot = d->mask_ctot; a.mask_coff = d->mask_coff;
  a.mask_scale = d->mask_scale; a.mask_shift = d->mask_shift;
  a.bn_a = d->bn_a; a.bn_b = d->bn_b;
  a.slab = d->slab; a.slab_cnt = d->slab_cnt;
  if (d->nsplit < -4 || d->nsplit > 3) return TSR_ERR_ARG;
  a.in_amax = d->in_amax; a.w_inv_scale = d->w_inv_scale; a.out_amax = d->out_amax; a.w_amax = d->w_amax;
  if (d->nsplit == -2 && (!d->in_amax || (!d->w_amax && !(d->w_inv_scale > 0.f)))) return TSR_ERR_ARG;
  if (d->nsplit != 0) return tsr_conv2d_ex_bf16s(a, d->cout, d->ks, d->nsplit, (hipStream_t)stream);
  return dispatch_conv<true>(a, d->cout, d->ks, (hipStream_t)stream);
}

// tPSFNet on MI355X: the tactile point-spread-function forward model
// (reference model/tPSFNet.py:78-141) and its backward, batched -- one workgroup per sample, no
// python loop over the batch (the reference iterates samples at :118-125).
//
//   alphaBeta = MLP(x)                           (tsr_sgemm below: Linear+ReLU x3, Linear+Softplus)
// The PSF convolution / pooling kernels (forward and backward) live in tpsf_mfma.hip.
//   psf  = a * exp(-sdf^2 / b^2),  sdf = 10*dist((u,v),(49,49))/dist_max            (:43-46,78-83)
//   HR   = conv2d(ZeroPad48(depth), psf, padding=1)  -> 100x100                      (:85-89)
//   HR[mask] = max(HR outside mask, 0), mask = depth > depth.max() - 1e-3            (:86,95-97)
//   LRd[i][j] = 1e-4 * sum(HR * m'_ij), m' = minmax-normalised exp(-msdf^2 / gamma)  (:129-141)
//
// The PSF is separable, psf[u][v] = a*g(u)*g(v), g(t) = exp(-Kp (t-49)^2 / b^2), Kp = 100/(2*49^2),
// so the 99x99 per-sample convolution (196 MFLOP direct) is two 99-tap passes (3.96 MFLOP) held in
// LDS; the 16 Gaussian masks are separable too, and their global min-max normaliser is analytic:
// max = 1 (d = 0), min = exp(-100/gamma) (the farthest pixel, where the scaled sdf is exactly 10).
// HBM traffic: depth in (40 KB), HR out (40 KB), psf out (39 KB) per sample -> HBM-bound.
#include "tsr_common.h"

#define PS 99
#define HS 100
#define NPIX (HS * HS)
#define KP (100.0f / 4802.0f)      // 10^2 / (49^2 + 49^2)
#define KM (100.0f / 15138.0f)     // 10^2 / (87^2 + 87^2): farthest pixel (99,99) from centre (12,12)

// ------------------------------------------------------------------------------------------
// strided SGEMM with fused bias + activation on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32
// products, fp32 accumulate), for the 48-256-1024-256-3 MLP and its backward:
//   C[i][j] = act( sum_k A(i,k) B(k,j) + bias[j] ),  A(i,k) = A[i*sa0+k*sa1], B(k,j) = B[k*sb0+j*sb1]
// act: 0 none, 1 ReLU, 2 Softplus(beta=1, threshold=20).  64x64 tile per workgroup (4 waves x one 32x32 block),
// K in steps of 16 through a double-buffered LDS tile; blockIdx.z selects a K range (split-K partial sums).
// ------------------------------------------------------------------------------------------
#define SG_KT 16
__global__ __launch_bounds__(256) void sgemm_mfma_kernel(const float* __restrict__ A, long sa0, long sa1,
                                                         const float* __restrict__ Bm, long sb0, long sb1,
                                                         const float* __restrict__ bias, float* __restrict__ C, int M,
                                                         int N, int K, int act, int kchunk) {
  __shared__ float As[2][SG_KT][65], Bs[2][SG_KT][65];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, li = lane & 31;
  const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  const int kbeg = blockIdx.z * kchunk;
  const int kend = kbeg + kchunk < K ? kbeg + kchunk : K;
  if (gridDim.z > 1) C += (size_t)blockIdx.z * M * N;
  // global -> LDS element mapping: consecutive lanes walk the unit-stride dimension of each operand
  const bool a_kfast = sa1 == 1, b_kfast = sb0 == 1;
  int am[4], ak[4], bn[4], bk[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = tid + 256 * q;
    am[q] = a_kfast ? e / SG_KT : e & 63;
    ak[q] = a_kfast ? e % SG_KT : e >> 6;
    bn[q] = b_kfast ? e / SG_KT : e & 63;
    bk[q] = b_kfast ? e % SG_KT : e >> 6;
  }
  float ra[4], rb[4];
  auto gload = [&](int k0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int gi = i0 + am[q], gk = k0 + ak[q];
      ra[q] = (gi < M && gk < kend) ? A[gi * sa0 + gk * sa1] : 0.f;
      const int gj = j0 + bn[q], gk2 = k0 + bk[q];
      rb[q] = (gj < N && gk2 < kend) ? Bm[gk2 * sb0 + gj * sb1] : 0.f;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      As[buf][ak[q]][am[q]] = ra[q];
      Bs[buf][bk[q]][bn[q]] = rb[q];
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  gload(kbeg);
  lstore(0);
  __syncthreads();
  int buf = 0;
  for (int k0 = kbeg; k0 < kend; k0 += SG_KT) {
    const bool more = k0 + SG_KT < kend;
    if (more) gload(k0 + SG_KT);
#pragma unroll
    for (int kk = 0; kk < SG_KT; kk += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[buf][kk + h][wm * 32 + li], Bs[buf][kk + h][wn * 32 + li], acc, 0, 0,
                                                 0);
    if (more) lstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  const int gj = j0 + wn * 32 + li;
  if (gj < N) {
    const float bv = bias ? bias[gj] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int gi = i0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (gi < M) {
        float v = acc[r] + bv;
        if (act == 1) v = tsr_relu(v);
        else if (act == 2) v = v > 20.f ? v : log1pf(expf(v));
        C[(size_t)gi * N + gj] = v;
      }
    }
  }
}

extern "C" int tsr_sgemm(const float* A, long long sa0, long long sa1, const float* B, long long sb0, long long sb1,
                         const float* bias, float* C, int M, int N, int K, int act, void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || act < 0 || act > 2) return TSR_ERR_ARG;
  hipLaunchKernelGGL(sgemm_mfma_kernel, dim3((N + 63) / 64, (M + 63) / 64, 1), dim3(256), 0, (hipStream_t)stream, A,
                     (long)sa0, (long)sa1, B, (long)sb0, (long)sb1, bias, C, M, N, K, act, K);
  return tsr_check_launch();
}

// Split-K form for reductions over the batch (dW = dy^T x, db = 1^T dy): slab[s][M][N] holds the partial product of
// K range s; tsr_reduce_splits adds the nsplit slabs in a fixed order (deterministic, no float atomics).
extern "C" int tsr_sgemm_splitk(const float* A, long long sa0, long long sa1, const float* B, long long sb0,
                                long long sb1, float* slab, int M, int N, int K, int nsplit, void* stream) {
  if (!A || !B || !slab || M <= 0 || N <= 0 || K <= 0 || nsplit <= 0 || nsplit > 65535) return TSR_ERR_ARG;
  int kchunk = (K + nsplit - 1) / nsplit;
  kchunk = (kchunk + SG_KT - 1) / SG_KT * SG_KT;          // whole K steps per split (trailing splits may be empty: zeros)
  hipLaunchKernelGGL(sgemm_mfma_kernel, dim3((N + 63) / 64, (M + 63) / 64, nsplit), dim3(256), 0, (hipStream_t)stream,
                     A, (long)sa0, (long)sa1, B, (long)sb0, (long)sb1, (const float*)nullptr, slab, M, N, K, 0, kchunk);
  return tsr_check_launch();
}

// dz = dy * act'(.)  in place, from the stored OUTPUT y of the activation: mode 1 ReLU (y > 0);
// mode 2 Softplus: y = log(1+e^z)  =>  sigmoid(z) = 1 - e^{-y} = -expm1(-y).
__global__ void act_bwd_kernel(float* __restrict__ dy, const float* __restrict__ ref, size_t n, int mode) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float r = ref[i];
    if (mode == 1) dy[i] = r > 0.f ? dy[i] : 0.f;
    else dy[i] = dy[i] * (-expm1f(-r));
  }
}

extern "C" int tsr_act_bwd(float* dy, const float* ref, long long n, int mode, void* stream) {
  if (!dy || !ref || n <= 0 || (mode != 1 && mode != 2)) return TSR_ERR_ARG;
  const size_t g = ((size_t)n + 255) / 256;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(g > 4096 ? 4096 : (int)g), dim3(256), 0, (hipStream_t)stream, dy, ref,
                     (size_t)n, mode);
  return tsr_check_launch();
}

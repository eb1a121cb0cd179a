// tPSFNet on MI355X: the tactile point-spread-function forward model
// (reference model/tPSFNet.py:78-141) and its backward, batched -- one workgroup per sample, no
// python loop over the batch (the reference iterates samples at :118-125).
//
//   alphaBeta = MLP(x)                           (tsr_sgemm below: Linear+ReLU x3, Linear+Softplus)
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
// small strided SGEMM with fused bias + activation, for the 48-256-1024-256-3 MLP and its backward:
//   C[i][j] = act( sum_k A(i,k) B(k,j) + bias[j] ),  A(i,k) = A[i*sa0+k*sa1], B(k,j) = B[k*sb0+j*sb1]
// act: 0 none, 1 ReLU, 2 Softplus(beta=1, threshold=20).  64x64 tile, 4x4 per thread, fp32 fma.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sgemm_kernel(const float* __restrict__ A, long sa0, long sa1,
                                                    const float* __restrict__ Bm, long sb0, long sb1,
                                                    const float* __restrict__ bias, float* __restrict__ C, int M,
                                                    int N, int K, int act) {
  __shared__ float As[16][65], Bs[16][65];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  float acc[4][4] = {};
  for (int k0 = 0; k0 < K; k0 += 16) {
    for (int t = tid; t < 16 * 64; t += 256) {
      const int kk = t & 15, ii = t >> 4;
      const int gi = i0 + ii, gk = k0 + kk;
      As[kk][ii] = (gi < M && gk < K) ? A[gi * sa0 + gk * sa1] : 0.f;
      const int gj = j0 + ii;
      Bs[kk][ii] = (gj < N && gk < K) ? Bm[gk * sb0 + gj * sb1] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float a[4], bb[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { a[r] = As[kk][ty * 4 + r]; bb[r] = Bs[kk][tx * 4 + r]; }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = fmaf(a[r], bb[c], acc[r][c]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int gi = i0 + ty * 4 + r, gj = j0 + tx * 4 + c;
      if (gi < M && gj < N) {
        float v = acc[r][c] + (bias ? bias[gj] : 0.f);
        if (act == 1) v = fmaxf(v, 0.f);
        else if (act == 2) v = v > 20.f ? v : log1pf(expf(v));
        C[(size_t)gi * N + gj] = v;
      }
    }
}

extern "C" int tsr_sgemm(const float* A, long long sa0, long long sa1, const float* B, long long sb0, long long sb1,
                         const float* bias, float* C, int M, int N, int K, int act, void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || act < 0 || act > 2) return TSR_ERR_ARG;
  hipLaunchKernelGGL(sgemm_kernel, dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, (hipStream_t)stream, A, (long)sa0,
                     (long)sa1, B, (long)sb0, (long)sb1, bias, C, M, N, K, act);
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

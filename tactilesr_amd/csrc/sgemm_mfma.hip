// Strided SGEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate) for the
// 48-256-1024-256-3 MLP of tPSFNet (reference model/tPSFNet.py:26-36) and its backward:
//
//   C[i][j] = epi( sum_k A(i,k) B(k,j) ),   A(i,k) = A[i*sa0 + k*sa1],  B(k,j) = B[k*sb0 + j*sb1]
//   epi(v)  = mask( act(v + bias[j]) ):  act 0 none / 1 ReLU / 2 Softplus(beta 1, threshold 20);
//             mask (optional): v if ref[i][j] > 0 else 0 -- the ReLU backward of the PREVIOUS layer fused into the dx GEMM.
//
// Tile: BM x 128 x 32 per 256-thread workgroup (BM = 128, or 64 when that is what fills the chip), waves 2 x 2, each
// wave (BM/2) x 64 = BM/64 x 2 MFMA blocks of 32 x 32.  Operand tiles sit in LDS k-major ([k][m], row stride BM + 4
// floats: the MFMA's A operand -- lane li supplies row li at k = h -- is then a conflict-free ds_read_b32 of
// consecutive words); global loads are 16 B per lane along whichever dimension of the operand has unit stride
// (k-fast rows are scattered into the k-major tile by four ds_write_b32, 2-way bank conflicts at most: free), and the
// next K step's loads are in flight during the current step's MFMAs (register double buffer + two LDS buffers, one
// barrier per step).  blockIdx.z = K range (split-K partial sums for the reductions over the batch): partial s lands at
// C + s * split_stride, to be added in a fixed order by tsr_reduce_splits (deterministic, no float atomics).
// Operands whose unit-stride dimension is not 16-B loadable (ragged K, odd strides) take the scalar generic path.
// Measured at B = 8192 (tools/sgemm_microbench.py): the six 8192 x 1024 x 256-sized GEMMs of a train step 53-65 us each =
// 66-81 TF (the single-launch 64 x 64 x 16 kernel it replaces: 46-64 TF); K steps of 16, 64-row tiles everywhere and both
// together measured 2-7 % slower (same-box A/B).
#include "tsr_common.h"
#include "tactilesr_hip.h"

#define SG_KT 32
#define SG_BN 128

struct SgemmArgs {
  const float* A; long sa0, sa1;
  const float* B; long sb0, sb1;
  const float* bias; const float* mask; float* C;
  int M, N, K, act, kchunk;
  long split_stride;
};

// AF / BF: 0 = generic scalar loads, 1 = unit stride along k (16-B loads along k), 2 = unit stride along m / n
template <int BM, int AF, int BF>
__global__ __launch_bounds__(256) void sgemm_tile_kernel(const SgemmArgs g) {
  // row strides: odd for k-fast operands (their four scattered ds_write_b32 then hit 32 distinct banks), a multiple of 4
  // for the others (16-B aligned ds_write_b128); two 32-deep buffers of both tiles = 67 KB: two workgroups per CU
  constexpr int LDA = BM + (AF == 1 ? 1 : 4), LDB = SG_BN + (BF == 1 ? 1 : 4);
  constexpr int MT = BM / 64;                      // 32-row blocks per wave
  __shared__ __attribute__((aligned(16))) float As[2][SG_KT][LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][SG_KT][LDB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, li = lane & 31;
  const int i0 = blockIdx.y * BM, j0 = blockIdx.x * SG_BN;
  const int kbeg = blockIdx.z * g.kchunk;
  const int kend = kbeg + g.kchunk < g.K ? kbeg + g.kchunk : g.K;
  float* C = g.C + (size_t)blockIdx.z * g.split_stride;

  constexpr int NA = BM * SG_KT / 4 / 256, NBV = SG_BN * SG_KT / 4 / 256;      // 16-B items per thread
  f32x4 ra[NA], rb[NBV];
  // item -> (row-in-tile, k) of its first element
  auto a_item = [&](int q, int& m, int& k) {
    const int e = tid + 256 * q;
    if (AF == 1) { m = e / (SG_KT / 4); k = (e % (SG_KT / 4)) * 4; }     // SG_KT/4 lanes cover the k of a row
    else { k = e / (BM / 4); m = (e - k * (BM / 4)) * 4; }        // BM/4 lanes cover a k row
  };
  auto b_item = [&](int q, int& n, int& k) {
    const int e = tid + 256 * q;
    if (BF == 1) { n = e / (SG_KT / 4); k = (e % (SG_KT / 4)) * 4; }
    else { k = e / (SG_BN / 4); n = (e - k * (SG_BN / 4)) * 4; }
  };
  // Branch-free loads (hipcc wraps a conditional load in its own exec-masked branch with a wait in front of it, which
  // turns the loads of a step into serial round trips): an out-of-range item reads element 0 of its operand and is zeroed
  // by a select.  The 16-B paths are only chosen when the vectorised extent is a multiple of 4 (sgemm_dispatch), so the
  // four elements of an item are inside or outside together.
  auto gload = [&](int k0) {
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      int m, k;
      a_item(q, m, k);
      const int gi = i0 + m, gk = k0 + k;
      f32x4 v;
      if (AF == 1) {
        const bool ok = gi < g.M && gk < kend;
        v = *(const f32x4*)(g.A + (ok ? (size_t)gi * g.sa0 + gk : (size_t)0));
        v = ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
      } else if (AF == 2) {
        const bool ok = gk < kend && gi < g.M;
        v = *(const f32x4*)(g.A + (ok ? (size_t)gk * g.sa1 + gi : (size_t)0));
        v = ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const bool ok = gk < kend && gi + c < g.M;
          const float t = g.A[ok ? (size_t)(gi + c) * g.sa0 + (size_t)gk * g.sa1 : (size_t)0];
          v[c] = ok ? t : 0.f;
        }
      }
      ra[q] = v;
    }
#pragma unroll
    for (int q = 0; q < NBV; ++q) {
      int n, k;
      b_item(q, n, k);
      const int gj = j0 + n, gk = k0 + k;
      f32x4 v;
      if (BF == 1) {
        const bool ok = gj < g.N && gk < kend;
        v = *(const f32x4*)(g.B + (ok ? (size_t)gj * g.sb1 + gk : (size_t)0));
        v = ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
      } else if (BF == 2) {
        const bool ok = gk < kend && gj < g.N;
        v = *(const f32x4*)(g.B + (ok ? (size_t)gk * g.sb0 + gj : (size_t)0));
        v = ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const bool ok = gk < kend && gj + c < g.N;
          const float t = g.B[ok ? (size_t)gk * g.sb0 + (size_t)(gj + c) * g.sb1 : (size_t)0];
          v[c] = ok ? t : 0.f;
        }
      }
      rb[q] = v;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      int m, k;
      a_item(q, m, k);
      if (AF == 1) {
#pragma unroll
        for (int c = 0; c < 4; ++c) As[buf][k + c][m] = ra[q][c];
      } else {
        *(f32x4*)&As[buf][k][m] = ra[q];
      }
    }
#pragma unroll
    for (int q = 0; q < NBV; ++q) {
      int n, k;
      b_item(q, n, k);
      if (BF == 1) {
#pragma unroll
        for (int c = 0; c < 4; ++c) Bs[buf][k + c][n] = rb[q][c];
      } else {
        *(f32x4*)&Bs[buf][k][n] = rb[q];
      }
    }
  };

  f32x16 acc[MT][2];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  if (kbeg < kend) {
    gload(kbeg);
    lstore(0);
  }
  __syncthreads();
  int buf = 0;
  for (int k0 = kbeg; k0 < kend; k0 += SG_KT) {
    const bool more = k0 + SG_KT < kend;
    if (more) gload(k0 + SG_KT);
#pragma unroll
    for (int kk = 0; kk < SG_KT; kk += 2) {
      float av[MT], bv[2];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) av[mt] = As[buf][kk + h][wm * (BM / 2) + mt * 32 + li];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) bv[nt] = Bs[buf][kk + h][wn * 64 + nt * 32 + li];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt], bv[nt], acc[mt][nt], 0, 0, 0);
    }
    if (more) lstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  // epilogue: a tile's 16 mask values are requested together (clamped row: no branch, no wait per element)
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int gj = j0 + wn * 64 + nt * 32 + li;
    const int gjc = gj < g.N ? gj : g.N - 1;
    const float bvj = g.bias ? g.bias[gjc] : 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int gi0 = i0 + wm * (BM / 2) + mt * 32 + 4 * h;
      float mk[16];
      if (g.mask) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int gi = gi0 + (r & 3) + 8 * (r >> 2);
          mk[r] = g.mask[(size_t)(gi < g.M ? gi : g.M - 1) * g.N + gjc];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gi = gi0 + (r & 3) + 8 * (r >> 2);
        float v = acc[mt][nt][r] + bvj;
        if (g.act == 1) v = tsr_relu(v);
        else if (g.act == 2) v = v > 20.f ? v : log1pf(expf(v));
        if (g.mask) v = mk[r] > 0.f ? v : 0.f;
        if (gi < g.M && gj < g.N) C[(size_t)gi * g.N + gj] = v;
      }
    }
  }
}

template <int BM, int AF>
static void sgemm_launch_b(const SgemmArgs& g, int bf, dim3 grid, hipStream_t st) {
  if (bf == 1) hipLaunchKernelGGL((sgemm_tile_kernel<BM, AF, 1>), grid, dim3(256), 0, st, g);
  else if (bf == 2) hipLaunchKernelGGL((sgemm_tile_kernel<BM, AF, 2>), grid, dim3(256), 0, st, g);
  else hipLaunchKernelGGL((sgemm_tile_kernel<BM, AF, 0>), grid, dim3(256), 0, st, g);
}
template <int BM>
static void sgemm_launch_a(const SgemmArgs& g, int af, int bf, dim3 grid, hipStream_t st) {
  if (af == 1) sgemm_launch_b<BM, 1>(g, bf, grid, st);
  else if (af == 2) sgemm_launch_b<BM, 2>(g, bf, grid, st);
  else sgemm_launch_b<BM, 0>(g, bf, grid, st);
}

static bool al16(const void* p) { return (((size_t)p) & 15) == 0; }

static int sgemm_dispatch(SgemmArgs g, int nsplit, hipStream_t st) {
  // which dimension of each operand is 16-B loadable: the unit-stride one, with the other stride a multiple of 4 floats
  // (and the vectorised extent itself a multiple of 4: the four elements of a 16-B item are then in range together)
  const int af = (g.sa1 == 1 && (g.sa0 & 3) == 0 && (g.K & 3) == 0 && al16(g.A)) ? 1
               : ((g.sa0 == 1 && (g.sa1 & 3) == 0 && (g.M & 3) == 0 && al16(g.A)) ? 2 : 0);
  const int bf = (g.sb0 == 1 && (g.sb1 & 3) == 0 && (g.K & 3) == 0 && al16(g.B)) ? 1
               : ((g.sb1 == 1 && (g.sb0 & 3) == 0 && (g.N & 3) == 0 && al16(g.B)) ? 2 : 0);
  const int nx = (g.N + SG_BN - 1) / SG_BN;
  // 128-row tiles unless they leave the chip less than half full and 64-row tiles do better
  const long wg128 = (long)nx * ((g.M + 127) / 128) * nsplit;
  const bool small = wg128 < 384 && g.M > 64;
  if (small) {
    dim3 grid(nx, (g.M + 63) / 64, nsplit);
    sgemm_launch_a<64>(g, af, bf, grid, st);
  } else {
    dim3 grid(nx, (g.M + 127) / 128, nsplit);
    sgemm_launch_a<128>(g, af, bf, grid, st);
  }
  return tsr_check_launch();
}

extern "C" int tsr_sgemm(const float* A, long long sa0, long long sa1, const float* B, long long sb0, long long sb1,
                         const float* bias, float* C, int M, int N, int K, int act, void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || act < 0 || act > 2) return TSR_ERR_ARG;
  SgemmArgs g = {A, (long)sa0, (long)sa1, B, (long)sb0, (long)sb1, bias, nullptr, C, M, N, K, act, K, 0};
  return sgemm_dispatch(g, 1, (hipStream_t)stream);
}

extern "C" int tsr_sgemm_masked(const float* A, long long sa0, long long sa1, const float* B, long long sb0,
                                long long sb1, const float* mask_ref, float* C, int M, int N, int K, void* stream) {
  if (!A || !B || !C || !mask_ref || M <= 0 || N <= 0 || K <= 0) return TSR_ERR_ARG;
  SgemmArgs g = {A, (long)sa0, (long)sa1, B, (long)sb0, (long)sb1, nullptr, mask_ref, C, M, N, K, 0, K, 0};
  return sgemm_dispatch(g, 1, (hipStream_t)stream);
}

static int splitk_chunk(int K, int nsplit) {
  int kchunk = (K + nsplit - 1) / nsplit;
  return (kchunk + SG_KT - 1) / SG_KT * SG_KT;          // whole K steps per split (trailing splits may be empty: zeros)
}

extern "C" int tsr_sgemm_splitk(const float* A, long long sa0, long long sa1, const float* B, long long sb0,
                                long long sb1, float* slab, int M, int N, int K, int nsplit, void* stream) {
  if (!A || !B || !slab || M <= 0 || N <= 0 || K <= 0 || nsplit <= 0 || nsplit > 65535) return TSR_ERR_ARG;
  SgemmArgs g = {A, (long)sa0, (long)sa1, B, (long)sb0, (long)sb1, nullptr, nullptr, slab, M, N, K, 0,
                 splitk_chunk(K, nsplit), (long)M * N};
  return sgemm_dispatch(g, nsplit, (hipStream_t)stream);
}

extern "C" int tsr_sgemm_splitk_strided(const float* A, long long sa0, long long sa1, const float* B, long long sb0,
                                        long long sb1, float* slab, long long split_stride, int M, int N, int K,
                                        int nsplit, void* stream) {
  if (!A || !B || !slab || M <= 0 || N <= 0 || K <= 0 || nsplit <= 0 || nsplit > 65535 || split_stride < (long long)M * N)
    return TSR_ERR_ARG;
  SgemmArgs g = {A, (long)sa0, (long)sa1, B, (long)sb0, (long)sb1, nullptr, nullptr, slab, M, N, K, 0,
                 splitk_chunk(K, nsplit), (long)split_stride};
  return sgemm_dispatch(g, nsplit, (hipStream_t)stream);
}

// Column sums over a row range per split (db = 1^T dy): slab[s * split_stride + j] = sum_{i in range s} Y[i][j]; same K
// ranges as the split-K GEMM.  A workgroup owns 64 columns of one range: 4 row phases x 64 columns (a wave reads one
// 256-B row segment per instruction), four independent chains per thread, the phases added in a fixed order through LDS
// (deterministic).  (One thread per column over the whole range left N = 1024 with 128 latency-bound workgroups: 19 us.)
__global__ __launch_bounds__(256) void colsum_splitk_kernel(const float* __restrict__ Y, float* __restrict__ slab,
                                                            long split_stride, int M, int N, int kchunk) {
  __shared__ float part[4][64];
  const int c = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + c;
  const int r0 = blockIdx.y * kchunk;
  const int r1 = r0 + kchunk < M ? r0 + kchunk : M;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (j < N) {
    int i = r0 + ph;
    for (; i + 12 < r1; i += 16) {
      s0 += Y[(size_t)i * N + j];
      s1 += Y[(size_t)(i + 4) * N + j];
      s2 += Y[(size_t)(i + 8) * N + j];
      s3 += Y[(size_t)(i + 12) * N + j];
    }
    for (; i < r1; i += 4) s0 += Y[(size_t)i * N + j];
  }
  part[ph][c] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ph == 0 && j < N) slab[(size_t)blockIdx.y * split_stride + j] = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
}

extern "C" int tsr_colsum_splitk(const float* Y, float* slab, long long split_stride, int M, int N, int nsplit,
                                 void* stream) {
  if (!Y || !slab || M <= 0 || N <= 0 || nsplit <= 0 || nsplit > 65535 || split_stride < N) return TSR_ERR_ARG;
  hipLaunchKernelGGL(colsum_splitk_kernel, dim3((N + 63) / 64, nsplit), dim3(256), 0, (hipStream_t)stream, Y, slab,
                     (long)split_stride, M, N, splitk_chunk(M, nsplit));
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
  const size_t gsz = ((size_t)n + 255) / 256;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(gsz > 4096 ? 4096 : (int)gsz), dim3(256), 0, (hipStream_t)stream, dy, ref,
                     (size_t)n, mode);
  return tsr_check_launch();
}

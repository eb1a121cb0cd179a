// Shared device/host helpers for the tactileSR HIP kernels (gfx950 / MI355X only).
//
// Internal activation layout "CB16": float act[B][C/16][H*W][16]
//   - channel-blocked NHWC: the 16 channels of one block are contiguous (64 B), the
//     pixels of one (image, block) plane are contiguous, so a conv workgroup stages a
//     [halo rows][halo cols][16] slab with fully coalesced 16-B loads and an MFMA A
//     fragment (4 consecutive channels of one pixel) is a single ds_read_b128.
//   - boundary tensors (LR taxels in, SR image out) stay NCHW as the reference's
//     callers hand them over (train/tactileSR_train.py:43-47).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define TSR_CB 16   // channel block

#define TSR_OK 0
#define TSR_ERR_ARG 1
#define TSR_ERR_LAUNCH 2

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int tsr_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? TSR_OK : TSR_ERR_LAUNCH;
}

// Publish max|.| of a launch's output into a device scalar (|v| >= 0, so the uint order of the bit patterns is the
// float order).  One lane per wave calls this.  The plain load first: once a few workgroups have raised the slot
// to (nearly) the tensor's maximum almost every later wave skips the atomic -- 100 k same-address atomics per
// launch otherwise serialize at ~10 ns each in the L2 (measured: 0.8 ms on the 0.6 ms stem kernel).  A stale
// (smaller) cached value only costs a redundant atomic.
// A non-finite wave maximum is NOT published (fmaxf already drops NaN; +Inf would turn the consumer's power-of-two
// operand scale into ~0 for every image of the batch): images without non-finite values keep their exact results, the
// image that holds the Inf / NaN stays non-finite through the planes of the split (Inf - Inf = NaN) -- frames are
// independent in eval mode in the reference too (model/tactileSR_model.py:67-84), and a non-finite loss is what its
// trainer raises on (cpu/trainer.py:280-284).
__device__ __forceinline__ void publish_amax(float* slot, float amax) {
  const unsigned v = __float_as_uint(amax);
  if (v < 0x7f800000u && v > __builtin_nontemporal_load((const unsigned*)slot)) atomicMax((unsigned int*)slot, v);
}

// ReLU that PROPAGATES NaN like torch's relu / clamp_min (fmaxf / v_max_f32 return the non-NaN operand and would turn a
// NaN activation -- a diverged weight, a NaN taxel -- into 0, hiding it from the trainer's non-finite-loss check).
__device__ __forceinline__ float tsr_relu(float v) { return v < 0.f ? 0.f : v; }
// 2*relu(t) in one VALU instruction (v_add_f32 v, v, |v|), NaN-propagating; callers fold the factor 1/2 into the
// power-of-two operand scale that multiplies the value next (staging loops of the split-operand kernels).
__device__ __forceinline__ float tsr_relu_x2(float t) { return t + fabsf(t); }

// 4 consecutive channels of a CB16 tensor whose storage type is fp32 (B16 = false) or bf16 (B16 = true, the
// activation-storage format of the "bf16" configurations); `idx` counts ELEMENTS in both cases.
typedef __bf16 tsr_bf16x4 __attribute__((ext_vector_type(4)));
template <bool B16> __device__ __forceinline__ f32x4 tsr_ld4(const void* base, size_t idx) {
  if constexpr (B16) {
    const tsr_bf16x4 v = *(const tsr_bf16x4*)((const __bf16*)base + idx);
    return (f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  } else {
    return *(const f32x4*)((const float*)base + idx);
  }
}
template <bool B16> __device__ __forceinline__ void tsr_st4(void* base, size_t idx, f32x4 v) {
  if constexpr (B16) {
    tsr_bf16x4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = (__bf16)v[c];
    *(tsr_bf16x4*)((__bf16*)base + idx) = o;
  } else {
    *(f32x4*)((float*)base + idx) = v;
  }
}
template <bool B16> __device__ __forceinline__ float tsr_ld1(const void* base, size_t idx) {
  if constexpr (B16) return (float)((const __bf16*)base)[idx];
  else return ((const float*)base)[idx];
}

__device__ __forceinline__ size_t cb16_index(int b, int c, int pix, int C, int HW) {
  return (((size_t)b * (C >> 4) + (c >> 4)) * HW + pix) * 16 + (c & 15);
}

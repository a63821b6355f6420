// Element type of the ACTIVATION tensors in HBM (channels-last activations, activation gradients): fp32, or bf16
// in the bf16 storage mode (tdx_unet_set_precision(TDX_PREC_BF16): BASELINE.json configs[3]/[4] name bf16; the
// reference itself is fp32-only).  The HBM-bound kernels are templated on it and do their arithmetic in fp32
// either way: a tensor is widened when it is loaded and rounded to nearest-even when it is stored, so the mode
// halves the bytes of every stream and changes nothing else.  Parameters, BatchNorm statistics / scale / shift,
// the time path, the weight-gradient slabs, gradients of parameters and the optimizer stay fp32.
#pragma once
#include "common.h"

typedef __bf16 tdx_bf16;
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

// four consecutive channels <-> float4 (16 B of fp32 or 8 B of bf16)
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const tdx_bf16* p) {
  const bf16x4_t v = *reinterpret_cast<const bf16x4_t*>(p);
  return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
__device__ __forceinline__ void st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(tdx_bf16* p, const float4& v) {
  bf16x4_t o;
  o[0] = (tdx_bf16)v.x; o[1] = (tdx_bf16)v.y; o[2] = (tdx_bf16)v.z; o[3] = (tdx_bf16)v.w;   // v_cvt_pk_bf16_f32: RNE
  *reinterpret_cast<bf16x4_t*>(p) = o;
}
// one element
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const tdx_bf16* p) { return (float)*p; }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(tdx_bf16* p, float v) { *p = (tdx_bf16)v; }

// host-side dispatch on the storage flag: TDX_IO(io16, T, expr-using-T)
#define TDX_IO_DISPATCH(io16, T, ...)      \
  do {                                     \
    if (io16) {                            \
      typedef tdx_bf16 T;                  \
      __VA_ARGS__;                         \
    } else {                               \
      typedef float T;                     \
      __VA_ARGS__;                         \
    }                                      \
  } while (0)

// Shared host/device helpers for libtdx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/tdx.h"

#define TDX_CHECK_LAUNCH()                           \
  do {                                               \
    hipError_t e__ = hipGetLastError();              \
    if (e__ != hipSuccess) return (int)e__;          \
  } while (0)

#define TDX_HIP(call)                                \
  do {                                               \
    hipError_t e__ = (call);                         \
    if (e__ != hipSuccess) return (int)e__;          \
  } while (0)

static inline hipStream_t to_stream(tdx_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- Philox4x32-10 (counter-based RNG), one 128-bit block per call ----------
struct Philox4 {
  uint32_t v[4];
};
__host__ __device__ static inline Philox4 philox4x32_10(uint64_t ctr_lo, uint64_t ctr_hi,
                                                        uint64_t key) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32), c2 = (uint32_t)ctr_hi,
           c3 = (uint32_t)(ctr_hi >> 32);
  uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  Philox4 o; o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
  return o;
}

// Four N(0,1) samples from one Philox block (Box-Muller, fp32).
__device__ static inline float4 philox_normal4(uint64_t ctr, uint64_t stream, uint64_t seed) {
  Philox4 r = philox4x32_10(ctr, stream, seed);
  const float S = 2.3283064365386963e-10f;  // 2^-32
  float u0 = ((float)r.v[0] + 0.5f) * S, u1 = ((float)r.v[1] + 0.5f) * S;
  float u2 = ((float)r.v[2] + 0.5f) * S, u3 = ((float)r.v[3] + 0.5f) * S;
  float ra = sqrtf(-2.0f * __logf(u0)), rb = sqrtf(-2.0f * __logf(u2));
  float sa, ca, sb, cb;
  __sincosf(6.283185307179586f * u1, &sa, &ca);
  __sincosf(6.283185307179586f * u3, &sb, &cb);
  return make_float4(ra * ca, ra * sa, rb * cb, rb * sb);
}

// x' = c1*(x - c2*eps) + sigma*z in the reference's operation order (diffusion.py:272-274): mul, sub, mul, mul, add,
// each rounded separately
__device__ static inline float p_step(float x, float e, float z, float c1, float c2, float sg) {
  float inner = __fsub_rn(x, __fmul_rn(c2, e));
  return __fadd_rn(__fmul_rn(c1, inner), __fmul_rn(sg, z));
}

__device__ static inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

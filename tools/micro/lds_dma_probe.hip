// Probe: does a range-checked (out-of-bounds) lane of `buffer_load_dwordx4 ... lds` write zeros
// to LDS or leave the old bytes?  (decides whether LDS-DMA can do the conv's zero padding)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float* in, float* out, int n_floats) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 4];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = -7.0f;  // sentinel
  __syncthreads();
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, n_floats * 4, 0x00020000);
  unsigned voff = (threadIdx.x & 1) ? 0x80000000u : threadIdx.x * 16u;  // odd lanes out of range
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}
int main() {
  float *in, *out; float h[256], hin[256];
  for (int i = 0; i < 256; ++i) hin[i] = 100.0f + i;
  hipMalloc(&in, 1024); hipMalloc(&out, 1024);
  hipMemcpy(in, hin, 1024, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(in, out, 256);
  hipMemcpy(h, out, 1024, hipMemcpyDeviceToHost);
  printf("lane0 chunk: %g %g %g %g (expect 100..103)\n", h[0], h[1], h[2], h[3]);
  printf("lane1 chunk (OOB): %g %g %g %g (0 = zero-filled, -7 = untouched)\n", h[4], h[5], h[6], h[7]);
  printf("lane2 chunk: %g %g %g %g (expect 108..111)\n", h[8], h[9], h[10], h[11]);
  return 0;
}

// ds_read_b64_tr_b16 addressing probe (gfx950): an LDS image [64 slots][64 channels] of bf16 in 128-byte rows with the
// 16-byte chunk index XOR-ed with ((row >> 1) & 1) << 2 (the image conv3x3_wgrad9_bf16_kernel uses) is read the way
// that kernel reads its MFMA operands, at a row shift, and every lane reports what it received.  Expected: lane l
// gets channel c0 + (l & 31) for the four slots shift + 16 ks + 8 (l >> 5) + 4 j + 0..3.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/tr_read_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x4 __attribute__((__vector_size__(4 * sizeof(__bf16))));
typedef __attribute__((address_space(3))) bf16x4* lds_v4;

__global__ void probe(float* out_row, float* out_ch, int shift, int c0chunk) {
  __shared__ __attribute__((aligned(16))) unsigned char img_row[192 * 128];
  __shared__ __attribute__((aligned(16))) unsigned char img_ch[192 * 128];
  const int lane = threadIdx.x;
  for (int e = lane; e < 192 * 64; e += 64) {
    const int row = e / 64, ch = e % 64;
    const int off = row * 128 + ((((ch >> 3) ^ (((row >> 1) & 1) << 2))) << 4) + (ch & 7) * 2;
    *reinterpret_cast<__bf16*>(img_row + off) = (__bf16)(float)row;
    *reinterpret_cast<__bf16*>(img_ch + off) = (__bf16)(float)ch;
  }
  __syncthreads();
  const int g = lane >> 4, q4 = (lane >> 2) & 3, pp = lane & 3;
  const int lrow = 8 * (g >> 1) + q4 + shift;
  const int chunk = c0chunk + 2 * (g & 1) + (pp >> 1);
  const int off0 = lrow * 128 + ((chunk ^ (((lrow >> 1) & 1) << 2)) << 4) + 8 * (pp & 1);
  for (int ks = 0; ks < 4; ++ks)
    for (int j = 0; j < 2; ++j) {
      const int off = off0 + (16 * ks + 4 * j) * 128;
      const bf16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(img_row + off));
      const bf16x4 c = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(img_ch + off));
      for (int e = 0; e < 4; ++e) {
        out_row[((ks * 2 + j) * 64 + lane) * 4 + e] = (float)r[e];
        out_ch[((ks * 2 + j) * 64 + lane) * 4 + e] = (float)c[e];
      }
    }
}

int main() {
  float *d_row, *d_ch;
  hipMalloc(&d_row, 8 * 64 * 4 * 4);
  hipMalloc(&d_ch, 8 * 64 * 4 * 4);
  int bad = 0;
  for (int shift : {0, 1, 2, 3, 29, 30, 31, 65}) {
    for (int c0chunk : {0, 4}) {
      probe<<<1, 64>>>(d_row, d_ch, shift, c0chunk);
      std::vector<float> r(8 * 64 * 4), c(8 * 64 * 4);
      hipMemcpy(r.data(), d_row, r.size() * 4, hipMemcpyDeviceToHost);
      hipMemcpy(c.data(), d_ch, c.size() * 4, hipMemcpyDeviceToHost);
      for (int ks = 0; ks < 4; ++ks)
        for (int j = 0; j < 2; ++j)
          for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 4; ++e) {
              const int i = ((ks * 2 + j) * 64 + l) * 4 + e;
              const int want_row = shift + 16 * ks + 8 * (l >> 5) + 4 * j + e, want_ch = c0chunk * 8 + (l & 31);
              if ((int)r[i] != want_row || (int)c[i] != want_ch) {
                if (bad < 12) printf("shift %d c0 %d ks %d j %d lane %d e %d: got (row %d, ch %d) want (%d, %d)\n", shift, c0chunk, ks, j, l, e, (int)r[i], (int)c[i], want_row, want_ch);
                ++bad;
              }
            }
    }
  }
  printf("tr_read_probe: %d mismatches\n", bad);
  return bad != 0;
}

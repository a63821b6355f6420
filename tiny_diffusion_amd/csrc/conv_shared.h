// Pieces shared by the fp32 (conv3x3.hip) and bf16 (conv3x3_bf16.hip) implicit-GEMM convolutions:
// the argument block and the epilogue (bias, BatchNorm statistics / BN+ReLU, split-K partials).  The C/D
// register map of the 32x32 MFMA does not depend on the operand type, so one epilogue serves both.
#pragma once
#include "internal.h"

enum { EPI_PLAIN = 0, EPI_STATS = 1, EPI_BNRELU = 2 };

struct ConvArgs {
  const float* in;
  const float* w;  // [Cout][9][Cin]
  const float* bias;
  float* out;
  const float* in_scale;
  const float* in_shift;
  const float* out_scale;
  const float* out_shift;
  float* stats;  // [tilesM][2][Cout]
  int B, H, W, Cin, Cout, M, tilesN;
  int splits, kt_per_split;  // split-K (variant 2): blockIdx.y = split, raw partials to `out`
  int dbg;                   // timing experiments only (tdx_tune_set "conv_dbg"): 1 no barrier,
                             // 2 no LDS stores, 4 no global loads in the main loop -> WRONG results
  unsigned long long* stamps;  // diagnostics (tools/gpu_clock_probe.py): per workgroup {shader cycles, 100 MHz ticks}
                               // around the main loop; null in every product launch
};

// ---------------------------------------------------------------------------
// Shared epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
//   SPLITK      raw partial sums to a.out + split*M*Cout (bias/BN applied by splitk_reduce_kernel)
//   EPI_PLAIN   + bias
//   EPI_BNRELU  relu((acc + bias) * scale + shift)                       (inference)
//   EPI_STATS   + bias, and per-tile per-channel (sum, M2 about the TILE mean) from the accumulators
//               still in registers (two reductions); centred partials are merged with Chan's
//               formula in bn_finalize, so the variance never sees E[y^2]-E[y]^2 cancellation.
template <int BM, int BN, int EPI, bool SPLITK>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[BM / 64][BN / 64],
                                              float* smem, int tile_m, int m0, int n0, int wm, int wn,
                                              int l31, int half, int tid) {
  constexpr int WGM = 2;
  constexpr int WTM = BM / 2, WTN = BN / 2, TM = BM / 64, TN = BN / 64;
  if (SPLITK) {
    float* part = a.out + (size_t)blockIdx.y * a.M * a.Cout;
#pragma unroll
    for (int in = 0; in < TN; ++in) {
      const int col = n0 + wn * WTN + in * 32 + l31;
#pragma unroll
      for (int im = 0; im < TM; ++im)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int p = m0 + wm * WTM + im * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (p < a.M) part[(size_t)p * a.Cout + col] = acc[im][in][r];
        }
    }
    return;
  }
  float csum[TN];
#pragma unroll
  for (int in = 0; in < TN; ++in) {
    const int col = n0 + wn * WTN + in * 32 + l31;
    const float bv = a.bias ? a.bias[col] : 0.f;
    float osc = 1.f, osh = 0.f;
    if (EPI == EPI_BNRELU) {
      osc = a.out_scale[col];
      osh = a.out_shift[col];
    }
    float s = 0.f;
#pragma unroll
    for (int im = 0; im < TM; ++im) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * WTM + im * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int p = m0 + row;
        float v = acc[im][in][r] + bv;
        if (EPI == EPI_BNRELU) v = fmaxf(fmaf(v, osc, osh), 0.f);
        acc[im][in][r] = v;
        if (p < a.M) {
          a.out[(size_t)p * a.Cout + col] = v;
          s += v;
        }
      }
    }
    csum[in] = s;
  }
  if (EPI == EPI_STATS) {
    float* red = smem;  // [WGM][BN] + [BN] means, re-uses the tile buffers (K loop is over)
    const int rows_valid = min(BM, a.M - m0);
    __syncthreads();    // every wave is done with the tile buffers
#pragma unroll
    for (int in = 0; in < TN; ++in) {
      const float s = csum[in] + __shfl_xor(csum[in], 32, 64);
      if (half == 0) red[wm * BN + wn * WTN + in * 32 + l31] = s;
    }
    __syncthreads();
    float* tsum = red + WGM * BN;  // [BN] tile column sums
    for (int c = tid; c < BN; c += 256) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < WGM; ++w) v += red[w * BN + c];
      tsum[c] = v;
      a.stats[((size_t)tile_m * 2 + 0) * a.Cout + n0 + c] = v;
    }
    __syncthreads();
    const float inv_n = 1.0f / (float)rows_valid;
    float cm2[TN];
#pragma unroll
    for (int in = 0; in < TN; ++in) {
      const float mean = tsum[wn * WTN + in * 32 + l31] * inv_n;
      float q = 0.f;
#pragma unroll
      for (int im = 0; im < TM; ++im)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * WTM + im * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (m0 + row < a.M) {
            const float dlt = acc[im][in][r] - mean;
            q = fmaf(dlt, dlt, q);
          }
        }
      cm2[in] = q + __shfl_xor(q, 32, 64);
    }
    __syncthreads();  // everyone has read tsum/red
#pragma unroll
    for (int in = 0; in < TN; ++in)
      if (half == 0) red[wm * BN + wn * WTN + in * 32 + l31] = cm2[in];
    __syncthreads();
    for (int c = tid; c < BN; c += 256) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < WGM; ++w) v += red[w * BN + c];
      a.stats[((size_t)tile_m * 2 + 1) * a.Cout + n0 + c] = v;
    }
  }
}


// ---------------------------------------------------------------------- wgrad
// GEMM: rows = output channels, cols = input channels of ONE tap, K = pixels, split over pixel chunks
// into deterministic slabs [S][Cout][9][Cin] that tdx_conv3x3_wgrad_reduce sums in a fixed order.
struct WgradArgs {
  const float* in;  // (B,H,W,Cin)
  const float* dy;  // (B,H,W,Cout)
  float* slabs;     // [S][Cout][9][Cin]
  const float* in_scale;
  const float* in_shift;
  int B, H, W, Cin, Cout, M, tilesCi, tilesCo, groups, chunk;
  int adv_q, adv_s;  // 32 pixels = adv_q rows + adv_s columns of a W-wide image
};

void tdx_wgrad_plan(int64_t M, int cin, int cout, int* bm, int* bn, int* splits, int* chunk);

// BatchNorm2d (+ReLU) around the MFMA convolutions (diffusion.py:34 and siblings).
//
// Forward: the convolution epilogue leaves per-tile (sum, sumsq) partials; this
// file turns them into per-channel scale/shift that CONSUMERS apply on load
// (relu(y*scale+shift)), so the normalised activation is never written to HBM.
// Backward: two per-channel reductions over (g, y), then one in-place pass.
#include "internal.h"
#include "io16.h"
#include <algorithm>

#define BN_EPS 1e-5f
#define BN_MOMENTUM 0.1f

// Merge per-tile (sum S_t, M2_t about the tile mean) partials.  Chan's parallel-variance
// formula summed over all tiles at once:
//     M2 = sum_t M2_t + sum_t S_t^2 / n_t - S^2 / N ,   S = sum_t S_t
// The last two terms cancel, but they are accumulated in double: with |mean|/std up to 1e4 that
// still leaves 8 significant digits, while the fp32-sensitive part (deviations inside a tile)
// was centred before it was ever summed.  block = 4 channels (one float4 per tile) x 256
// interleaved tile slices, then a fixed-order tree (the kernel sits on the critical path between two
// convolutions: all loads of a thread are in flight together).
__global__ void __launch_bounds__(256)
bn_finalize_kernel(const float* __restrict__ stats, int tiles, int tile_rows, int64_t count, int C,
                   const float* __restrict__ gamma, const float* __restrict__ beta,
                   float* __restrict__ rmean, float* __restrict__ rvar,
                   int64_t* __restrict__ nbt, float* __restrict__ scale, float* __restrict__ shift,
                   float* __restrict__ save_mean, float* __restrict__ save_rstd, int training) {
  __shared__ double red[3][4][256];
  const int c0 = blockIdx.x * 4;      // this block's 4 channels (C % 4 == 0)
  const int sl = threadIdx.x;         // tile slice: every thread reads float4 = 4 channels of one tile
  const int cl = threadIdx.x & 3;
  const int c = c0 + cl;
  if (training) {
    double S[4] = {0, 0, 0, 0}, Q[4] = {0, 0, 0, 0}, R[4] = {0, 0, 0, 0};
    {
      const double inv_full = 1.0 / (double)tile_rows;
      const int64_t last_rows = count - (int64_t)(tiles - 1) * tile_rows;
      const double inv_last = 1.0 / (double)last_rows;
#pragma unroll 4
      for (int t = sl; t < tiles; t += 256) {
        const float4 st = *reinterpret_cast<const float4*>(stats + ((size_t)t * 2 + 0) * C + c0);
        const float4 qt = *reinterpret_cast<const float4*>(stats + ((size_t)t * 2 + 1) * C + c0);
        const double inv = t == tiles - 1 ? inv_last : inv_full;
        const float sv[4] = {st.x, st.y, st.z, st.w}, qv[4] = {qt.x, qt.y, qt.z, qt.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          S[k] += (double)sv[k];
          Q[k] += (double)qv[k];
          R[k] += (double)sv[k] * (double)sv[k] * inv;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { red[0][k][sl] = S[k]; red[1][k][sl] = Q[k]; red[2][k][sl] = R[k]; }
    __syncthreads();
    // fixed-order tree over the 256 slices (deterministic)
    for (int half = 128; half > 0; half >>= 1) {
      if (sl < half) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          red[0][k][sl] += red[0][k][sl + half];
          red[1][k][sl] += red[1][k][sl + half];
          red[2][k][sl] += red[2][k][sl + half];
        }
      }
      __syncthreads();
    }
    if (threadIdx.x < 4 && c < C) {
      const double S = red[0][cl][0], Q = red[1][cl][0], R = red[2][cl][0];
      const double n = (double)count;
      const double mean = S / n;
      double m2 = Q + (R - S * S / n);
      if (m2 < 0.0) m2 = 0.0;
      const double var = m2 / n;  // biased, used to normalise
      const float rstd = (float)(1.0 / sqrt(var + (double)BN_EPS));
      const float sc = gamma[c] * rstd;
      scale[c] = sc;
      shift[c] = beta[c] - (float)mean * sc;
      if (save_mean) save_mean[c] = (float)mean;
      if (save_rstd) save_rstd[c] = rstd;
      if (rmean) {
        const double unbiased = n > 1.0 ? m2 / (n - 1.0) : var;
        rmean[c] = (1.0f - BN_MOMENTUM) * rmean[c] + BN_MOMENTUM * (float)mean;
        rvar[c] = (1.0f - BN_MOMENTUM) * rvar[c] + BN_MOMENTUM * (float)unbiased;
      }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) nbt[0] += 1;
  } else if (threadIdx.x < 4 && c < C) {
    const float mean = rmean[c];
    const float rstd = 1.0f / sqrtf(rvar[c] + BN_EPS);
    const float sc = gamma[c] * rstd;
    scale[c] = sc;
    shift[c] = beta[c] - mean * sc;
    if (save_mean) save_mean[c] = mean;
    if (save_rstd) save_rstd[c] = rstd;
  }
}

extern "C" int tdx_bn_finalize(const float* stats_partial, int tiles, int tile_rows, int64_t count,
                               int C, const float* gamma, const float* beta, float* running_mean,
                               float* running_var, int64_t* num_batches_tracked, float* scale,
                               float* shift, float* save_mean, float* save_rstd, int training,
                               tdx_stream_t stream) {
  if (!gamma || !beta || !scale || !shift || C <= 0) return TDX_E_BADARG;
  if (training && (!stats_partial || tiles <= 0 || tile_rows <= 0 || count <= 0)) return TDX_E_BADARG;
  if (training && ((int64_t)tiles * tile_rows < count || (int64_t)(tiles - 1) * tile_rows >= count))
    return TDX_E_BADARG;
  if (!training && (!running_mean || !running_var)) return TDX_E_BADARG;
  bn_finalize_kernel<<<cdiv(C, 4), 256, 0, to_stream(stream)>>>(
      stats_partial, tiles, tile_rows, count, C, gamma, beta, running_mean, running_var,
      num_batches_tracked, scale, shift, save_mean, save_rstd, training);
  TDX_CHECK_LAUNCH();
  return 0;
}

// ---- synchronised BatchNorm (data-parallel training with statistics over the GLOBAL batch) -------
// The reference has no distributed code; under DistributedDataParallel its BatchNorm2d layers would
// normalise with rank-local statistics, which is this library's default.  SyncBN is
// the option that makes an N-rank step equal the single-process step on the concatenated batch:
//   forward   per-channel (sum x, sum x^2, count) in double from the convolution's tile partials ->
//             all-reduce(SUM) by the caller's callback -> scale / shift / running statistics from the
//             global moments;
//   backward  per-channel (sum gz, sum gz*xhat) in double -> all-reduce(SUM) -> the input gradient
//             with the global sums and the global count; dgamma / dbeta keep the LOCAL sums (the
//             gradient all-reduce averages them, like every other parameter gradient).
// mom: [sum x (C) | sum x^2 (C) | count (1)] doubles.  Sums of squares in double (53 bits) do not suffer
// the E[x^2]-E[x]^2 cancellation that made the fp32 path merge centred partials.
__global__ void __launch_bounds__(256)
bn_moments_kernel(const float* __restrict__ stats, int tiles, int tile_rows, int64_t count, int C,
                  double* __restrict__ mom) {
  __shared__ double red[2][4][256];
  const int c0 = blockIdx.x * 4, sl = threadIdx.x, cl = threadIdx.x & 3;
  double S[4] = {0, 0, 0, 0}, Q[4] = {0, 0, 0, 0};
  const double inv_full = 1.0 / (double)tile_rows;
  const double inv_last = 1.0 / (double)(count - (int64_t)(tiles - 1) * tile_rows);
  for (int t = sl; t < tiles; t += 256) {
    const float4 st = *reinterpret_cast<const float4*>(stats + ((size_t)t * 2 + 0) * C + c0);
    const float4 qt = *reinterpret_cast<const float4*>(stats + ((size_t)t * 2 + 1) * C + c0);
    const double inv = t == tiles - 1 ? inv_last : inv_full;
    const float sv[4] = {st.x, st.y, st.z, st.w}, qv[4] = {qt.x, qt.y, qt.z, qt.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      S[k] += (double)sv[k];
      Q[k] += (double)qv[k] + (double)sv[k] * (double)sv[k] * inv;   // sum x^2 of the tile = M2 + S^2 / n
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) { red[0][k][sl] = S[k]; red[1][k][sl] = Q[k]; }
  __syncthreads();
  for (int half = 128; half > 0; half >>= 1) {
    if (sl < half) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { red[0][k][sl] += red[0][k][sl + half]; red[1][k][sl] += red[1][k][sl + half]; }
    }
    __syncthreads();
  }
  if (threadIdx.x < 4 && c0 + cl < C) {
    mom[c0 + cl] = red[0][cl][0];
    mom[C + c0 + cl] = red[1][cl][0];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) mom[2 * C] = (double)count;
}

__global__ void bn_finalize_moments_kernel(const double* __restrict__ mom, int C, const float* __restrict__ gamma,
                                           const float* __restrict__ beta, float* __restrict__ rmean,
                                           float* __restrict__ rvar, int64_t* __restrict__ nbt,
                                           float* __restrict__ scale, float* __restrict__ shift,
                                           float* __restrict__ save_mean, float* __restrict__ save_rstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && nbt) nbt[0] += 1;
  if (c >= C) return;
  const double n = mom[2 * C], mean = mom[c] / n;
  double var = mom[C + c] / n - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)BN_EPS));
  const float sc = gamma[c] * rstd;
  scale[c] = sc;
  shift[c] = beta[c] - (float)mean * sc;
  if (save_mean) save_mean[c] = (float)mean;
  if (save_rstd) save_rstd[c] = rstd;
  if (rmean) {
    const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
    rmean[c] = (1.0f - BN_MOMENTUM) * rmean[c] + BN_MOMENTUM * (float)mean;
    rvar[c] = (1.0f - BN_MOMENTUM) * rvar[c] + BN_MOMENTUM * (float)unbiased;
  }
}

int tdx_bn_moments(const float* stats_partial, int tiles, int tile_rows, int64_t count, int C, double* mom,
                   hipStream_t st) {
  if (!stats_partial || !mom || tiles <= 0 || tile_rows <= 0 || count <= 0 || C <= 0 || C % 4) return TDX_E_BADARG;
  bn_moments_kernel<<<C / 4, 256, 0, st>>>(stats_partial, tiles, tile_rows, count, C, mom);
  TDX_CHECK_LAUNCH();
  return 0;
}

int tdx_bn_finalize_moments(const double* mom, int C, const float* gamma, const float* beta, float* running_mean,
                            float* running_var, int64_t* nbt, float* scale, float* shift, float* save_mean,
                            float* save_rstd, hipStream_t st) {
  if (!mom || !gamma || !beta || !scale || !shift || C <= 0) return TDX_E_BADARG;
  bn_finalize_moments_kernel<<<cdiv(C, 256), 256, 0, st>>>(mom, C, gamma, beta, running_mean, running_var, nbt, scale,
                                                           shift, save_mean, save_rstd);
  TDX_CHECK_LAUNCH();
  return 0;
}

// out = relu(y*scale + shift): the post-activation tensor, materialised only for the six units
// that feed another 3x3 convolution directly, so that convolution can fetch its tiles by LDS-DMA
// (no BN-on-load register path); everything else keeps applying scale/shift on load.
__global__ void __launch_bounds__(256)
bn_relu_apply_kernel(const float* __restrict__ y, float* __restrict__ out, int64_t n4, int C,
                     const float* __restrict__ scale, const float* __restrict__ shift) {
  const int c4n = C / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    const float4 sc = *reinterpret_cast<const float4*>(scale + c);
    const float4 sh = *reinterpret_cast<const float4*>(shift + c);
    const float4 v = reinterpret_cast<const float4*>(y)[i];
    float4 o;
    o.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f);
    o.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
    o.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f);
    o.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
    reinterpret_cast<float4*>(out)[i] = o;
  }
}

// the same on bf16 tensors (bf16 storage mode): eight channels per thread, fp32 arithmetic, one rounding on store -
// the values a BN+ReLU-on-load convolution would have staged
__global__ void __launch_bounds__(256)
bn_relu_apply16_kernel(const tdx_bf16* __restrict__ y, tdx_bf16* __restrict__ out, int64_t n8, int C,
                       const float* __restrict__ scale, const float* __restrict__ shift) {
  const int c8n = C / 8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c8n) * 8;
    const float4 a0 = ld4(y + i * 8), a1 = ld4(y + i * 8 + 4);
    const float4 s0 = *reinterpret_cast<const float4*>(scale + c), s1 = *reinterpret_cast<const float4*>(scale + c + 4);
    const float4 h0 = *reinterpret_cast<const float4*>(shift + c), h1 = *reinterpret_cast<const float4*>(shift + c + 4);
    float4 o0, o1;
    o0.x = fmaxf(fmaf(a0.x, s0.x, h0.x), 0.f); o0.y = fmaxf(fmaf(a0.y, s0.y, h0.y), 0.f);
    o0.z = fmaxf(fmaf(a0.z, s0.z, h0.z), 0.f); o0.w = fmaxf(fmaf(a0.w, s0.w, h0.w), 0.f);
    o1.x = fmaxf(fmaf(a1.x, s1.x, h1.x), 0.f); o1.y = fmaxf(fmaf(a1.y, s1.y, h1.y), 0.f);
    o1.z = fmaxf(fmaf(a1.z, s1.z, h1.z), 0.f); o1.w = fmaxf(fmaf(a1.w, s1.w, h1.w), 0.f);
    st4(out + i * 8, o0);
    st4(out + i * 8 + 4, o1);
  }
}

int tdx_bn_relu_apply(const float* y, float* out, int64_t rows, int C, const float* scale, const float* shift,
                      hipStream_t st, int io16) {
  if (!y || !out || !scale || !shift || rows <= 0 || C <= 0 || C % 4 || (io16 && C % 8)) return TDX_E_BADARG;
  const int64_t n4 = rows * C / 4;
  int grid = (int)(((io16 ? n4 / 2 : n4) + 255) / 256);
  if (grid > 8192) grid = 8192;
  if (io16)
    bn_relu_apply16_kernel<<<grid, 256, 0, st>>>(reinterpret_cast<const tdx_bf16*>(y), reinterpret_cast<tdx_bf16*>(out),
                                                 n4 / 2, C, scale, shift);
  else
    bn_relu_apply_kernel<<<grid, 256, 0, st>>>(y, out, n4, C, scale, shift);
  TDX_CHECK_LAUNCH();
  return 0;
}

extern "C" int tdx_bn_apply_relu_fwd(const float* y, float* out, int64_t rows, int C, const float* scale,
                                     const float* shift, tdx_stream_t stream) {
  return tdx_bn_relu_apply(y, out, rows, C, scale, shift, to_stream(stream), 0);
}

// ------------------------------------------------------------------ backward
// Rows per block of the reduction pass: sized so that the grid has ~2048 workgroups whatever the
// layer (the deep layers have few rows but many channels; a fixed 512 rows per block left
// them with 8..98 workgroups on 256 CUs and the pass ran at a fraction of HBM speed).
static int bwd_rows_per_block(int64_t rows, int C) {
  const int rgroups = 256 / (C / 4);
  int64_t rpb = (rows + 2047) / 2048;
  if (rpb < 2 * rgroups) rpb = 2 * rgroups;
  if (rpb > 512) rpb = 512;
  return (int)((rpb + rgroups - 1) / rgroups * rgroups);
}

// partial[blk][2][C]: sum gz, sum gz*xhat over the block's rows
template <typename T>   // element type of g and y in HBM (io16.h)
__global__ void __launch_bounds__(256)
bn_bwd_reduce_kernel(const T* __restrict__ g, const T* __restrict__ y, int64_t rows, int C,
                     const float* __restrict__ scale, const float* __restrict__ shift,
                     const float* __restrict__ mean, const float* __restrict__ rstd,
                     float* __restrict__ partial, int rpb) {
  extern __shared__ float red[];  // [rgroups][2][C]
  const int c4n = C / 4;
  const int col = threadIdx.x % c4n, rg = threadIdx.x / c4n, rgroups = 256 / c4n;
  const int c = col * 4;
  const float4 sc = *reinterpret_cast<const float4*>(scale + c);
  const float4 sh = *reinterpret_cast<const float4*>(shift + c);
  const float4 mu = *reinterpret_cast<const float4*>(mean + c);
  const float4 rs = *reinterpret_cast<const float4*>(rstd + c);
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  const int64_t r0 = (int64_t)blockIdx.x * rpb;
  const int64_t r1 = min(r0 + rpb, rows);
#pragma unroll 4
  for (int64_t r = r0 + rg; r < r1; r += rgroups) {
    const float4 gv = ld4(g + r * C + c);
    const float4 yv = ld4(y + r * C + c);
#define ACC(k)                                                   \
    {                                                            \
      const float gz = fmaf(yv.k, sc.k, sh.k) > 0.f ? gv.k : 0.f; \
      s1.k += gz;                                                \
      s2.k += gz * ((yv.k - mu.k) * rs.k);                       \
    }
    ACC(x) ACC(y) ACC(z) ACC(w)
#undef ACC
  }
  *reinterpret_cast<float4*>(red + (rg * 2 + 0) * C + c) = s1;
  *reinterpret_cast<float4*>(red + (rg * 2 + 1) * C + c) = s2;
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    float v = 0.f;
    for (int k = 0; k < rgroups; ++k) v += red[k * 2 * C + i];
    partial[(size_t)blockIdx.x * 2 * C + i] = v;
  }
}

// coef[5][C]: k1 = gamma*rstd (train) or scale (eval), m1 = S1/N, m2 = S2/N
__global__ void __launch_bounds__(256)
bn_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, double count, int C,
                       const float* __restrict__ gamma, const float* __restrict__ rstd,
                       const float* __restrict__ scale, float* __restrict__ dgamma,
                       float* __restrict__ dbeta, float* __restrict__ dbias,
                       float* __restrict__ coef, int training) {
  __shared__ double red[2][4][256];
  const int c0 = blockIdx.x * 4, sl = threadIdx.x, cl = threadIdx.x & 3;
  const int c = c0 + cl;
  double a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
#pragma unroll 4
  for (int t = sl; t < nblk; t += 256) {
    const float4 p1 = *reinterpret_cast<const float4*>(partial + ((size_t)t * 2 + 0) * C + c0);
    const float4 p2 = *reinterpret_cast<const float4*>(partial + ((size_t)t * 2 + 1) * C + c0);
    a1[0] += (double)p1.x; a1[1] += (double)p1.y; a1[2] += (double)p1.z; a1[3] += (double)p1.w;
    a2[0] += (double)p2.x; a2[1] += (double)p2.y; a2[2] += (double)p2.z; a2[3] += (double)p2.w;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) { red[0][k][sl] = a1[k]; red[1][k][sl] = a2[k]; }
  __syncthreads();
  for (int half = 128; half > 0; half >>= 1) {
    if (sl < half) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        red[0][k][sl] += red[0][k][sl + half];
        red[1][k][sl] += red[1][k][sl + half];
      }
    }
    __syncthreads();
  }
  if (threadIdx.x < 4 && c < C) {
    const double s1 = red[0][cl][0], s2 = red[1][cl][0];
    if (dgamma) dgamma[c] = (float)s2;
    if (dbeta) dbeta[c] = (float)s1;
    if (training) {
      coef[0 * C + c] = gamma[c] * rstd[c];
      coef[1 * C + c] = (float)(s1 / count);
      coef[2 * C + c] = (float)(s2 / count);
      // d(conv bias) = sum over rows of dy == 0 analytically: the batch mean removes it
      if (dbias) dbias[c] = 0.f;
    } else {
      coef[0 * C + c] = scale[c];
      coef[1 * C + c] = 0.f;
      coef[2 * C + c] = 0.f;
      if (dbias) dbias[c] = (float)(s1 * (double)scale[c]);
    }
  }
}

// SyncBN backward, step 2 of 3: the block partials -> LOCAL per-channel sums; dgamma / dbeta take them,
// mom = [sum gz (C) | sum gz*xhat (C) | rows (1)] in double goes to the all-reduce
__global__ void __launch_bounds__(256)
bn_bwd_sums_kernel(const float* __restrict__ partial, int nblk, double count, int C,
                   float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dbias,
                   double* __restrict__ mom) {
  __shared__ double red[2][4][256];
  const int c0 = blockIdx.x * 4, sl = threadIdx.x, cl = threadIdx.x & 3;
  const int c = c0 + cl;
  double a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
  for (int t = sl; t < nblk; t += 256) {
    const float4 p1 = *reinterpret_cast<const float4*>(partial + ((size_t)t * 2 + 0) * C + c0);
    const float4 p2 = *reinterpret_cast<const float4*>(partial + ((size_t)t * 2 + 1) * C + c0);
    a1[0] += (double)p1.x; a1[1] += (double)p1.y; a1[2] += (double)p1.z; a1[3] += (double)p1.w;
    a2[0] += (double)p2.x; a2[1] += (double)p2.y; a2[2] += (double)p2.z; a2[3] += (double)p2.w;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) { red[0][k][sl] = a1[k]; red[1][k][sl] = a2[k]; }
  __syncthreads();
  for (int half = 128; half > 0; half >>= 1) {
    if (sl < half) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { red[0][k][sl] += red[0][k][sl + half]; red[1][k][sl] += red[1][k][sl + half]; }
    }
    __syncthreads();
  }
  if (threadIdx.x < 4 && c < C) {
    const double s1 = red[0][cl][0], s2 = red[1][cl][0];
    if (dgamma) dgamma[c] = (float)s2;
    if (dbeta) dbeta[c] = (float)s1;
    if (dbias) dbias[c] = 0.f;   // train mode: the batch mean removes the bias gradient
    mom[c] = s1;
    mom[C + c] = s2;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) mom[2 * C] = count;
}

// step 3: coef = {gamma*rstd, S1/N, S2/N} from the all-reduced sums and the global count
__global__ void bn_bwd_coef_moments_kernel(const double* __restrict__ mom, int C, const float* __restrict__ gamma,
                                           const float* __restrict__ rstd, float* __restrict__ coef) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double n = mom[2 * C];
  coef[0 * C + c] = gamma[c] * rstd[c];
  coef[1 * C + c] = (float)(mom[c] / n);
  coef[2 * C + c] = (float)(mom[C + c] / n);
}

template <typename T>
__global__ void __launch_bounds__(256)
bn_bwd_apply_kernel(T* __restrict__ g, const T* __restrict__ y, int64_t n4, int C,
                    const float* __restrict__ scale, const float* __restrict__ shift,
                    const float* __restrict__ mean, const float* __restrict__ rstd,
                    const float* __restrict__ coef) {
  const int c4n = C / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    const float4 sc = *reinterpret_cast<const float4*>(scale + c);
    const float4 sh = *reinterpret_cast<const float4*>(shift + c);
    const float4 mu = *reinterpret_cast<const float4*>(mean + c);
    const float4 rs = *reinterpret_cast<const float4*>(rstd + c);
    const float4 k1 = *reinterpret_cast<const float4*>(coef + c);
    const float4 m1 = *reinterpret_cast<const float4*>(coef + C + c);
    const float4 m2 = *reinterpret_cast<const float4*>(coef + 2 * C + c);
    float4 gv = ld4(g + i * 4);
    const float4 yv = ld4(y + i * 4);
#define APPLY(k)                                                     \
    {                                                                \
      const float gz = fmaf(yv.k, sc.k, sh.k) > 0.f ? gv.k : 0.f;    \
      const float xh = (yv.k - mu.k) * rs.k;                         \
      gv.k = k1.k * (gz - m1.k - xh * m2.k);                         \
    }
    APPLY(x) APPLY(y) APPLY(z) APPLY(w)
#undef APPLY
    st4(g + i * 4, gv);
  }
}

extern "C" size_t tdx_bn_relu_bwd_scratch_floats(int64_t rows, int C) {
  if (rows <= 0 || C <= 0 || C % 4 || C > 1024 || (256 % (C / 4)) != 0) return 0;
  // room for whichever kernel writes the partial rows: the reduction pass below, a convolution epilogue (one row per
  // tile of >= 64 pixels) or a spatial producer (at most TDX_BNBWD_MAX_PRODUCER_BLOCKS workgroups)
  size_t nblk = (size_t)cdiv(rows, bwd_rows_per_block(rows, C));
  nblk = std::max(nblk, (size_t)cdiv(rows, 64));
  nblk = std::max(nblk, (size_t)TDX_BNBWD_MAX_PRODUCER_BLOCKS);
  return nblk * 2 * C + 3 * (size_t)C;
}

extern "C" int tdx_bn_relu_bwd(float* g, const float* y, int64_t rows, int C, const float* scale,
                               const float* shift, const float* save_mean, const float* save_rstd,
                               const float* gamma, float* dgamma, float* dbeta, float* dbias,
                               float* scratch, int training, tdx_stream_t stream) {
  return tdx_bn_relu_bwd_sync(g, y, rows, C, scale, shift, save_mean, save_rstd, gamma, dgamma, dbeta, dbias, scratch,
                              training, nullptr, nullptr, nullptr, stream, 0);
}

// The same with the per-channel sums all-reduced across ranks between the reduction and the apply
// pass (sync != NULL; train mode only): `mom` is the caller's device buffer of 2*C + 1 doubles.
int tdx_bn_relu_bwd_sync(float* g, const float* y, int64_t rows, int C, const float* scale,
                         const float* shift, const float* save_mean, const float* save_rstd,
                         const float* gamma, float* dgamma, float* dbeta, float* dbias,
                         float* scratch, int training, tdx_allreduce_fn sync, void* sync_user, double* mom,
                         tdx_stream_t stream, int io16) {   // io16: g and y hold bf16 (io16.h)
  if (!g || !y || !scale || !shift || !save_mean || !save_rstd || !gamma || !scratch || rows <= 0)
    return TDX_E_BADARG;
  if (sync && (!mom || !training)) return TDX_E_BADARG;
  if (C % 4 || C > 1024 || (256 % (C / 4)) != 0) return TDX_E_SHAPE;
  hipStream_t st = to_stream(stream);
  const int rpb = bwd_rows_per_block(rows, C);
  const int nblk = cdiv(rows, rpb);
  float* partial = scratch;
  float* coef = scratch + (size_t)nblk * 2 * C;
  const int rgroups = 256 / (C / 4);
  TDX_IO_DISPATCH(io16, T, bn_bwd_reduce_kernel<T><<<nblk, 256, (size_t)rgroups * 2 * C * sizeof(float), st>>>(
      reinterpret_cast<const T*>(g), reinterpret_cast<const T*>(y), rows, C, scale, shift, save_mean, save_rstd, partial, rpb));
  TDX_CHECK_LAUNCH();
  return tdx_bn_relu_bwd_tail(g, y, rows, C, scale, shift, save_mean, save_rstd, gamma, dgamma, dbeta, dbias, partial,
                              nblk, coef, training, sync, sync_user, mom, stream, io16);
}

// Finalize + apply from partial sums [nblk][2][C] that are already in memory: written by the reduction kernel
// above, or by the kernel that produced g (conv_epilogue EPI_BNBWD, bilinear_bwd_rows / maxpool_bwd with the BN
// operands), in which case no reduction pass runs at all.
int tdx_bn_relu_bwd_tail(float* g, const float* y, int64_t rows, int C, const float* scale, const float* shift,
                         const float* save_mean, const float* save_rstd, const float* gamma, float* dgamma,
                         float* dbeta, float* dbias, const float* partial, int nblk, float* coef, int training,
                         tdx_allreduce_fn sync, void* sync_user, double* mom, tdx_stream_t stream, int io16) {
  if (!g || !y || !partial || !coef || nblk <= 0 || rows <= 0) return TDX_E_BADARG;
  if (sync && (!mom || !training)) return TDX_E_BADARG;
  if (C % 4 || C > 1024) return TDX_E_SHAPE;
  hipStream_t st = to_stream(stream);
  if (sync) {
    bn_bwd_sums_kernel<<<cdiv(C, 4), 256, 0, st>>>(partial, nblk, (double)rows, C, dgamma, dbeta, dbias, mom);
    TDX_CHECK_LAUNCH();
    const int rc = sync(sync_user, mom, 2 * C + 1, stream);
    if (rc) return rc;
    bn_bwd_coef_moments_kernel<<<cdiv(C, 256), 256, 0, st>>>(mom, C, gamma, save_rstd, coef);
    TDX_CHECK_LAUNCH();
  } else {
    bn_bwd_finalize_kernel<<<cdiv(C, 4), 256, 0, st>>>(partial, nblk, (double)rows, C, gamma,
                                                       save_rstd, scale, dgamma, dbeta, dbias, coef,
                                                       training);
    TDX_CHECK_LAUNCH();
  }
  const int64_t n4 = rows * C / 4;
  int grid = (int)((n4 + 255) / 256);
  if (grid > 4096) grid = 4096;
  TDX_IO_DISPATCH(io16, T, bn_bwd_apply_kernel<T><<<grid, 256, 0, st>>>(reinterpret_cast<T*>(g), reinterpret_cast<const T*>(y), n4, C,
                                                                        scale, shift, save_mean, save_rstd, coef));
  TDX_CHECK_LAUNCH();
  return 0;
}

// HBM-bound spatial glue of the UNet, channels-last, with BatchNorm+ReLU applied
// on load so that normalised activations are never materialised:
//   MaxPool2d(2, ceil_mode=True)                 diffusion.py:101, 120-124
//   bilinear resize, align_corners=True          diffusion.py:102, 135-159
// and their backward passes (the boundary convolutions initial_conv / final_conv live in edge_conv.hip).
#include "internal.h"
#include "io16.h"

__device__ static inline float4 bnrelu4(float4 v, const float4& sc, const float4& sh) {
  v.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f);
  v.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
  v.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f);
  v.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
  return v;
}

// BatchNorm-backward partial sums emitted by the kernel that PRODUCES dL/d(activation) of a unit (the max-pool
// backward and the resize adjoint below; the input-gradient convolutions do the same in conv_epilogue): per
// workgroup and channel, sum gz and sum gz * xhat over the elements the workgroup wrote, gz = the gradient where
// relu(bn(y)) is active, xhat = (y - mean) * rstd - what bn_bwd_reduce_kernel would compute in a pass of its own
// over both tensors.  Every thread of these kernels keeps ONE channel quad for its whole life (256 % (C/4) == 0 and
// its stride is a multiple of 256), so the sums live in registers and meet in LDS once, in a fixed order.
struct BnBwdOps {
  const void* y;        // pre-BN output of the unit, same layout and element type as the gradient being written
  const float* scale; const float* shift; const float* mean; const float* rstd;
  float* partial;       // [gridDim.x][2][C]
};
__device__ static inline void bnbwd_acc(const float4& g, const float4& y, const float4& sc, const float4& sh,
                                        const float4& mu, const float4& rs, float4& s1, float4& s2) {
#define ACC(k)                                                  \
  {                                                             \
    const float gz = fmaf(y.k, sc.k, sh.k) > 0.f ? g.k : 0.f;   \
    s1.k += gz;                                                 \
    s2.k += gz * ((y.k - mu.k) * rs.k);                         \
  }
  ACC(x) ACC(y) ACC(z) ACC(w)
#undef ACC
}
// red: 2 * 256 * 4 floats of LDS; thread t owns channels (t % c4n) * 4 .. + 3
__device__ static inline void bnbwd_flush(float* red, const float4& s1, const float4& s2, int C, float* partial) {
  const int c4n = C / 4, col = threadIdx.x % c4n, rg = threadIdx.x / c4n, rgroups = 256 / c4n;
  __syncthreads();
  *reinterpret_cast<float4*>(red + (rg * 2 + 0) * C + col * 4) = s1;
  *reinterpret_cast<float4*>(red + (rg * 2 + 1) * C + col * 4) = s2;
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    float v = 0.f;
    for (int k = 0; k < rgroups; ++k) v += red[k * 2 * C + i];
    partial[(size_t)blockIdx.x * 2 * C + i] = v;
  }
}

static inline int ew_grid(int64_t n, int block = 256, int cap = 4096) {
  int64_t g = (n + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// ------------------------------------------------------------------- max-pool
template <bool BN, typename T>
__global__ void maxpool_fwd_kernel(const T* __restrict__ y, const float* __restrict__ scale,
                                   const float* __restrict__ shift, T* __restrict__ out, int B,
                                   int H, int W, int C) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, c4n = C / 4;
  const int64_t n = (int64_t)B * Ho * Wo * c4n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    int64_t p = i / c4n;
    const int ow = (int)(p % Wo); p /= Wo;
    const int oh = (int)(p % Ho);
    const int b = (int)(p / Ho);
    float4 sc, sh;
    if (BN) {
      sc = *reinterpret_cast<const float4*>(scale + c);
      sh = *reinterpret_cast<const float4*>(shift + c);
    }
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
      for (int dw = 0; dw < 2; ++dw) {
        const int ih = 2 * oh + dh, iw = 2 * ow + dw;
        if (ih < H && iw < W) {
          float4 v = ld4(y + (((int64_t)b * H + ih) * W + iw) * C + c);
          if (BN) v = bnrelu4(v, sc, sh);
          m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y);
          m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
      }
    st4(out + i * 4, m);
  }
}

// io16: y and out hold bf16 (io16.h); the C-ABI entry is the fp32 form
int tdx_maxpool2_ceil_fwd_t(const void* y, const float* scale, const float* shift, void* out, int B, int H, int W,
                            int C, int io16, tdx_stream_t stream) {
  if (!y || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0) return TDX_E_BADARG;
  if (C % 4) return TDX_E_SHAPE;
  const int64_t n = (int64_t)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
  TDX_IO_DISPATCH(io16, T,
    if (scale) maxpool_fwd_kernel<true, T><<<ew_grid(n), 256, 0, to_stream(stream)>>>((const T*)y, scale, shift, (T*)out, B, H, W, C);
    else maxpool_fwd_kernel<false, T><<<ew_grid(n), 256, 0, to_stream(stream)>>>((const T*)y, scale, shift, (T*)out, B, H, W, C));
  TDX_CHECK_LAUNCH();
  return 0;
}
extern "C" int tdx_maxpool2_ceil_fwd(const float* y, const float* scale, const float* shift,
                                     float* out, int B, int H, int W, int C, tdx_stream_t stream) {
  return tdx_maxpool2_ceil_fwd_t(y, scale, shift, out, B, H, W, C, 0, stream);
}

// One thread owns one 2x2 window (windows do not overlap): route the pooled
// gradient to the FIRST maximum in scan order (ATen: `val > maxval`), add the
// skip-path gradient, and write all (up to) four input positions.
template <bool BN, bool BNP, typename T>
__global__ void maxpool_bwd_kernel(const T* __restrict__ y, const float* __restrict__ scale,
                                   const float* __restrict__ shift, const T* __restrict__ g_out,
                                   const T* __restrict__ skip, T* __restrict__ g_in, int B,
                                   int H, int W, int C, BnBwdOps bw) {
  __shared__ __attribute__((aligned(16))) float red[BNP ? 2048 : 4];
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, c4n = C / 4;
  const int64_t n = (int64_t)B * Ho * Wo * c4n;
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1, mu = s1, rs = s1;
  if (BNP) {
    const int cq = (threadIdx.x % c4n) * 4;
    mu = *reinterpret_cast<const float4*>(bw.mean + cq);
    rs = *reinterpret_cast<const float4*>(bw.rstd + cq);
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    int64_t p = i / c4n;
    const int ow = (int)(p % Wo); p /= Wo;
    const int oh = (int)(p % Ho);
    const int b = (int)(p / Ho);
    float4 sc, sh;
    if (BN) {
      sc = *reinterpret_cast<const float4*>(scale + c);
      sh = *reinterpret_cast<const float4*>(shift + c);
    }
    const float4 go = ld4(g_out + i * 4);
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int arg[4] = {0, 0, 0, 0};
    float4 yraw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int ih = 2 * oh + (k >> 1), iw = 2 * ow + (k & 1);
      if (ih < H && iw < W) {
        float4 v = ld4(y + (((int64_t)b * H + ih) * W + iw) * C + c);
        yraw[k] = v;
        if (BN) v = bnrelu4(v, sc, sh);
        if (v.x > m[0]) { m[0] = v.x; arg[0] = k; }
        if (v.y > m[1]) { m[1] = v.y; arg[1] = k; }
        if (v.z > m[2]) { m[2] = v.z; arg[2] = k; }
        if (v.w > m[3]) { m[3] = v.w; arg[3] = k; }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int ih = 2 * oh + (k >> 1), iw = 2 * ow + (k & 1);
      if (ih < H && iw < W) {
        const int64_t off = (((int64_t)b * H + ih) * W + iw) * C + c;
        float4 r = skip ? ld4(skip + off) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (arg[0] == k) r.x += go.x;
        if (arg[1] == k) r.y += go.y;
        if (arg[2] == k) r.z += go.z;
        if (arg[3] == k) r.w += go.w;
        st4(g_in + off, r);
        if (BNP) bnbwd_acc(r, yraw[k], sc, sh, mu, rs, s1, s2);
      }
    }
  }
  if (BNP) bnbwd_flush(red, s1, s2, C, bw.partial);
}

static int maxpool_bwd_t(const void* y, const float* scale, const float* shift, const void* g_out,
                         const void* skip_grad, void* g_in, int B, int H, int W, int C, int io16, tdx_stream_t stream) {
  if (!y || !g_out || !g_in || B <= 0 || H <= 0 || W <= 0 || C <= 0) return TDX_E_BADARG;
  if (C % 4) return TDX_E_SHAPE;
  const int64_t n = (int64_t)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
  TDX_IO_DISPATCH(io16, T,
    if (scale) maxpool_bwd_kernel<true, false, T><<<ew_grid(n), 256, 0, to_stream(stream)>>>((const T*)y, scale, shift, (const T*)g_out, (const T*)skip_grad, (T*)g_in, B, H, W, C, BnBwdOps{});
    else maxpool_bwd_kernel<false, false, T><<<ew_grid(n), 256, 0, to_stream(stream)>>>((const T*)y, scale, shift, (const T*)g_out, (const T*)skip_grad, (T*)g_in, B, H, W, C, BnBwdOps{}));
  TDX_CHECK_LAUNCH();
  return 0;
}
extern "C" int tdx_maxpool2_ceil_bwd(const float* y, const float* scale, const float* shift,
                                     const float* g_out, const float* skip_grad, float* g_in, int B,
                                     int H, int W, int C, tdx_stream_t stream) {
  return maxpool_bwd_t(y, scale, shift, g_out, skip_grad, g_in, B, H, W, C, 0, stream);
}

int tdx_maxpool2_ceil_bwd_bn(const void* y, const float* scale, const float* shift, const void* g_out,
                             const void* skip_grad, void* g_in, int B, int H, int W, int C, const float* bn_mean,
                             const float* bn_rstd, float* partial, int* nblk, tdx_stream_t stream, int io16) {
  *nblk = 0;
  if (!(g_tdx_bnbwd_fused & 4) || !scale || !partial || C % 4 || C > 1024 || 256 % (C / 4))
    return maxpool_bwd_t(y, scale, shift, g_out, skip_grad, g_in, B, H, W, C, io16, stream);
  if (!y || !g_out || !g_in || !bn_mean || !bn_rstd || B <= 0 || H <= 0 || W <= 0) return TDX_E_BADARG;
  const int64_t n = (int64_t)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
  const int grid = ew_grid(n, 256, TDX_BNBWD_MAX_PRODUCER_BLOCKS);
  TDX_IO_DISPATCH(io16, T, maxpool_bwd_kernel<true, true, T><<<grid, 256, 0, to_stream(stream)>>>(
      (const T*)y, scale, shift, (const T*)g_out, (const T*)skip_grad, (T*)g_in, B, H, W, C,
      BnBwdOps{y, scale, shift, bn_mean, bn_rstd, partial}));
  TDX_CHECK_LAUNCH();
  *nblk = grid;
  return 0;
}

// ------------------------------------------------------------------- bilinear
// align_corners=True source coordinate, fp32 exactly like ATen:
//   scale = (in-1)/(out-1);  src = scale*dst;  i0 = (int)src;  i1 = i0 + (i0 < in-1);  l1 = src - i0
struct AxisTap {
  int i0, i1;
  float l0, l1;
};
__device__ static inline AxisTap axis_tap(int dst, float scale, int n_in) {
  AxisTap t;
  const float src = scale * (float)dst;
  t.i0 = min((int)src, n_in - 1);
  t.i1 = t.i0 + (t.i0 < n_in - 1 ? 1 : 0);
  t.l1 = src - (float)t.i0;
  t.l0 = 1.0f - t.l1;
  return t;
}

template <bool BN, typename T>
__global__ void bilinear_fwd_kernel(const T* __restrict__ in, const float* __restrict__ scale,
                                    const float* __restrict__ shift,
                                    const float* __restrict__ addend, T* __restrict__ out, int B,
                                    int Hi, int Wi, int Ho, int Wo, int C, int ocs, int ocoff,
                                    float sch, float scw) {
  const int c4n = C / 4;
  const int64_t n = (int64_t)B * Ho * Wo * c4n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    int64_t p = i / c4n;
    const int ow = (int)(p % Wo); p /= Wo;
    const int oh = (int)(p % Ho);
    const int b = (int)(p / Ho);
    const AxisTap th = axis_tap(oh, sch, Hi), tw = axis_tap(ow, scw, Wi);
    float4 sc, sh, ad = make_float4(0.f, 0.f, 0.f, 0.f);
    if (BN) {
      sc = *reinterpret_cast<const float4*>(scale + c);
      sh = *reinterpret_cast<const float4*>(shift + c);
    }
    if (addend) ad = *reinterpret_cast<const float4*>(addend + (int64_t)b * C + c);
    const T* base = in + (int64_t)b * Hi * Wi * C + c;
    auto ld = [&](int h, int w) {
      float4 v = ld4(base + ((int64_t)h * Wi + w) * C);
      if (BN) v = bnrelu4(v, sc, sh);
      v.x += ad.x; v.y += ad.y; v.z += ad.z; v.w += ad.w;
      return v;
    };
    const float4 v00 = ld(th.i0, tw.i0), v01 = ld(th.i0, tw.i1);
    const float4 v10 = ld(th.i1, tw.i0), v11 = ld(th.i1, tw.i1);
    float4 o;
    // width first, then height (ATen's separable evaluation order)
#define LERP(k)                                            \
    {                                                      \
      const float top = tw.l0 * v00.k + tw.l1 * v01.k;     \
      const float bot = tw.l0 * v10.k + tw.l1 * v11.k;     \
      o.k = th.l0 * top + th.l1 * bot;                     \
    }
    LERP(x) LERP(y) LERP(z) LERP(w)
#undef LERP
    st4(out + (((int64_t)b * Ho + oh) * Wo + ow) * ocs + ocoff + c, o);
  }
}

// Both halves of a decoder input in ONE launch (inference / sampling, where every launch is ~5 us
// of a ~550 us step): channels [0, CA) = resize(A), channels [CA, CA+CB) = resize(Bsrc + addend).
// A source may be a DEFERRED split-K result (sampling: TdxSplitDefer, internal.h): `in` then points at `splits` raw
// partial slabs of `slab` floats and the loader forms relu((bias + sum_s partial_s) * scale + shift) itself, in
// the fixed order 0..splits-1 - the separate reduction launch of that convolution (and the tensor it would have
// written, which nothing else reads in INFER mode) disappears.  An up-sampled source is touched by ~4 outputs
// per element, so the sums are formed ~4 times over; the partials of a 16-sample step are L2-resident.
struct ResizeSrc {
  const void* in; const float* addend; int Hi, Wi, C; float sch, scw;   // in: T elements (fp32 partials when splits > 0)
  int splits; size_t slab; const float* bias; const float* scale; const float* shift;
};
template <typename T>
__global__ void bilinear_pair_fwd_kernel(ResizeSrc A, ResizeSrc Bs, T* __restrict__ out, int B, int Ho,
                                         int Wo) {
  const int ocs = A.C + Bs.C, c4n = ocs / 4;
  const int64_t n = (int64_t)B * Ho * Wo * c4n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int oc = (int)(i % c4n) * 4;
    int64_t p = i / c4n;
    const int ow = (int)(p % Wo); p /= Wo;
    const int oh = (int)(p % Ho);
    const int b = (int)(p / Ho);
    const bool first = oc < A.C;
    const ResizeSrc& S = first ? A : Bs;
    const int c = first ? oc : oc - A.C;
    const AxisTap th = axis_tap(oh, S.sch, S.Hi), tw = axis_tap(ow, S.scw, S.Wi);
    float4 ad = make_float4(0.f, 0.f, 0.f, 0.f);
    if (S.addend) ad = *reinterpret_cast<const float4*>(S.addend + (int64_t)b * S.C + c);
    const int64_t boff = (int64_t)b * S.Hi * S.Wi * S.C + c;
    float4 dbv = make_float4(0.f, 0.f, 0.f, 0.f), dsc = dbv, dsh = dbv;
    if (S.splits > 0) {
      if (S.bias) dbv = *reinterpret_cast<const float4*>(S.bias + c);
      dsc = *reinterpret_cast<const float4*>(S.scale + c);
      dsh = *reinterpret_cast<const float4*>(S.shift + c);
    }
    auto ld = [&](int h, int w) {
      const int64_t qo = boff + ((int64_t)h * S.Wi + w) * S.C;
      float4 v;
      if (S.splits > 0) {   // deferred split-K source (fp32 partials): the epilogue of splitk_reduce_kernel<true>, on load
        const float* q = static_cast<const float*>(S.in) + qo;
        v = dbv;
        for (int sp = 0; sp < S.splits; ++sp) {
          const float4 t = *reinterpret_cast<const float4*>(q + (size_t)sp * S.slab);
          v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        v = bnrelu4(v, dsc, dsh);
      } else {
        v = ld4(static_cast<const T*>(S.in) + qo);
      }
      v.x += ad.x; v.y += ad.y; v.z += ad.z; v.w += ad.w;
      return v;
    };
    const float4 v00 = ld(th.i0, tw.i0), v01 = ld(th.i0, tw.i1);
    const float4 v10 = ld(th.i1, tw.i0), v11 = ld(th.i1, tw.i1);
    float4 o;
#define LERP(k)                                            \
    {                                                      \
      const float top = tw.l0 * v00.k + tw.l1 * v01.k;     \
      const float bot = tw.l0 * v10.k + tw.l1 * v11.k;     \
      o.k = th.l0 * top + th.l1 * bot;                     \
    }
    LERP(x) LERP(y) LERP(z) LERP(w)
#undef LERP
    st4(out + (((int64_t)b * Ho + oh) * Wo + ow) * ocs + oc, o);
  }
}

static inline float ac_scale(int n_in, int n_out) {
  return n_out > 1 ? (float)(n_in - 1) / (float)(n_out - 1) : 0.0f;
}

int tdx_bilinear_ac_fwd_t(const void* in, const float* scale, const float* shift, const float* addend, void* out, int B,
                          int Hi, int Wi, int Ho, int Wo, int C, int out_cstride, int out_coff, int io16,
                          tdx_stream_t stream) {
  if (!in || !out || B <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || C <= 0) return TDX_E_BADARG;
  if (C % 4 || out_cstride % 4 || out_coff % 4 || out_coff + C > out_cstride) return TDX_E_SHAPE;
  const int64_t n = (int64_t)B * Ho * Wo * (C / 4);
  const float sch = ac_scale(Hi, Ho), scw = ac_scale(Wi, Wo);
  TDX_IO_DISPATCH(io16, T,
    if (scale)
      bilinear_fwd_kernel<true, T><<<ew_grid(n), 256, 0, to_stream(stream)>>>(
          (const T*)in, scale, shift, addend, (T*)out, B, Hi, Wi, Ho, Wo, C, out_cstride, out_coff, sch, scw);
    else
      bilinear_fwd_kernel<false, T><<<ew_grid(n), 256, 0, to_stream(stream)>>>(
          (const T*)in, scale, shift, addend, (T*)out, B, Hi, Wi, Ho, Wo, C, out_cstride, out_coff, sch, scw));
  TDX_CHECK_LAUNCH();
  return 0;
}
extern "C" int tdx_bilinear_ac_fwd(const float* in, const float* scale, const float* shift,
                                   const float* addend, float* out, int B, int Hi, int Wi, int Ho,
                                   int Wo, int C, int out_cstride, int out_coff,
                                   tdx_stream_t stream) {
  return tdx_bilinear_ac_fwd_t(in, scale, shift, addend, out, B, Hi, Wi, Ho, Wo, C, out_cstride, out_coff, 0, stream);
}

int tdx_bilinear_pair_fwd(const float* a, int Ha, int Wa, int Ca, const float* b, const float* b_addend, int Hb,
                          int Wb, int Cb, float* out, int B, int Ho, int Wo, hipStream_t st) {
  return tdx_bilinear_pair_fwd_ex(a, nullptr, Ha, Wa, Ca, b, b_addend, Hb, Wb, Cb, out, B, Ho, Wo, st);
}

// the same with source A optionally a deferred split-K result (a_defer != null: `a` is ignored) and source B optional
// (Cb == 0: a plain resize of A into `out`)
int tdx_bilinear_pair_fwd_ex(const void* a, const TdxSplitDefer* a_defer, int Ha, int Wa, int Ca, const void* b,
                             const float* b_addend, int Hb, int Wb, int Cb, void* out, int B, int Ho, int Wo,
                             hipStream_t st, int io16) {
  const bool deferred = a_defer && a_defer->splits > 0;
  if ((!a && !deferred) || (!b && Cb) || !out || Ca % 4 || Cb % 4 || Ca <= 0 || Cb < 0) return TDX_E_BADARG;
  if (deferred && (!a_defer->partial || !a_defer->scale || !a_defer->shift)) return TDX_E_BADARG;
  ResizeSrc A{deferred ? static_cast<const void*>(a_defer->partial) : a, nullptr, Ha, Wa, Ca, ac_scale(Ha, Ho), ac_scale(Wa, Wo),
              deferred ? a_defer->splits : 0, deferred ? a_defer->slab : 0, deferred ? a_defer->bias : nullptr,
              deferred ? a_defer->scale : nullptr, deferred ? a_defer->shift : nullptr};
  ResizeSrc Bs{b, b_addend, Cb ? Hb : 1, Cb ? Wb : 1, Cb, Cb ? ac_scale(Hb, Ho) : 0.f, Cb ? ac_scale(Wb, Wo) : 0.f,
               0, 0, nullptr, nullptr, nullptr};
  const int64_t n = (int64_t)B * Ho * Wo * ((Ca + Cb) / 4);
  TDX_IO_DISPATCH(io16, T, bilinear_pair_fwd_kernel<T><<<ew_grid(n), 256, 0, st>>>(A, Bs, (T*)out, B, Ho, Wo));
  TDX_CHECK_LAUNCH();
  return 0;
}

// Adjoint (gather form): every input position sums the output positions whose
// two taps touch it.  Candidate outputs of input index i lie in
// [ (i-1)/scale , (i+1)/scale ]; each candidate's taps are recomputed with the
// forward's exact arithmetic, so forward and backward use identical weights.
#define BIL_MAXC 8
__device__ static inline void axis_adjoint(int i, float scale, int n_in, int n_out, int& lo,
                                           float (&w)[BIL_MAXC]) {
  int hi;
  if (scale > 0.f) {
    lo = max(0, (int)floorf((float)(i - 1) / scale) - 1);
    hi = min(n_out - 1, (int)ceilf((float)(i + 1) / scale) + 1);
  } else {
    lo = 0;
    hi = n_out - 1;
  }
  if (hi - lo + 1 > BIL_MAXC) hi = lo + BIL_MAXC - 1;  // host guarantees this never truncates
#pragma unroll
  for (int k = 0; k < BIL_MAXC; ++k) {
    const int o = lo + k;
    float wk = 0.f;
    if (o <= hi) {
      const AxisTap t = axis_tap(o, scale, n_in);
      if (t.i0 == i) wk += t.l0;
      if (t.i1 == i) wk += t.l1;
    }
    w[k] = wk;
  }
}

template <typename T>
__global__ void bilinear_bwd_kernel(const T* __restrict__ g_out, T* __restrict__ g_in, int B,
                                    int Hi, int Wi, int Ho, int Wo, int C, int gcs, int gcoff,
                                    float sch, float scw) {
  const int c4n = C / 4;
  const int64_t n = (int64_t)B * Hi * Wi * c4n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    int64_t p = i / c4n;
    const int iw = (int)(p % Wi); p /= Wi;
    const int ih = (int)(p % Hi);
    const int b = (int)(p / Hi);
    int hlo, wlo;
    float wh[BIL_MAXC], ww[BIL_MAXC];
    axis_adjoint(ih, sch, Hi, Ho, hlo, wh);
    axis_adjoint(iw, scw, Wi, Wo, wlo, ww);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int a = 0; a < BIL_MAXC; ++a) {
      if (wh[a] == 0.f) continue;
#pragma unroll
      for (int d = 0; d < BIL_MAXC; ++d) {
        if (ww[d] == 0.f) continue;
        const float4 g = ld4(g_out + (((int64_t)b * Ho + hlo + a) * Wo + wlo + d) * gcs + gcoff + c);
        const float w = wh[a] * ww[d];
        acc.x = fmaf(w, g.x, acc.x); acc.y = fmaf(w, g.y, acc.y);
        acc.z = fmaf(w, g.z, acc.z); acc.w = fmaf(w, g.w, acc.w);
      }
    }
    st4(g_in + i * 4, acc);
  }
}

// Row form of the same adjoint (what the networks run): a workgroup owns whole rows (b, ih) of
// g_in.  The contributor window of the row (H axis, one thread) and of every column (W axis, one
// thread per iw) is computed ONCE per workgroup into LDS with the arithmetic above, so the inner
// loop is loads and fmas only - the generic kernel spends ~5x the HBM time of its traffic on
// re-deriving 16 axis taps and testing 64 weight products per element.  Same contributors, same
// weights, same summation order (rows ascending, columns ascending) => bit-identical results.
#define BIL_ROW_MAXW 64
#define BIL_BATCH 5
// one axis' contributor window of input index i, compacted to its non-zero span: weights w[0..n), first output lo
__device__ static inline void axis_window(int i, float scale, int n_in, int n_out, float* w_out, int& lo_out,
                                          int& n_out_w) {
  int lo;
  float w[BIL_MAXC];
  axis_adjoint(i, scale, n_in, n_out, lo, w);
  int first = BIL_MAXC, last = -1;
#pragma unroll
  for (int k = 0; k < BIL_MAXC; ++k)
    if (w[k] != 0.f) { first = min(first, k); last = k; }
#pragma unroll
  for (int k = 0; k < BIL_MAXC; ++k) w_out[k] = 0.f;
#pragma unroll
  for (int k = 0; k < BIL_MAXC; ++k)
    if (k >= first && k <= last) w_out[k - first] = w[k];
  lo_out = lo + (last >= 0 ? first : 0);
  n_out_w = last >= 0 ? last - first + 1 : 0;
}

template <bool BNP, typename T>
__global__ void __launch_bounds__(256)
bilinear_bwd_rows_kernel(const T* __restrict__ g_out, T* __restrict__ g_in, int rows,
                         int Hi, int Wi, int Ho, int Wo, int C, int gcs, int gcoff, float sch,
                         float scw, BnBwdOps bw) {
  __shared__ float s_ww[BIL_ROW_MAXW][BIL_MAXC], s_wh[BIL_ROW_MAXW][BIL_MAXC];
  __shared__ int s_wlo[BIL_ROW_MAXW], s_wn[BIL_ROW_MAXW], s_hlo[BIL_ROW_MAXW], s_hn[BIL_ROW_MAXW];
  __shared__ __attribute__((aligned(16))) float red[BNP ? 2048 : 4];
  const int tid = threadIdx.x, c4n = C / 4, per_row = Wi * c4n;
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1, bsc = s1, bsh = s1, bmu = s1, brs = s1;
  if (BNP) {   // this thread's channel quad never changes: j = tid + 256 k and 256 % c4n == 0
    const int cq = (tid % c4n) * 4;
    bsc = *reinterpret_cast<const float4*>(bw.scale + cq);
    bsh = *reinterpret_cast<const float4*>(bw.shift + cq);
    bmu = *reinterpret_cast<const float4*>(bw.mean + cq);
    brs = *reinterpret_cast<const float4*>(bw.rstd + cq);
  }
  // both axes' windows depend only on the index along the axis: all of them once per workgroup (Hi, Wi <= 64),
  // then rows are walked with no barrier and no tap arithmetic in the loop
  if (tid < Wi) axis_window(tid, scw, Wi, Wo, s_ww[tid], s_wlo[tid], s_wn[tid]);
  else if (tid >= 64 && tid - 64 < Hi) axis_window(tid - 64, sch, Hi, Ho, s_wh[tid - 64], s_hlo[tid - 64], s_hn[tid - 64]);
  __syncthreads();
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const int b = r / Hi, ih = r - b * Hi;
    const int hlo = s_hlo[ih], hn = s_hn[ih];
    const T* grow = g_out + ((int64_t)b * Ho + hlo) * Wo * gcs + gcoff;
    T* orow = g_in + (int64_t)r * per_row * 4;
    for (int j = tid; j < per_row; j += 256) {
      const int iw = j / c4n, c = (j - iw * c4n) * 4;
      const int wlo = s_wlo[iw], wn = s_wn[iw];
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (wn <= BIL_BATCH) {
        // the usual case (a 2x up-sampling has at most five contributors per axis): the row's loads are issued
        // together instead of one per loop trip - the trip-at-a-time form ran at a fifth of its HBM time.
        // Same contributors, same order, zero weights skipped as below: bit-identical.
        float wd[BIL_BATCH];
#pragma unroll
        for (int d = 0; d < BIL_BATCH; ++d) wd[d] = d < wn ? s_ww[iw][d] : 0.f;
        for (int a = 0; a < hn; ++a) {
          const float wa = s_wh[ih][a];
          if (wa == 0.f) continue;
          const T* gp = grow + ((int64_t)a * Wo + wlo) * gcs + c;
          float4 g[BIL_BATCH];
#pragma unroll
          for (int d = 0; d < BIL_BATCH; ++d)
            g[d] = d < wn ? ld4(gp + (int64_t)d * gcs) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int d = 0; d < BIL_BATCH; ++d) {
            if (wd[d] == 0.f) continue;
            const float w = wa * wd[d];
            acc.x = fmaf(w, g[d].x, acc.x); acc.y = fmaf(w, g[d].y, acc.y);
            acc.z = fmaf(w, g[d].z, acc.z); acc.w = fmaf(w, g[d].w, acc.w);
          }
        }
      } else {
        for (int a = 0; a < hn; ++a) {
          const float wa = s_wh[ih][a];
          if (wa == 0.f) continue;
          const T* gp = grow + ((int64_t)a * Wo + wlo) * gcs + c;
          for (int d = 0; d < wn; ++d) {
            const float wd = s_ww[iw][d];
            if (wd == 0.f) continue;
            const float4 g = ld4(gp + (int64_t)d * gcs);
            const float w = wa * wd;
            acc.x = fmaf(w, g.x, acc.x); acc.y = fmaf(w, g.y, acc.y);
            acc.z = fmaf(w, g.z, acc.z); acc.w = fmaf(w, g.w, acc.w);
          }
        }
      }
      st4(orow + (int64_t)j * 4, acc);
      if (BNP) {
        const float4 yv = ld4(static_cast<const T*>(bw.y) + (int64_t)r * per_row * 4 + (int64_t)j * 4);
        bnbwd_acc(acc, yv, bsc, bsh, bmu, brs, s1, s2);
      }
    }
  }
  if (BNP) bnbwd_flush(red, s1, s2, C, bw.partial);
}

// host-side check that BIL_MAXC candidates cover every contributor of every input index
static bool adjoint_window_ok(int n_in, int n_out) {
  const float scale = ac_scale(n_in, n_out);
  if (!(scale > 0.f)) return n_out <= BIL_MAXC;
  for (int i = 0; i < n_in; ++i) {
    int lo = (int)floorf((float)(i - 1) / scale) - 1;
    if (lo < 0) lo = 0;
    int hi = (int)ceilf((float)(i + 1) / scale) + 1;
    if (hi > n_out - 1) hi = n_out - 1;
    if (hi - lo + 1 > BIL_MAXC) return false;
  }
  return true;
}

static int bilinear_ac_bwd_t(const void* g_out, void* g_in, int B, int Hi, int Wi, int Ho, int Wo, int C, int g_cstride,
                             int g_coff, int io16, tdx_stream_t stream) {
  if (!g_out || !g_in || B <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || C <= 0) return TDX_E_BADARG;
  if (C % 4 || g_cstride % 4 || g_coff % 4 || g_coff + C > g_cstride) return TDX_E_SHAPE;
  if (!adjoint_window_ok(Hi, Ho) || !adjoint_window_ok(Wi, Wo)) return TDX_E_SHAPE;
  const int64_t n = (int64_t)B * Hi * Wi * (C / 4);
  if (Wi <= BIL_ROW_MAXW && Hi <= BIL_ROW_MAXW && (int64_t)B * Hi < (1 << 30)) {
    const int rows = B * Hi;
    TDX_IO_DISPATCH(io16, T, bilinear_bwd_rows_kernel<false, T><<<rows < 16384 ? rows : 16384, 256, 0, to_stream(stream)>>>(
        (const T*)g_out, (T*)g_in, rows, Hi, Wi, Ho, Wo, C, g_cstride, g_coff, ac_scale(Hi, Ho), ac_scale(Wi, Wo), BnBwdOps{}));
    TDX_CHECK_LAUNCH();
    return 0;
  }
  TDX_IO_DISPATCH(io16, T, bilinear_bwd_kernel<T><<<ew_grid(n), 256, 0, to_stream(stream)>>>(
      (const T*)g_out, (T*)g_in, B, Hi, Wi, Ho, Wo, C, g_cstride, g_coff, ac_scale(Hi, Ho), ac_scale(Wi, Wo)));
  TDX_CHECK_LAUNCH();
  return 0;
}
extern "C" int tdx_bilinear_ac_bwd(const float* g_out, float* g_in, int B, int Hi, int Wi, int Ho,
                                   int Wo, int C, int g_cstride, int g_coff, tdx_stream_t stream) {
  return bilinear_ac_bwd_t(g_out, g_in, B, Hi, Wi, Ho, Wo, C, g_cstride, g_coff, 0, stream);
}

int tdx_bilinear_ac_bwd_bn(const void* g_out, void* g_in, int B, int Hi, int Wi, int Ho, int Wo, int C,
                           int g_cstride, int g_coff, const void* bn_y, const float* bn_scale,
                           const float* bn_shift, const float* bn_mean, const float* bn_rstd, float* partial,
                           int* nblk, tdx_stream_t stream, int io16) {
  *nblk = 0;
  const bool rows_form = Wi <= BIL_ROW_MAXW && Hi <= BIL_ROW_MAXW && (int64_t)B * Hi < (1 << 30);
  if (!(g_tdx_bnbwd_fused & 2) || !bn_y || !partial || !rows_form || C % 4 || C > 1024 || 256 % (C / 4))
    return bilinear_ac_bwd_t(g_out, g_in, B, Hi, Wi, Ho, Wo, C, g_cstride, g_coff, io16, stream);
  if (!g_out || !g_in || !bn_scale || !bn_shift || !bn_mean || !bn_rstd || B <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 ||
      Wo <= 0)
    return TDX_E_BADARG;
  if (g_cstride % 4 || g_coff % 4 || g_coff + C > g_cstride) return TDX_E_SHAPE;
  if (!adjoint_window_ok(Hi, Ho) || !adjoint_window_ok(Wi, Wo)) return TDX_E_SHAPE;
  const int rows = B * Hi;
  const int grid = rows < TDX_BNBWD_MAX_PRODUCER_BLOCKS ? rows : TDX_BNBWD_MAX_PRODUCER_BLOCKS;
  TDX_IO_DISPATCH(io16, T, bilinear_bwd_rows_kernel<true, T><<<grid, 256, 0, to_stream(stream)>>>(
      (const T*)g_out, (T*)g_in, rows, Hi, Wi, Ho, Wo, C, g_cstride, g_coff, ac_scale(Hi, Ho), ac_scale(Wi, Wo),
      BnBwdOps{bn_y, bn_scale, bn_shift, bn_mean, bn_rstd, partial}));
  TDX_CHECK_LAUNCH();
  *nblk = grid;
  return 0;
}

// ------------------------------------------------- per-(n,c) sum over pixels
// out[n][c] = sum_{h,w} g[n][h][w][c]   (gradient of the broadcast time/class add)
template <typename T>
__global__ void __launch_bounds__(256)
pixel_sum_kernel(const T* __restrict__ g, float* __restrict__ out, int HW, int C) {
  extern __shared__ float red[];  // [rgroups][C]
  const int c4n = C / 4, col = threadIdx.x % c4n, rg = threadIdx.x / c4n, rgroups = 256 / c4n;
  const T* base = g + (int64_t)blockIdx.x * HW * C + col * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int p = rg; p < HW; p += rgroups) {
    const float4 v = ld4(base + (int64_t)p * C);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  *reinterpret_cast<float4*>(red + rg * C + col * 4) = s;
  __syncthreads();
  for (int i = threadIdx.x; i < C; i += 256) {
    float v = 0.f;
    for (int k = 0; k < rgroups; ++k) v += red[k * C + i];
    out[(int64_t)blockIdx.x * C + i] = v;
  }
}

int tdx_pixel_sum(const void* g, float* out, int B, int HW, int C, hipStream_t st, int io16) {
  if (C % 4 || C > 1024 || 256 % (C / 4)) return TDX_E_SHAPE;
  const int rgroups = 256 / (C / 4);
  TDX_IO_DISPATCH(io16, T, pixel_sum_kernel<T><<<B, 256, (size_t)rgroups * C * sizeof(float), st>>>((const T*)g, out, HW, C));
  TDX_CHECK_LAUNCH();
  return 0;
}

// initial_conv / final_conv and their backward passes: edge_conv.hip

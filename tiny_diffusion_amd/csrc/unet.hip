// Whole-network orchestration of NoiseModel.forward (diffusion.py:109-162,
// conditional_diffusion.py:115-172) and its autograd backward, as a fixed
// sequence of kernel launches on one HIP stream.  No allocation, no host sync:
// the sequence can be captured into a hipGraph by the caller.
//
// Data flow (channels-last, fp32).  Only PRE-BatchNorm convolution outputs Y[u]
// are stored; every consumer applies relu(y*scale+shift) on load:
//
//   x -> initial_conv -> x0 -> U0 -> U1 -+-> pool -> U2 -> U3 -+-> pool -> U4 -> U5 -+-> pool -> U6
//                                        |                     |                     |           |
//        cat1 = [up(U10) | rs(U1 + t1)] <+  cat2 = [up(U8) | rs(U3 + t2)]  cat3 = [up(U6) | rs(U5 + t3)]
//   cat3 -> U7 -> U8 ; cat2 -> U9 -> U10 ; cat1 -> U11 -> U12 -> resize 32->28 -> final_conv -> eps
//
// The channel concatenations (diffusion.py:140,147,154) cost nothing: the two
// resize kernels write straight into the two channel halves of one buffer.
#include "internal.h"
#include <algorithm>
#include <new>
#include <string>
#include <cstdlib>

namespace {

struct UnitDef {
  int cin, cout, hw;   // square maps
  int in_bn;           // 1: input is Y[u-1] (BN+ReLU on load); 0: raw tensor
};

// enc1.0 enc1.3 enc2.0 enc2.3 enc3.0 enc3.3 bottleneck dec3.0 dec3.3 dec2.0 dec2.3 dec1.0 dec1.3
const UnitDef UNITS[13] = {
    {64, 128, 28, 0},  {128, 128, 28, 1}, {128, 256, 14, 0}, {256, 256, 14, 1}, {256, 512, 7, 0},
    {512, 512, 7, 1},  {512, 512, 4, 0},  {1024, 256, 8, 0}, {256, 256, 8, 1},  {512, 128, 16, 0},
    {128, 128, 16, 1}, {256, 64, 32, 0},  {64, 64, 32, 1}};

constexpr int TD = 256;
constexpr int N_STAGES = 15;

inline size_t align64(size_t n) { return (n + 63) / 64 * 64; }

// Workspace layout in floats, a pure function of the batch size.
struct Layout {
  size_t x, t, y;                       // copies of the inputs (t, y are int64 -> 2 floats each)
  size_t pre, emb, t1, t2, t3;          // time path
  size_t x0, Y[13], ss[13];             // ss: scale | shift | mean | rstd (4*cout)
  size_t e1p, e2p, e3p, cat3, cat2, cat1, d1a;
  size_t stats;                         // conv epilogue partials (largest unit)
  // backward
  size_t G1, G2, GS1, GS2, GS3, gt1, gt2, gt3, slabs, bnscr, smallp, timescr;
  size_t total;
};

Layout make_layout(int B) {
  Layout L;
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o += align64(n); return r; };
  const size_t b = (size_t)B;
  L.x = take(b * 784); L.t = take(2 * b); L.y = take(2 * b);
  L.pre = take(b * TD); L.emb = take(b * TD);
  L.t1 = take(b * 128); L.t2 = take(b * 256); L.t3 = take(b * 512);
  L.x0 = take(b * 784 * 64);
  size_t stats = 0, slabs = 0, bnscr = 0;
  for (int u = 0; u < 13; ++u) {
    const UnitDef& d = UNITS[u];
    L.Y[u] = take(b * d.hw * d.hw * d.cout);
    L.ss[u] = take(4 * (size_t)d.cout);
    stats = std::max(stats, (size_t)tdx_conv3x3_stat_tiles(B, d.hw, d.hw, d.cin, d.cout) * 2 * d.cout);
    slabs = std::max(slabs, (size_t)tdx_conv3x3_wgrad_splits(B, d.hw, d.hw, d.cin, d.cout) * 9 *
                                (size_t)d.cin * d.cout);
    bnscr = std::max(bnscr, tdx_bn_relu_bwd_scratch_floats((int64_t)b * d.hw * d.hw, d.cout));
  }
  L.e1p = take(b * 14 * 14 * 128); L.e2p = take(b * 7 * 7 * 256); L.e3p = take(b * 4 * 4 * 512);
  L.cat3 = take(b * 8 * 8 * 1024); L.cat2 = take(b * 16 * 16 * 512); L.cat1 = take(b * 32 * 32 * 256);
  L.d1a = take(b * 784 * 64);
  L.stats = take(stats);
  L.G1 = take(b * 32 * 32 * 256); L.G2 = take(b * 32 * 32 * 256);
  L.GS1 = take(b * 784 * 128); L.GS2 = take(b * 196 * 256); L.GS3 = take(b * 49 * 512);
  L.gt1 = take(b * 128); L.gt2 = take(b * 256); L.gt3 = take(b * 512);
  L.slabs = take(slabs);
  L.bnscr = take(bnscr);
  L.smallp = take((size_t)tdx_initial_conv_wgrad_blocks(B, 28, 28) * 640);
  L.timescr = take(3 * b * TD);
  L.total = o;
  return L;
}

}  // namespace

struct tdx_unet {
  int max_batch, num_classes;
  float* wpack;            // device: per unit fwd pack then dgrad pack
  size_t wf_off[13], wd_off[13];
  float* infer_ss;         // device: per unit scale|shift from running stats (INFER mode)
  size_t iss_off[13];
  bool packed;
  int saved_batch, saved_mode;  // state of the last forward (for backward)
  // backward runs the weight-gradient GEMMs on a second (low-priority) HIP stream so that
  // they fill the tail of the input-gradient GEMM and overlap the HBM-bound BN/pool/resize
  // kernels of the next unit; fork/join with events, so the caller still sees ONE stream
  hipStream_t side;
  hipEvent_t ev_dy[13], ev_w[13], ev_join;
};

extern "C" int tdx_unet_create(tdx_unet** out, int max_batch, int num_classes) {
  if (!out || max_batch <= 0 || num_classes < 0) return TDX_E_BADARG;
  tdx_unet* u = new (std::nothrow) tdx_unet();
  if (!u) return TDX_E_BADARG;
  u->max_batch = max_batch;
  u->num_classes = num_classes;
  size_t o = 0, so = 0;
  for (int i = 0; i < 13; ++i) {
    const size_t n = (size_t)UNITS[i].cin * UNITS[i].cout * 9;
    u->wf_off[i] = o; o += align64(n);
    u->wd_off[i] = o; o += align64(n);
    u->iss_off[i] = so; so += align64(2 * (size_t)UNITS[i].cout);
  }
  hipError_t e = hipMalloc(&u->wpack, o * sizeof(float));
  if (e != hipSuccess) { delete u; return (int)e; }
  e = hipMalloc(&u->infer_ss, so * sizeof(float));
  if (e != hipSuccess) { hipFree(u->wpack); delete u; return (int)e; }
  u->packed = false;
  u->saved_batch = 0;
  u->saved_mode = -1;
  int lo = 0, hi = 0;
  hipDeviceGetStreamPriorityRange(&lo, &hi);  // lo = least urgent
  e = hipStreamCreateWithPriority(&u->side, hipStreamNonBlocking, lo);
  if (e != hipSuccess) { hipFree(u->wpack); hipFree(u->infer_ss); delete u; return (int)e; }
  for (int i = 0; i < 13; ++i) {
    hipEventCreateWithFlags(&u->ev_dy[i], hipEventDisableTiming);
    hipEventCreateWithFlags(&u->ev_w[i], hipEventDisableTiming);
  }
  hipEventCreateWithFlags(&u->ev_join, hipEventDisableTiming);
  *out = u;
  return 0;
}

extern "C" int tdx_unet_destroy(tdx_unet* u) {
  if (!u) return TDX_E_BADARG;
  hipStreamSynchronize(u->side);
  for (int i = 0; i < 13; ++i) {
    hipEventDestroy(u->ev_dy[i]);
    hipEventDestroy(u->ev_w[i]);
  }
  hipEventDestroy(u->ev_join);
  hipStreamDestroy(u->side);
  hipFree(u->wpack);
  hipFree(u->infer_ss);
  delete u;
  return 0;
}

extern "C" size_t tdx_unet_workspace_bytes(const tdx_unet* u, int batch, int mode) {
  (void)mode;
  if (!u || batch <= 0 || batch > u->max_batch) return 0;
  return make_layout(batch).total * sizeof(float);
}

extern "C" int tdx_unet_backward_stages(void) { return N_STAGES; }

// Where a named intermediate lives inside the workspace (testing / debugging aid):
// "x0", "Y0".."Y12" (pre-BN conv outputs; post-activation in INFER mode), "ss0".."ss12"
// (scale|shift|mean|rstd), "e1p" "e2p" "e3p" "cat3" "cat2" "cat1" "d1a" "emb" "t1" "t2" "t3",
// "G1" "G2" "GS1" "GS2" "GS3".
extern "C" int tdx_unet_tensor(const tdx_unet* u, int batch, const char* name, size_t* offset_floats,
                               size_t* numel) {
  if (!u || !name || !offset_floats || !numel || batch <= 0 || batch > u->max_batch) return TDX_E_BADARG;
  const Layout L = make_layout(batch);
  const size_t b = (size_t)batch;
  std::string n(name);
  auto unit_index = [&](const std::string& s, size_t prefix) -> int {
    if (s.size() <= prefix) return -1;
    int v = atoi(s.c_str() + prefix);
    return (v >= 0 && v < 13) ? v : -1;
  };
  struct { const char* nm; size_t off, cnt; } fixed[] = {
      {"x0", L.x0, b * 784 * 64}, {"e1p", L.e1p, b * 196 * 128}, {"e2p", L.e2p, b * 49 * 256},
      {"e3p", L.e3p, b * 16 * 512}, {"cat3", L.cat3, b * 64 * 1024}, {"cat2", L.cat2, b * 256 * 512},
      {"cat1", L.cat1, b * 1024 * 256}, {"d1a", L.d1a, b * 784 * 64}, {"emb", L.emb, b * TD},
      {"t1", L.t1, b * 128}, {"t2", L.t2, b * 256}, {"t3", L.t3, b * 512},
      {"G1", L.G1, b * 1024 * 256}, {"G2", L.G2, b * 1024 * 256}, {"GS1", L.GS1, b * 784 * 128},
      {"GS2", L.GS2, b * 196 * 256}, {"GS3", L.GS3, b * 49 * 512}};
  for (auto& f : fixed)
    if (n == f.nm) { *offset_floats = f.off; *numel = f.cnt; return 0; }
  if (n.rfind("ss", 0) == 0) {
    int i = unit_index(n, 2);
    if (i < 0) return TDX_E_BADARG;
    *offset_floats = L.ss[i]; *numel = 4 * (size_t)UNITS[i].cout; return 0;
  }
  if (n[0] == 'Y') {
    int i = unit_index(n, 1);
    if (i < 0) return TDX_E_BADARG;
    *offset_floats = L.Y[i]; *numel = b * UNITS[i].hw * UNITS[i].hw * UNITS[i].cout; return 0;
  }
  return TDX_E_BADARG;
}

// weights always; the INFER-mode scale/shift (from the running statistics) only when
// `buffers` is given
static int pack_impl(tdx_unet* u, const void* const* params, void* const* buffers,
                     tdx_stream_t stream) {
  const float* const* P = reinterpret_cast<const float* const*>(params);
  for (int i = 0; i < 13; ++i) {
    int rc = tdx_pack_conv3x3(P[TDX_P_UNIT0 + 4 * i], u->wpack + u->wf_off[i],
                              u->wpack + u->wd_off[i], UNITS[i].cout, UNITS[i].cin, stream);
    if (rc) return rc;
    if (buffers) {
      float* ss = u->infer_ss + u->iss_off[i];
      rc = tdx_bn_finalize(nullptr, 0, 0, 0, UNITS[i].cout, P[TDX_P_UNIT0 + 4 * i + 2],
                           P[TDX_P_UNIT0 + 4 * i + 3], (float*)buffers[3 * i],
                           (float*)buffers[3 * i + 1], nullptr, ss, ss + UNITS[i].cout, nullptr,
                           nullptr, 0, stream);
      if (rc) return rc;
    }
  }
  u->packed = buffers != nullptr;
  return 0;
}

extern "C" int tdx_unet_pack(tdx_unet* u, const void* const* params, void* const* buffers,
                             tdx_stream_t stream) {
  if (!u || !params || !buffers) return TDX_E_BADARG;
  return pack_impl(u, params, buffers, stream);
}

#define RC(call)            \
  do {                      \
    int rc__ = (call);      \
    if (rc__) return rc__;  \
  } while (0)

extern "C" int tdx_unet_forward(tdx_unet* u, const void* const* params, void* const* buffers,
                                const float* x, const int64_t* t, const int64_t* y, float* out,
                                void* workspace, size_t workspace_bytes, int batch, int mode,
                                tdx_stream_t stream) {
  if (!u || !params || !buffers || !x || !t || !out || !workspace) return TDX_E_BADARG;
  if (batch <= 0 || batch > u->max_batch) return TDX_E_BADARG;
  if (mode < TDX_MODE_TRAIN || mode > TDX_MODE_INFER) return TDX_E_BADARG;
  if ((u->num_classes > 0) != (y != nullptr)) return TDX_E_BADARG;
  const Layout L = make_layout(batch);
  if (workspace_bytes < L.total * sizeof(float)) return TDX_E_WORKSPACE;
  const float* const* P = reinterpret_cast<const float* const*>(params);
  float* ws = reinterpret_cast<float*>(workspace);
  hipStream_t st = to_stream(stream);
  const int B = batch;
  const bool infer = mode == TDX_MODE_INFER;
  const bool training = mode == TDX_MODE_TRAIN;

  if (!infer) RC(pack_impl(u, params, nullptr, stream));  // weights change every step
  else if (!u->packed) RC(pack_impl(u, params, buffers, stream));
  if (!infer) {
    // keep the inputs for backward (caller tensors may be gone by then)
    TDX_HIP(hipMemcpyAsync(ws + L.x, x, (size_t)B * 784 * sizeof(float), hipMemcpyDeviceToDevice, st));
    TDX_HIP(hipMemcpyAsync(ws + L.t, t, (size_t)B * sizeof(int64_t), hipMemcpyDeviceToDevice, st));
    if (y) TDX_HIP(hipMemcpyAsync(ws + L.y, y, (size_t)B * sizeof(int64_t), hipMemcpyDeviceToDevice, st));
  }

  RC(tdx_time_embed_fwd(t, y, P, ws + L.pre, ws + L.emb, ws + L.t1, ws + L.t2, ws + L.t3, B, st));
  RC(tdx_initial_conv_fwd(x, P[TDX_P_INIT_W], P[TDX_P_INIT_B], ws + L.x0, B, 28, 28, st));

  // scale/shift of unit i as seen by its consumers (null in INFER mode: already applied)
  auto sc = [&](int i) -> const float* { return infer ? nullptr : ws + L.ss[i]; };
  auto sh = [&](int i) -> const float* { return infer ? nullptr : ws + L.ss[i] + UNITS[i].cout; };

  auto run_unit = [&](int i, const float* in) -> int {
    const UnitDef& d = UNITS[i];
    const float* wf = u->wpack + u->wf_off[i];
    const float* bias = P[TDX_P_UNIT0 + 4 * i + 1];
    float* Y = ws + L.Y[i];
    float* ss = ws + L.ss[i];
    if (infer) {
      const float* iss = u->infer_ss + u->iss_off[i];
      // small-batch sampling is latency-bound: split K over more workgroups where the tile
      // grid would not fill the chip; the (unused in INFER mode) gradient buffer is the scratch
      return tdx_conv3x3_fwd_splitk(in, wf, bias, Y, B, d.hw, d.hw, d.cin, d.cout, TDX_CONV_OUT_BNRELU,
                                    nullptr, nullptr, iss, iss + d.cout, ws + L.G1,
                                    (size_t)B * 32 * 32 * 256 * 2, stream);
    }
    int flags = (d.in_bn ? TDX_CONV_IN_BNRELU : 0) | (training ? TDX_CONV_OUT_STATS : 0);
    RC(tdx_conv3x3_fwd(in, wf, bias, Y, B, d.hw, d.hw, d.cin, d.cout, flags,
                       d.in_bn ? sc(i - 1) : nullptr, d.in_bn ? sh(i - 1) : nullptr, nullptr, nullptr,
                       ws + L.stats, stream));
    const int tiles = tdx_conv3x3_stat_tiles(B, d.hw, d.hw, d.cin, d.cout);
    return tdx_bn_finalize(ws + L.stats, tiles,
                           tdx_conv3x3_stat_tile_rows(B, d.hw, d.hw, d.cin, d.cout),
                           (int64_t)B * d.hw * d.hw, d.cout,
                           P[TDX_P_UNIT0 + 4 * i + 2], P[TDX_P_UNIT0 + 4 * i + 3],
                           (float*)buffers[3 * i], (float*)buffers[3 * i + 1],
                           (int64_t*)buffers[3 * i + 2], ss, ss + d.cout, ss + 2 * d.cout,
                           ss + 3 * d.cout, training ? 1 : 0, stream);
  };

  // encoder
  RC(run_unit(0, ws + L.x0));
  RC(run_unit(1, ws + L.Y[0]));
  RC(tdx_maxpool2_ceil_fwd(ws + L.Y[1], sc(1), sh(1), ws + L.e1p, B, 28, 28, 128, stream));
  RC(run_unit(2, ws + L.e1p));
  RC(run_unit(3, ws + L.Y[2]));
  RC(tdx_maxpool2_ceil_fwd(ws + L.Y[3], sc(3), sh(3), ws + L.e2p, B, 14, 14, 256, stream));
  RC(run_unit(4, ws + L.e2p));
  RC(run_unit(5, ws + L.Y[4]));
  RC(tdx_maxpool2_ceil_fwd(ws + L.Y[5], sc(5), sh(5), ws + L.e3p, B, 7, 7, 512, stream));
  RC(run_unit(6, ws + L.e3p));
  // decoder level 3: cat3 = [up(b) | resize(e3 + t3)], diffusion.py:135-140
  RC(tdx_bilinear_ac_fwd(ws + L.Y[6], sc(6), sh(6), nullptr, ws + L.cat3, B, 4, 4, 8, 8, 512, 1024, 0, stream));
  RC(tdx_bilinear_ac_fwd(ws + L.Y[5], sc(5), sh(5), ws + L.t3, ws + L.cat3, B, 7, 7, 8, 8, 512, 1024, 512, stream));
  RC(run_unit(7, ws + L.cat3));
  RC(run_unit(8, ws + L.Y[7]));
  // level 2, diffusion.py:142-147
  RC(tdx_bilinear_ac_fwd(ws + L.Y[8], sc(8), sh(8), nullptr, ws + L.cat2, B, 8, 8, 16, 16, 256, 512, 0, stream));
  RC(tdx_bilinear_ac_fwd(ws + L.Y[3], sc(3), sh(3), ws + L.t2, ws + L.cat2, B, 14, 14, 16, 16, 256, 512, 256, stream));
  RC(run_unit(9, ws + L.cat2));
  RC(run_unit(10, ws + L.Y[9]));
  // level 1, diffusion.py:149-154
  RC(tdx_bilinear_ac_fwd(ws + L.Y[10], sc(10), sh(10), nullptr, ws + L.cat1, B, 16, 16, 32, 32, 128, 256, 0, stream));
  RC(tdx_bilinear_ac_fwd(ws + L.Y[1], sc(1), sh(1), ws + L.t1, ws + L.cat1, B, 28, 28, 32, 32, 128, 256, 128, stream));
  RC(run_unit(11, ws + L.cat1));
  RC(run_unit(12, ws + L.Y[11]));
  // 32 -> 28 and the output convolution, diffusion.py:157-160
  RC(tdx_bilinear_ac_fwd(ws + L.Y[12], sc(12), sh(12), nullptr, ws + L.d1a, B, 32, 32, 28, 28, 64, 64, 0, stream));
  RC(tdx_final_conv_fwd(ws + L.d1a, P[TDX_P_FINAL_W], P[TDX_P_FINAL_B], out, B, 28, 28, st));

  u->saved_batch = infer ? 0 : B;
  u->saved_mode = mode;
  return 0;
}

extern "C" int tdx_unet_backward(tdx_unet* u, const void* const* params, void* const* grads,
                                 const float* d_out, void* workspace, size_t workspace_bytes,
                                 int batch, int stage_lo, int stage_hi, tdx_stream_t stream) {
  if (!u || !params || !grads || !d_out || !workspace) return TDX_E_BADARG;
  if (batch != u->saved_batch || u->saved_mode == TDX_MODE_INFER || u->saved_mode < 0) return TDX_E_STATE;
  if (stage_lo < 0 || stage_hi > N_STAGES || stage_lo >= stage_hi) return TDX_E_BADARG;
  const Layout L = make_layout(batch);
  if (workspace_bytes < L.total * sizeof(float)) return TDX_E_WORKSPACE;
  const float* const* P = reinterpret_cast<const float* const*>(params);
  float* const* G = reinterpret_cast<float* const*>(grads);
  float* ws = reinterpret_cast<float*>(workspace);
  hipStream_t st = to_stream(stream);
  const int B = batch;
  const int training = u->saved_mode == TDX_MODE_TRAIN ? 1 : 0;
  float* G1 = ws + L.G1;
  float* G2 = ws + L.G2;

  // Gradient buffers alternate between G1 and G2 in a fixed pattern (see the table in
  // DESIGN.md); `g_of[u]` is where the gradient w.r.t. unit u's activation lives when
  // its stage starts.
  // unit:            0   1   2   3   4   5   6   7   8   9   10  11  12
  float* g_of[13] = {G2, G1, G1, G2, G2, G1, G1, G1, G2, G2, G1, G1, G2};

  tdx_stream_t side = reinterpret_cast<tdx_stream_t>(u->side);
  int pending_w = -1;  // unit whose wgrad (side stream) still reads its dy buffer
  // Every kernel on the main stream that OVERWRITES the buffer holding dy of `pending_w`
  // must first wait for that unit's wgrad.
  auto wait_wgrad = [&]() -> int {
    if (pending_w >= 0) {
      TDX_HIP(hipStreamWaitEvent(st, u->ev_w[pending_w], 0));
      pending_w = -1;
    }
    return 0;
  };
  auto unit_bwd = [&](int i, const float* in, float* g_in) -> int {
    // g_of[i] holds dL/d(activation of unit i); afterwards it holds dL/d(conv output)
    const UnitDef& d = UNITS[i];
    float* g = g_of[i];
    const float* ss = ws + L.ss[i];
    const int64_t rows = (int64_t)B * d.hw * d.hw;
    RC(tdx_bn_relu_bwd(g, ws + L.Y[i], rows, d.cout, ss, ss + d.cout, ss + 2 * d.cout,
                       ss + 3 * d.cout, P[TDX_P_UNIT0 + 4 * i + 2], G[TDX_P_UNIT0 + 4 * i + 2],
                       G[TDX_P_UNIT0 + 4 * i + 3], G[TDX_P_UNIT0 + 4 * i + 1], ws + L.bnscr, training,
                       stream));
    // fork: weight gradient on the side stream (input is the previous unit's pre-BN tensor
    // when in_bn).  The slab buffer is only ever touched by the side stream, in order.
    TDX_HIP(hipEventRecord(u->ev_dy[i], st));
    TDX_HIP(hipStreamWaitEvent(u->side, u->ev_dy[i], 0));
    const float* isc = d.in_bn ? ws + L.ss[i - 1] : nullptr;
    const float* ish = d.in_bn ? ws + L.ss[i - 1] + UNITS[i - 1].cout : nullptr;
    RC(tdx_conv3x3_wgrad(in, g, ws + L.slabs, B, d.hw, d.hw, d.cin, d.cout,
                         d.in_bn ? TDX_CONV_IN_BNRELU : 0, isc, ish, side));
    RC(tdx_conv3x3_wgrad_reduce(ws + L.slabs, G[TDX_P_UNIT0 + 4 * i],
                                tdx_conv3x3_wgrad_splits(B, d.hw, d.hw, d.cin, d.cout), d.cout, d.cin,
                                side));
    TDX_HIP(hipEventRecord(u->ev_w[i], u->side));
    // main: input gradient = the forward kernel on the flipped pack, channels swapped.
    // It writes g_in (the OTHER ping-pong buffer, whose previous dy reader must be done).
    if (g_in) {
      RC(wait_wgrad());
      RC(tdx_conv3x3_fwd(g, u->wpack + u->wd_off[i], nullptr, g_in, B, d.hw, d.hw, d.cout, d.cin, 0,
                         nullptr, nullptr, nullptr, nullptr, nullptr, stream));
    }
    pending_w = i;
    return 0;
  };
  auto ssc = [&](int i) { return ws + L.ss[i]; };
  auto ssh = [&](int i) { return ws + L.ss[i] + UNITS[i].cout; };

  for (int s = stage_lo; s < stage_hi; ++s) {
    switch (s) {
      case 0:  // final_conv + the 32->28 resize
        RC(tdx_final_conv_wgrad(ws + L.d1a, d_out, ws + L.smallp, G[TDX_P_FINAL_W], G[TDX_P_FINAL_B], B, 28, 28, st));
        RC(tdx_final_conv_dgrad(d_out, P[TDX_P_FINAL_W], G1, B, 28, 28, st));
        RC(tdx_bilinear_ac_bwd(G1, G2, B, 32, 32, 28, 28, 64, 64, 0, stream));  // -> g(U12) in G2
        break;
      case 1: RC(unit_bwd(12, ws + L.Y[11], G1)); break;  // g(U11) in G1
      case 2:
        RC(unit_bwd(11, ws + L.cat1, G2));  // g(cat1) in G2
        RC(wait_wgrad());                   // G1 (dy11) is overwritten next
        RC(tdx_bilinear_ac_bwd(G2, G1, B, 16, 16, 32, 32, 128, 256, 0, stream));          // g(U10) in G1
        RC(tdx_bilinear_ac_bwd(G2, ws + L.GS1, B, 28, 28, 32, 32, 128, 256, 128, stream));  // g(e1+t1)
        RC(tdx_pixel_sum(ws + L.GS1, ws + L.gt1, B, 784, 128, st));
        break;
      case 3: RC(unit_bwd(10, ws + L.Y[9], G2)); break;  // g(U9) in G2
      case 4:
        RC(unit_bwd(9, ws + L.cat2, G1));  // g(cat2) in G1
        RC(wait_wgrad());
        RC(tdx_bilinear_ac_bwd(G1, G2, B, 8, 8, 16, 16, 256, 512, 0, stream));            // g(U8) in G2
        RC(tdx_bilinear_ac_bwd(G1, ws + L.GS2, B, 14, 14, 16, 16, 256, 512, 256, stream));
        RC(tdx_pixel_sum(ws + L.GS2, ws + L.gt2, B, 196, 256, st));
        break;
      case 5: RC(unit_bwd(8, ws + L.Y[7], G1)); break;  // g(U7) in G1
      case 6:
        RC(unit_bwd(7, ws + L.cat3, G2));  // g(cat3) in G2
        RC(wait_wgrad());
        RC(tdx_bilinear_ac_bwd(G2, G1, B, 4, 4, 8, 8, 512, 1024, 0, stream));             // g(U6) in G1
        RC(tdx_bilinear_ac_bwd(G2, ws + L.GS3, B, 7, 7, 8, 8, 512, 1024, 512, stream));
        RC(tdx_pixel_sum(ws + L.GS3, ws + L.gt3, B, 49, 512, st));
        break;
      case 7:
        RC(unit_bwd(6, ws + L.e3p, G2));  // g(e3p) in G2
        RC(wait_wgrad());
        RC(tdx_maxpool2_ceil_bwd(ws + L.Y[5], ssc(5), ssh(5), G2, ws + L.GS3, G1, B, 7, 7, 512, stream));  // g(U5) in G1
        break;
      case 8: RC(unit_bwd(5, ws + L.Y[4], G2)); break;  // g(U4) in G2
      case 9:
        RC(unit_bwd(4, ws + L.e2p, G1));  // g(e2p) in G1
        RC(wait_wgrad());
        RC(tdx_maxpool2_ceil_bwd(ws + L.Y[3], ssc(3), ssh(3), G1, ws + L.GS2, G2, B, 14, 14, 256, stream));  // g(U3) in G2
        break;
      case 10: RC(unit_bwd(3, ws + L.Y[2], G1)); break;  // g(U2) in G1
      case 11:
        RC(unit_bwd(2, ws + L.e1p, G2));  // g(e1p) in G2
        RC(wait_wgrad());
        RC(tdx_maxpool2_ceil_bwd(ws + L.Y[1], ssc(1), ssh(1), G2, ws + L.GS1, G1, B, 28, 28, 128, stream));  // g(U1) in G1
        break;
      case 12: RC(unit_bwd(1, ws + L.Y[0], G2)); break;  // g(U0) in G2
      case 13: RC(unit_bwd(0, ws + L.x0, G1)); break;    // g(x0) in G1
      case 14:
        RC(tdx_initial_conv_wgrad(ws + L.x, G1, ws + L.smallp, G[TDX_P_INIT_W], G[TDX_P_INIT_B], B, 28, 28, st));
        RC(tdx_time_embed_bwd(reinterpret_cast<const int64_t*>(ws + L.t),
                              u->num_classes > 0 ? reinterpret_cast<const int64_t*>(ws + L.y) : nullptr,
                              P, G, ws + L.pre, ws + L.emb, ws + L.gt1, ws + L.gt2, ws + L.gt3,
                              ws + L.timescr, B, u->num_classes, st));
        break;
    }
  }
  // join: everything the side stream did is ordered before whatever follows on `stream`
  TDX_HIP(hipEventRecord(u->ev_join, u->side));
  TDX_HIP(hipStreamWaitEvent(st, u->ev_join, 0));
  return 0;
}

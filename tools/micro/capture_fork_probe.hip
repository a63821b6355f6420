// Stand-alone probe of hipStreamEndCapture with forked helper streams (ROCm 7.2, gfx950).
//
// Why: capturing one three-stream training step of libtdx crashed INSIDE hipStreamEndCapture
// (segmentation fault, gpurun_out/r2_tests11.log).  Each case below isolates one fork/join pattern the
// step uses; every case is a separate process (run_capture_probe.sh) so that a crash in one does not
// hide the others.  All of them are legal under the rules of stream capture: every event waited for by
// a capturing stream was recorded inside the same capture, and every forked stream is joined to the
// origin before EndCapture.
//
//   case 0  one fork, one join
//   case 1  the SAME helper stream forked from the origin twice (two origin-recorded events)
//   case 2  helper B forks from helper A (event recorded on a non-origin stream), both joined
//   case 3  one event re-recorded several times inside the capture (origin side)
//   case 4  events that were also recorded in an eager warm-up pass before the capture
//   case 5  cases 1 + 2 + 3 + 4 together, 3 streams, the shape of a libtdx step
//   case 6  case 1 with the helper re-joined to the origin BETWEEN the two forks
//   case 7  case 1 with a helper created at default priority / blocking flags
//   case 8  A and B both forked from the origin; then B waits for an event recorded on A; both joined
//   case 9  case 8 + A then waits for an event recorded on B (mutual cross-waits between two helpers)
//   case 10 case 9 thirteen times over with distinct events (the backward's wgrad / slab-reduce ping-pong)
//   case 11 ONE event forks B, is re-recorded on the origin, then forks A (libtdx forward: ev_fork)
//   case 12 the literal event sequence of one libtdx training step (forward + 15 backward stages + join)
//
//   hipcc --offload-arch=gfx950 -O2 -o capture_fork_probe capture_fork_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                                   \
  do {                                                                                          \
    hipError_t e__ = (x);                                                                       \
    if (e__ != hipSuccess) {                                                                    \
      printf("case %d: %s -> %s (line %d)\n", g_case, #x, hipGetErrorString(e__), __LINE__);    \
      fflush(stdout);                                                                           \
      return 2;                                                                                 \
    }                                                                                           \
  } while (0)

static int g_case = 0;
static int g_skip = 0;   // case 12: bit mask of parts left out (bisection, see body2)

__global__ void add_kernel(float* p, float v, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] += v;
}

static void launch(float* p, float v, int n, hipStream_t s) {
  add_kernel<<<(n + 255) / 256, 256, 0, s>>>(p, v, n);
}

int main(int argc, char** argv) {
  g_case = argc > 1 ? atoi(argv[1]) : 0;
  g_skip = argc > 2 ? atoi(argv[2]) : 0;
  const int n = 1 << 16;
  float *a, *b, *c;
  CK(hipMalloc(&a, n * sizeof(float)));
  CK(hipMalloc(&b, n * sizeof(float)));
  CK(hipMalloc(&c, n * sizeof(float)));
  CK(hipMemset(a, 0, n * sizeof(float)));
  CK(hipMemset(b, 0, n * sizeof(float)));
  CK(hipMemset(c, 0, n * sizeof(float)));
  int lo = 0, hi = 0;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t S, A, B;
  CK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking));   // origin (torch's capture stream is a pool stream)
  if (g_case == 7) {
    CK(hipStreamCreate(&A));
    CK(hipStreamCreate(&B));
  } else {
    CK(hipStreamCreateWithPriority(&A, hipStreamNonBlocking, lo));
    CK(hipStreamCreateWithPriority(&B, hipStreamNonBlocking, lo));
  }
  hipEvent_t ef, ef2, ea, eb, ej;
  CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&ef2, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&ea, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&eb, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));

  auto body = [&](bool capturing) -> int {
    (void)capturing;
    launch(a, 1.f, n, S);
    // fork 1
    CK(hipEventRecord(ef, S));
    CK(hipStreamWaitEvent(A, ef, 0));
    launch(b, 1.f, n, A);
    if (g_case == 2 || g_case == 5) {  // helper B forks from helper A
      CK(hipEventRecord(ea, A));
      CK(hipStreamWaitEvent(B, ea, 0));
      launch(c, 1.f, n, B);
      CK(hipEventRecord(eb, B));
    }
    launch(a, 1.f, n, S);
    if (g_case == 6) {  // join before forking again
      CK(hipEventRecord(ej, A));
      CK(hipStreamWaitEvent(S, ej, 0));
    }
    if (g_case == 1 || g_case == 5 || g_case == 6 || g_case == 7) {  // fork 2 of the same helper
      CK(hipEventRecord(g_case == 5 ? ef : ef2, S));                 // case 5: the same event re-recorded
      CK(hipStreamWaitEvent(A, g_case == 5 ? ef : ef2, 0));
      launch(b, 1.f, n, A);
    }
    if (g_case == 3) {  // re-record one event, each waited for once
      for (int k = 0; k < 3; ++k) {
        CK(hipEventRecord(ef, S));
        launch(a, 1.f, n, S);
      }
      CK(hipStreamWaitEvent(A, ef, 0));
      launch(b, 1.f, n, A);
    }
    // join
    if (g_case == 2 || g_case == 5) CK(hipStreamWaitEvent(S, eb, 0));
    CK(hipEventRecord(ej, A));
    CK(hipStreamWaitEvent(S, ej, 0));
    launch(a, 1.f, n, S);
    return 0;
  };

  hipEvent_t ev[64];
  for (int i = 0; i < 64; ++i) CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
  auto body2 = [&]() -> int {   // cases 8..12
    launch(a, 1.f, n, S);
    if (g_case == 11) {
      CK(hipEventRecord(ef, S));
      CK(hipStreamWaitEvent(B, ef, 0));
      launch(c, 1.f, n, B);
      launch(a, 1.f, n, S);
      CK(hipEventRecord(ef, S));
      CK(hipStreamWaitEvent(A, ef, 0));
      launch(b, 1.f, n, A);
    } else if (g_case == 12) {
      // forward: ev_fork -> side2 (time path); unit 0; ev_fork again -> side (packs) -> ev_pack; three skip
      // resizes forked to side2 (ev_s2_fork[k] / ev_s2_done[k]); the main stream waits ev_pack and the three done
      hipEvent_t ev_fork = ev[0], ev_pack = ev[1], ev_join = ev[2], ev_join2 = ev[3];
      hipEvent_t* s2f = ev + 4; hipEvent_t* s2d = ev + 7; hipEvent_t* dy = ev + 10; hipEvent_t* w = ev + 23;
      hipEvent_t* red = ev + 36;
      // g_skip bits: 1 no forward, 2 no waits of A for B's events (ev_red), 4 no waits of B for A's events (ev_w),
      // 8 no waits of the origin for old events of A (acquire), 16 no skip-branch forks in the backward,
      // 32 no eager warm-up, 64 only four units, 128 the origin's acquire waits name A's NEWEST event instead
      if (!(g_skip & 1)) {
      CK(hipEventRecord(ev_fork, S)); CK(hipStreamWaitEvent(B, ev_fork, 0)); launch(c, 1.f, n, B);
      launch(a, 1.f, n, S);
      CK(hipEventRecord(ev_fork, S)); CK(hipStreamWaitEvent(A, ev_fork, 0)); launch(b, 1.f, n, A);
      CK(hipEventRecord(ev_pack, A));
      for (int k = 0; k < 3; ++k) {
        if (k == 1) CK(hipStreamWaitEvent(S, ev_pack, 0));
        launch(a, 1.f, n, S);
        CK(hipEventRecord(s2f[k], S)); CK(hipStreamWaitEvent(B, s2f[k], 0)); launch(c, 1.f, n, B);
        CK(hipEventRecord(s2d[k], B));
        launch(a, 1.f, n, S);
      }
      for (int k = 0; k < 3; ++k) { launch(a, 1.f, n, S); CK(hipStreamWaitEvent(S, s2d[2 - k], 0)); launch(a, 1.f, n, S); }
      }
      // backward stage 0: final_conv wgrad on side
      CK(hipEventRecord(ev_fork, S)); CK(hipStreamWaitEvent(A, ev_fork, 0)); launch(b, 1.f, n, A);
      launch(a, 1.f, n, S);
      bool pending[13] = {false};
      for (int i = 12; i >= ((g_skip & 64) ? 9 : 0); --i) {   // unit_bwd(i)
        launch(a, 1.f, n, S);                                            // BN backward
        CK(hipEventRecord(dy[i], S)); CK(hipStreamWaitEvent(A, dy[i], 0));
        if (!(g_skip & 2) && i + 2 < 13 && pending[i + 2]) { CK(hipStreamWaitEvent(A, red[i + 2], 0)); pending[i + 2] = false; }
        launch(b, 1.f, n, A);                                            // wgrad
        CK(hipEventRecord(w[i], A)); if (!(g_skip & 4)) CK(hipStreamWaitEvent(B, w[i], 0));
        launch(c, 1.f, n, B);                                            // slab reduce
        CK(hipEventRecord(red[i], B)); pending[i] = true;
        if (!(g_skip & 8) && i + 4 < 13) CK(hipStreamWaitEvent(S, w[(g_skip & 128) ? i : i + 4], 0));          // acquire(): last reader of the LRU buffer
        launch(a, 1.f, n, S);                                            // dgrad
        if (!(g_skip & 16) && (i == 11 || i == 9 || i == 7)) {                               // decoder level: skip branch on side2
          const int k = (i - 7) / 2;
          CK(hipEventRecord(s2f[k], S)); CK(hipStreamWaitEvent(B, s2f[k], 0)); launch(c, 1.f, n, B);
          CK(hipEventRecord(s2d[k], B));
          launch(a, 1.f, n, S);
        }
        if (!(g_skip & 17) && (i == 6 || i == 4 || i == 2)) { CK(hipStreamWaitEvent(S, s2d[2 - (6 - i) / 2], 0)); launch(a, 1.f, n, S); }
      }
      CK(hipEventRecord(ev_fork, S)); CK(hipStreamWaitEvent(B, ev_fork, 0)); launch(c, 1.f, n, B);  // time path
      launch(a, 1.f, n, S);
      CK(hipEventRecord(ev_join, A)); CK(hipStreamWaitEvent(S, ev_join, 0));
      CK(hipEventRecord(ev_join2, B)); CK(hipStreamWaitEvent(S, ev_join2, 0));
      launch(a, 1.f, n, S);
      return 0;
    } else {
      CK(hipEventRecord(ef, S));
      CK(hipStreamWaitEvent(A, ef, 0));
      CK(hipEventRecord(ef2, S));
      CK(hipStreamWaitEvent(B, ef2, 0));
      const int rounds = g_case == 10 ? 13 : 1;
      for (int r = 0; r < rounds; ++r) {
        launch(b, 1.f, n, A);
        CK(hipEventRecord(ev[2 * r], A));
        CK(hipStreamWaitEvent(B, ev[2 * r], 0));
        launch(c, 1.f, n, B);
        if (g_case >= 9) {
          CK(hipEventRecord(ev[2 * r + 1], B));
          CK(hipStreamWaitEvent(A, ev[2 * r + 1], 0));
          launch(b, 1.f, n, A);
        }
        launch(a, 1.f, n, S);
      }
    }
    CK(hipEventRecord(ej, A));
    CK(hipStreamWaitEvent(S, ej, 0));
    CK(hipEventRecord(eb, B));
    CK(hipStreamWaitEvent(S, eb, 0));
    launch(a, 1.f, n, S);
    return 0;
  };
  if (g_case >= 8) {
    if (g_case == 12 && !(g_skip & 32)) { if (body2()) return 2; CK(hipDeviceSynchronize()); }   // eager warm-up, like the step
    hipGraph_t graph2;
    hipGraphExec_t exec2;
    CK(hipStreamBeginCapture(S, hipStreamCaptureModeGlobal));
    if (body2()) return 2;
    printf("case %d skip %d: ending capture ...\n", g_case, g_skip);
    fflush(stdout);
    CK(hipStreamEndCapture(S, &graph2));
    size_t nn2 = 0;
    CK(hipGraphGetNodes(graph2, nullptr, &nn2));
    CK(hipGraphInstantiate(&exec2, graph2, nullptr, nullptr, 0));
    for (int k = 0; k < 3; ++k) CK(hipGraphLaunch(exec2, S));
    CK(hipStreamSynchronize(S));
    printf("case %d skip %d: OK, %zu graph nodes\n", g_case, g_skip, nn2);
    return 0;
  }

  if (g_case == 4 || g_case == 5) {  // eager warm-up: the events carry an earlier (non-captured) record
    if (body(false)) return 2;
    CK(hipDeviceSynchronize());
  }
  hipGraph_t graph;
  hipGraphExec_t exec;
  CK(hipStreamBeginCapture(S, hipStreamCaptureModeGlobal));
  if (body(true)) return 2;
  printf("case %d: ending capture ...\n", g_case);
  fflush(stdout);
  CK(hipStreamEndCapture(S, &graph));
  size_t nn = 0;
  CK(hipGraphGetNodes(graph, nullptr, &nn));
  CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  for (int k = 0; k < 3; ++k) CK(hipGraphLaunch(exec, S));
  CK(hipStreamSynchronize(S));
  float ha = 0, hb = 0, hc = 0;
  CK(hipMemcpy(&ha, a, 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(&hb, b, 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(&hc, c, 4, hipMemcpyDeviceToHost));
  printf("case %d: OK, %zu graph nodes, a=%g b=%g c=%g\n", g_case, nn, ha, hb, hc);
  return 0;
}

// bf16 compute mode of the 3x3 convolutions (opt-in; BASELINE.json configs[3]/[4] name bf16, the
// reference itself is fp32-only - SURVEY.md 0): the same implicit GEMMs as conv3x3.hip on the
// bf16 matrix cores (v_mfma_f32_32x32x16_bf16, 16x the fp32 rate), fp32 accumulation.
//
// What stays fp32: every tensor in HBM (activations, gradients, weight-gradient slabs, parameters),
// the accumulators, the bias / BatchNorm-statistics epilogue, BatchNorm itself and the time MLP.
// What becomes bf16: the two MFMA operands, rounded to nearest-even while the tile is staged
// (activations: fp32 global load -> [BN+ReLU on load] -> v_cvt_pk_bf16_f32 -> LDS; weights: packed to
// bf16 once per step).  That is autocast-like arithmetic with fp32 outputs; its tolerance is stated
// in tests/test_gpu_bf16.py (eps_hat MSE <= 5e-4 against the fp32 goldens).
//
// At 16x the matrix rate these layers are HBM/L2-bound, not MFMA-bound: a K-tile is 64 channels of
// one tap (4 MFMA k-steps), tiles are 128 pixels x 128|64 output channels, the BN+ReLU of the
// producing layer is applied while staging (VALU is free here, so nothing is materialised), and the
// nine taps of a tile re-read their rows from L1/L2.
#include "conv_shared.h"
#include <type_traits>
#include <algorithm>

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define KT 64    // K-tile (bf16 elements): 64 channels of one tap / 64 pixels (wgrad)
#define KTP 72   // padded LDS row: 144 B, conflict-free for the 16-lane groups of ds_read_b128

// ---------------------------------------------------------------------------------- forward / dgrad
// IO16 (bf16 STORAGE, round 3): the input tensor holds bf16 - a 64-channel K-tile row is 128 bytes, eight 16-byte
// loads instead of sixteen, and a raw input goes from the load registers to LDS with no conversion at all - and
// the output is rounded to bf16 by the epilogue (ConvArgs::out_bf16; the BatchNorm statistics are still taken from
// the fp32 accumulators).
template <int BM, int BN, bool IN_BN, int EPI, bool IO16>
__global__ void __launch_bounds__(256)
conv3x3_bf16_kernel(ConvArgs a) {
  constexpr int WGN = 2;
  constexpr int WTM = BM / 2, WTN = BN / 2;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int ACH = IO16 ? 8 : 16;   // 16-byte loads per 64-channel row: 8 bf16 | 4 fp32 each
  constexpr int ARW = 256 / ACH;       // rows covered by one pass of the workgroup
  constexpr int AEL = 64 / ACH;        // channels per load
  constexpr int AES = IO16 ? 2 : 4;    // bytes per element of the input tensor
  constexpr int AI = BM / ARW;         // A loads per thread per K-tile
  constexpr int BI = BN / 32;  // 16-byte bf16 loads per thread per K-tile (8 per row)

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* As = reinterpret_cast<__bf16*>(smem_raw);  // [2][BM][KTP]
  __bf16* Bs = As + 2 * BM * KTP;                    // [2][BN][KTP]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wave / WGN, wn = wave % WGN;
  // XCD-aware tile ids, as in conv3x3_igemm_kernel: the column tiles of one row tile share an XCD
  const int xb = blockIdx.x / (8 * a.tilesN), xr = blockIdx.x % (8 * a.tilesN);
  const int tile_m = xb * 8 + (xr & 7), tile_n = xr >> 3;
  if (tile_m * BM >= a.M) return;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int HW = a.H * a.W;

  const int a_c4 = tid % ACH, a_row = tid / ACH;  // A: ACH 16-byte chunks per row, ARW rows per pass
  const int b_ch = tid & 7, b_row = tid >> 3;   // B: 8 x 16 B per row, 32 rows per pass
  const int neg = (a.W + 1) * a.Cin;
  const auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(const_cast<float*>(a.in)) - (int64_t)neg * AES, 0,
      (int)(((int64_t)a.M * a.Cin + 2 * neg) * AES), 0x00020000);
  const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0,
                                                        a.Cout * 9 * a.Cin * 2, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  unsigned a_off[AI], a_taps[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int p = m0 + a_row + ARW * i;
    unsigned taps = 0;
    if (p < a.M) {
      const int r = p % HW, oh = r / a.W, ow = r % a.W;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int ih = oh + t / 3 - 1, iw = ow + t % 3 - 1;
        if ((unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W) taps |= 1u << t;
      }
    }
    a_taps[i] = taps;
    a_off[i] = (unsigned)(p * a.Cin + a_c4 * AEL) * (unsigned)AES;
  }
  unsigned w_off[BI];
#pragma unroll
  for (int j = 0; j < BI; ++j) w_off[j] = (unsigned)((n0 + b_row + 32 * j) * 9 * a.Cin + b_ch * 8) * 2u;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int im = 0; im < TM; ++im)
#pragma unroll
    for (int in = 0; in < TN; ++in)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[im][in][r] = 0.f;

  const int nk = 9 * (a.Cin / KT);
  // Two register stages: at 16x the fp32 matrix rate one K-tile is 16 MFMAs (~0.2 us) per wave while
  // a global load takes 1-2 us, so the loop is LATENCY-bound unless a tile's loads are issued a whole
  // iteration before they are needed (measured with one stage: 190 us for a layer whose HBM time is 40).
  // Iteration kt: issue the loads of tile kt+2 into the free stage, run the MFMAs of tile kt from
  // LDS[kt&1], convert + store tile kt+1 (loaded during iteration kt-1) into LDS[(kt+1)&1].
  struct Stage {
    f32x4 ra[AI], rb[BI], sc, sh, sc2, sh2;   // (sc2 / sh2: channels 4..7 of an 8-channel bf16 chunk)
    unsigned ok;
  };
  Stage S0, S1;
  auto load_tile = [&](int kt, Stage& S) {
    const int kn = min(kt, nk - 1);
    const int cblk = kn / 9, tap = kn - cblk * 9;
    const unsigned soff_in = (unsigned)(((tap / 3) * a.W + (tap % 3)) * a.Cin + cblk * KT) * (unsigned)AES;
    const unsigned soff_w = (unsigned)(tap * a.Cin + cblk * KT) * 2u;
#pragma unroll
    for (int j = 0; j < BI; ++j)
      S.rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_off[j], soff_w, 0));
    if (IN_BN) {
      S.sc = *reinterpret_cast<const f32x4*>(a.in_scale + cblk * KT + a_c4 * AEL);
      S.sh = *reinterpret_cast<const f32x4*>(a.in_shift + cblk * KT + a_c4 * AEL);
      if (IO16) {
        S.sc2 = *reinterpret_cast<const f32x4*>(a.in_scale + cblk * KT + a_c4 * AEL + 4);
        S.sh2 = *reinterpret_cast<const f32x4*>(a.in_shift + cblk * KT + a_c4 * AEL + 4);
      }
    }
    unsigned ok = 0;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const bool v = (a_taps[i] >> tap) & 1u;
      if (IN_BN) ok |= v ? (1u << i) : 0u;
      S.ra[i] = __builtin_bit_cast(
          f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, v ? a_off[i] : OOB, soff_in, 0));
    }
    S.ok = ok;
  };
  // piece 0 .. AI-1: one 16-byte (bf16 storage) / 8-byte row chunk of the input tile; AI .. AI+BI-1: of the weights
  auto store_piece = [&](const Stage& S, int buf, int piece) {
    __bf16* Ab = As + buf * BM * KTP;
    __bf16* Bb = Bs + buf * BN * KTP;
    if (piece < AI) {
      const int i = piece;
      if (IO16) {
        f32x4 raw = S.ra[i];   // eight bf16
        if (IN_BN) {
          const bf16x8 h = __builtin_bit_cast(bf16x8, raw);
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[e] = (__bf16)fmaxf(fmaf((float)h[e], S.sc[e], S.sh[e]), 0.f);
            o[e + 4] = (__bf16)fmaxf(fmaf((float)h[e + 4], S.sc2[e], S.sh2[e]), 0.f);
          }
          raw = __builtin_bit_cast(f32x4, o);
          if (!((S.ok >> i) & 1u)) raw = f32x4{0.f, 0.f, 0.f, 0.f};   // padding stays 0 AFTER the transform
        }
        *reinterpret_cast<f32x4*>(Ab + (a_row + ARW * i) * KTP + a_c4 * 8) = raw;
      } else {
        f32x4 v = S.ra[i];
        if (IN_BN) {
          // padding must stay 0 AFTER the transform (relu(shift) != 0)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(v[e], S.sc[e], S.sh[e]), 0.f);
          if (!((S.ok >> i) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        *reinterpret_cast<bf16x4*>(Ab + (a_row + ARW * i) * KTP + a_c4 * 4) = __builtin_convertvector(v, bf16x4);
      }
    } else {
      const int j = piece - AI;
      *reinterpret_cast<f32x4*>(Bb + (b_row + 32 * j) * KTP + b_ch * 8) = S.rb[j];
    }
  };
  auto store_tile = [&](const Stage& S, int buf) {
#pragma unroll
    for (int piece = 0; piece < AI + BI; ++piece) store_piece(S, buf, piece);
  };
  // Input-gradient launches (raw input, plain epilogue): the MFMAs of tile kt with the stores of tile kt+1 spread between
  // them, as in conv3x3_wgrad_bf16s_kernel - 931 -> 862 us over the MNIST layers.  ST is a compile-time flag (a branch
  // around every store would split the schedule).  The forward launches keep the phases apart: the same interleaving made
  // them SLOWER (1072 -> 1255-1324 us, with and without the BN + ReLU transform between the MFMAs).  (A third register
  // stage - the loads of tile kt+3 in flight too - measured no faster: the loop is not waiting for its loads.)
  constexpr bool IL = !IN_BN && EPI == EPI_PLAIN;
  constexpr int NM = 4 * TM * TN, NP = AI + BI;
  auto iteration_il = [&](int kt, const Stage& Sst, Stage& Sld, auto store) {
    constexpr bool ST = decltype(store)::value;
    const int cur = kt & 1;
    if (ST) load_tile(kt + 2, Sld);
    __builtin_amdgcn_sched_barrier(0);      // keep the loads at the top: hipcc otherwise sinks them
    const __bf16* Ab = As + cur * BM * KTP + (wm * WTM + l31) * KTP + half * 8;
    const __bf16* Bb = Bs + cur * BN * KTP + (wn * WTN + l31) * KTP + half * 8;
#pragma unroll
    for (int ks = 0; ks < KT / 16; ++ks) {
      bf16x8 af[TM], bf[TN];
#pragma unroll
      for (int im = 0; im < TM; ++im) af[im] = *reinterpret_cast<const bf16x8*>(Ab + im * 32 * KTP + ks * 16);
#pragma unroll
      for (int in = 0; in < TN; ++in) bf[in] = *reinterpret_cast<const bf16x8*>(Bb + in * 32 * KTP + ks * 16);
#pragma unroll
      for (int im = 0; im < TM; ++im)
#pragma unroll
        for (int in = 0; in < TN; ++in) {
          acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[im], bf[in], acc[im][in], 0, 0, 0);
          if (ST) {
            const int m = (ks * TM + im) * TN + in;   // store piece p behind MFMA m when p * NM / NP == m
#pragma unroll
            for (int pc = 0; pc < NP; ++pc)
              if (pc * NM / NP == m) store_piece(Sst, cur ^ 1, pc);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();
  };
  auto compute = [&](int buf) {
    const __bf16* Ab = As + buf * BM * KTP + (wm * WTM + l31) * KTP + half * 8;
    const __bf16* Bb = Bs + buf * BN * KTP + (wn * WTN + l31) * KTP + half * 8;
#pragma unroll
    for (int ks = 0; ks < KT / 16; ++ks) {
      bf16x8 af[TM], bf[TN];
#pragma unroll
      for (int im = 0; im < TM; ++im) af[im] = *reinterpret_cast<const bf16x8*>(Ab + im * 32 * KTP + ks * 16);
#pragma unroll
      for (int in = 0; in < TN; ++in) bf[in] = *reinterpret_cast<const bf16x8*>(Bb + in * 32 * KTP + ks * 16);
#pragma unroll
      for (int im = 0; im < TM; ++im)
#pragma unroll
        for (int in = 0; in < TN; ++in)
          acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[im], bf[in], acc[im][in], 0, 0, 0);
    }
  };
  auto iteration = [&](int kt, const Stage& Sst, Stage& Sld) {   // (a.dbg: ablation bits of tools/gpu_bf16_layers.py)
    const int cur = kt & 1;
    if (!(a.dbg & 64)) load_tile(kt + 2, Sld);   // (clamped at the end: a redundant re-load, never stored)
    __builtin_amdgcn_sched_barrier(0);      // keep the loads at the top: hipcc otherwise sinks them
    if (!(a.dbg & 16)) compute(cur);
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < nk && !(a.dbg & 32)) store_tile(Sst, cur ^ 1);
    __syncthreads();
  };
  load_tile(0, S0);
  load_tile(1, S1);
  store_tile(S0, 0);
  __syncthreads();
  if constexpr (IL) {
    int kt = 0;
    for (; kt + 2 < nk; kt += 2) {
      iteration_il(kt, S1, S0, std::true_type{});       // S1 holds tile kt+1; tile kt+2 -> S0
      iteration_il(kt + 1, S0, S1, std::true_type{});   // S0 holds tile kt+2; tile kt+3 -> S1
    }
    if (kt + 2 == nk) {
      iteration_il(kt, S1, S0, std::true_type{});
      iteration_il(kt + 1, S0, S1, std::false_type{});
    } else if (kt < nk) {
      iteration_il(kt, S1, S0, std::false_type{});
    }
  } else {
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
      iteration(kt, S1, S0);
      iteration(kt + 1, S0, S1);
    }
    if (kt < nk) iteration(kt, S1, S0);
  }
  conv_epilogue<BM, BN, EPI, false>(a, acc, reinterpret_cast<float*>(smem_raw), tile_m, m0, n0, wm, wn, l31, half,
                                    tid);
}

// ------------------------------------------------------------- forward / input gradient, ring (round 3)
// The bf16-storage forward and input-gradient GEMM for row counts that are a multiple of 256.  The kernel above keeps
// the shape of the fp32 one (128 x 128 tile, four waves, two workgroups per CU, register staging); at the bf16 matrix
// rate that shape is bound by everything except the matrix core - per K-tile and CU the loads take the L1 1024 cycles
// (64 KB at 64 B/clk), the ds_write_b128 staging another ~800, the operand reads 512, the MFMAs 1024, and the ablation
// bits showed the four add up instead of overlapping (tools/gpu_bf16_layers.py: 3300 cycles per K-tile).  Here:
//   * 256 x BN tile, eight waves (4 x 2, 64 x BN/2 each), one workgroup per CU: 48 KB of operands per K-tile for
//     twice the MFMAs of the old tile (87 FLOP per byte from L1 instead of 64);
//   * weights and raw inputs go global -> LDS by DMA (buffer_load ... lds, 16 B per lane): no staging registers, no
//     ds_write; a BN+ReLU input still passes through registers (16-byte loads, transform, ds_write_b128);
//   * three LDS stages: the DMA of tile kt+2 is issued when tile kt starts, two K-tiles of MFMAs ahead of its use, and
//     ONE barrier per K-tile - a counted s_waitcnt vmcnt in front of a raw s_barrier, because the workgroup-release
//     fence inside __syncthreads() drains every outstanding DMA;
//   * 128-byte rows, 16-byte chunk index XOR (row >> 1) & 7: the lane-linear DMA image and the ds_read_b128 operand
//     reads are both conflict-free.
// The epilogue is conv_epilogue<128, BN> run by each half of the workgroup on its 128 rows (statistics partials per 128
// rows, as bn_finalize expects).
template <int BN, bool IN_BN, int EPI>
__global__ void __launch_bounds__(512)
conv3x3_bf16_ring_kernel(ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)   // (address-space casts and s_waitcnt asm: device pass only)
  constexpr int BM = 256, NST = 3;
  constexpr int TM = 2, TN = BN / 64, WTN = BN / 2;
  constexpr int BJ = BN / 64;                      // weight DMA instructions per wave and K-tile (8 rows x 128 B each)
  constexpr int A_ST = BM * 64, B_ST = BN * 64;    // elements per stage
  constexpr int STAGE = A_ST + B_ST;
  constexpr int NV = IN_BN ? BJ + 4 : BJ + 4;      // VMEM instructions per thread and K-tile (DMA + register loads)
  typedef __attribute__((address_space(3))) void* lds_ptr_t;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* ring = reinterpret_cast<__bf16*>(smem_raw);                       // [NST][A: 256 x 64 | B: BN x 64]
  float* ssc = reinterpret_cast<float*>(smem_raw + NST * STAGE * 2);       // IN_BN: scale[Cin] | shift[Cin]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int xb = blockIdx.x / (8 * a.tilesN), xr = blockIdx.x % (8 * a.tilesN);
  const int tile_m = xb * 8 + (xr & 7), tile_n = xr >> 3;
  if (tile_m * BM >= a.M) return;   // (the launcher guarantees M % 256 == 0: no ragged tile)
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int HW = a.H * a.W;

  const int neg = (a.W + 1) * a.Cin;
  const auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(const_cast<float*>(a.in)) - (int64_t)neg * 2, 0,
      (int)(((int64_t)a.M * a.Cin + 2 * neg) * 2), 0x00020000);
  const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.Cout * 9 * a.Cin * 2, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;

  // ---- staging maps.  DMA: instruction i of wave w covers tile rows (i*8 + w)*8 .. +7, lane L -> row + (L >> 3),
  // stored chunk L & 7, which holds logical chunk (L & 7) ^ x(row).  Register path (IN_BN): thread -> logical chunk
  // tid & 7 of rows (tid >> 3) + 64 i.
  const int drow = lane >> 3, dch = lane & 7;
  unsigned a_off[4], a_taps[4];
  int a_lds[4];   // IN_BN: byte offset of the thread's 16-byte store inside the A stage
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = IN_BN ? (tid >> 3) + 64 * i : (i * 8 + wave) * 8 + drow;
    const int x = (r >> 1) & 7;
    const int p = m0 + r;
    const int rr = p % HW, oh = rr / a.W, ow = rr % a.W;
    unsigned taps = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int ih = oh + t / 3 - 1, iw = ow + t % 3 - 1;
      if ((unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W) taps |= 1u << t;
    }
    a_taps[i] = taps;
    const int chunk = IN_BN ? (tid & 7) : (dch ^ x);
    a_off[i] = (unsigned)(p * a.Cin + chunk * 8) * 2u;
    a_lds[i] = r * 128 + (((tid & 7) ^ x) << 4);
  }
  unsigned w_off[BJ];
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    const int r = (j * 8 + wave) * 8 + drow;
    w_off[j] = (unsigned)((n0 + r) * 9 * a.Cin + ((dch ^ ((r >> 1) & 7)) << 3)) * 2u;
  }
  if (IN_BN) {
    for (int c = tid; c < a.Cin; c += 512) {
      ssc[c] = a.in_scale[c];
      ssc[a.Cin + c] = a.in_shift[c];
    }
  }

  // ---- operand reads: row wm*64 + im*32 + l31 (A) / wn*WTN + in*32 + l31 (B), chunk (2 ks + half) ^ x, x from l31 only
  const int x31 = (l31 >> 1) & 7;
  const int a_rd = (wm * 64 + l31) * 128, b_rd = A_ST * 2 + (wn * WTN + l31) * 128;   // bytes inside a stage
  int frag_pos[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) frag_pos[ks] = ((2 * ks + half) ^ x31) << 4;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int im = 0; im < TM; ++im)
#pragma unroll
    for (int in = 0; in < TN; ++in)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[im][in][r] = 0.f;

  const int nk = 9 * (a.Cin / KT);
  struct Regs {
    u32x4 ra[4];
    unsigned ok;
    int cb;   // channel block of the tile (scale / shift look-up)
  };
  Regs R0, R1;
  // every call issues exactly NV VMEM instructions per thread (requests past the last tile repeat it and are never read)
  auto issue = [&](int kt, int st, Regs& R) {
    const int kn = min(kt, nk - 1);
    const int cblk = kn / 9, tap = kn - cblk * 9;
    const unsigned soff_in = (unsigned)(((tap / 3) * a.W + (tap % 3)) * a.Cin + cblk * KT) * 2u;
    const unsigned soff_w = (unsigned)(tap * a.Cin + cblk * KT) * 2u;
    unsigned char* sb = smem_raw + st * (STAGE * 2);
#pragma unroll
    for (int j = 0; j < BJ; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(sb + A_ST * 2 + (j * 8 + wave) * 1024), 16, w_off[j], soff_w, 0, 0);
    unsigned ok = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool v = (a_taps[i] >> tap) & 1u;
      if (IN_BN) {
        ok |= v ? (1u << i) : 0u;
        R.ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, v ? a_off[i] : OOB, soff_in, 0);
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_in, (lds_ptr_t)(sb + (i * 8 + wave) * 1024), 16, v ? a_off[i] : OOB, soff_in, 0, 0);
      }
    }
    R.ok = ok;
    R.cb = cblk;
  };
  // IN_BN: piece i of a tile held in registers: BN + ReLU on eight channels, padding stays 0 AFTER the transform
  auto put_piece = [&](const Regs& R, int st, int i) {
    const float* sc = ssc + R.cb * KT + (tid & 7) * 8;
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(sc), s1 = *reinterpret_cast<const f32x4*>(sc + 4);
    const f32x4 h0 = *reinterpret_cast<const f32x4*>(sc + a.Cin), h1 = *reinterpret_cast<const f32x4*>(sc + a.Cin + 4);
    const bf16x8 h = __builtin_bit_cast(bf16x8, R.ra[i]);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = (__bf16)fmaxf(fmaf((float)h[e], s0[e], h0[e]), 0.f);
      o[e + 4] = (__bf16)fmaxf(fmaf((float)h[e + 4], s1[e], h1[e]), 0.f);
    }
    u32x4 raw = __builtin_bit_cast(u32x4, o);
    if (!((R.ok >> i) & 1u)) raw = u32x4{0u, 0u, 0u, 0u};
    *reinterpret_cast<u32x4*>(smem_raw + st * (STAGE * 2) + a_lds[i]) = raw;
  };
  bf16x8 af[2][TM], bf[2][TN];
  auto read_frags = [&](int st, int ks, int slot) {
    const unsigned char* sb = smem_raw + st * (STAGE * 2);
#pragma unroll
    for (int im = 0; im < TM; ++im) af[slot][im] = *reinterpret_cast<const bf16x8*>(sb + a_rd + im * 32 * 128 + frag_pos[ks]);
#pragma unroll
    for (int in = 0; in < TN; ++in) bf[slot][in] = *reinterpret_cast<const bf16x8*>(sb + b_rd + in * 32 * 128 + frag_pos[ks]);
  };
  // top of iteration kt: this thread's loads of tile kt have landed when only the NV of tile kt+1 are outstanding; the
  // barrier then says the same of every wave, and that all of them are done with the stage tile kt+2 goes to
#define TDX_RING_BARRIER() asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NV) : "memory")
  auto iteration = [&](int kt, int st, Regs& Rnext, Regs& Rfree) {
    TDX_RING_BARRIER();
    int st2 = st + 2; st2 = st2 >= NST ? st2 - NST : st2;
    int st1 = st + 1; st1 = st1 >= NST ? st1 - NST : st1;
    if (!(a.dbg & 64)) issue(kt + 2, st2, Rfree);   // (ablation bits, tools/gpu_bf16_layers.py: 64 no loads, 16 no MFMAs)
    __builtin_amdgcn_sched_barrier(0);
    if (a.dbg & 16) return;
    read_frags(st, 0, 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks + 1 < 4) {
        read_frags(st, ks + 1, (ks + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);   // the reads of k-step ks+1 are issued before the MFMAs of k-step ks
      }
#pragma unroll
      for (int im = 0; im < TM; ++im)
#pragma unroll
        for (int in = 0; in < TN; ++in)
          acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][im], bf[ks & 1][in], acc[im][in], 0, 0, 0);
      if (IN_BN) put_piece(Rnext, st1, ks);   // tile kt+1 (loaded an iteration ago) -> its stage, behind this k-step's MFMAs
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  issue(0, 0, R0);
  issue(1, 1, R1);
  if (IN_BN) {
    __syncthreads();   // the scale / shift copy
#pragma unroll
    for (int i = 0; i < 4; ++i) put_piece(R0, 0, i);
  }
  {
    int st = 0;
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
      iteration(kt, st, R1, R0);       // tile kt+1 is in R1; tile kt+2 -> R0
      st = st + 1 == NST ? 0 : st + 1;
      iteration(kt + 1, st, R0, R1);
      st = st + 1 == NST ? 0 : st + 1;
    }
    if (kt < nk) iteration(kt, st, R1, R0);   // (nk = 9 Cin / 64 is odd for odd Cin / 64)
  }
#undef TDX_RING_BARRIER
  // the tail requests are still in flight towards LDS: drain them before the epilogue re-uses the tile buffers
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (a.dbg & 128) return;   // (ablation: no epilogue)
  const int h = wm >> 1;   // which 128-row half of the tile this wave belongs to
  conv_epilogue<128, BN, EPI, false>(a, acc, reinterpret_cast<float*>(smem_raw + h * 40960), tile_m * 2 + h, m0 + 128 * h, n0,
                                     wm & 1, wn, l31, half, tid & 255);
#endif
}

// ------------------------------------------------- forward / input gradient, thin layers (round 3)
// 64 output channels per workgroup.  Written for the layers at the top of the UNets (64 -> 64 at 64 x 64, 192 -> 64,
// 256 -> 64 ...), where the kernels above sit at 5-7x their HBM time - a K-tile is ONE tap, so every input row crosses
// the L1 nine times (24 KB of operands per 16 MFMAs of a wave) - and then measured faster on every other layer too
// (knob bf16_thin): it is the bf16-storage mode's forward / input-gradient kernel wherever the rows are a multiple of 256.  Same idea as conv3x3_wgrad9_bf16_kernel: in slot space (padded
// (H+1) x (W+1) images, one index) a tap is a row offset, so the input WINDOW of a 256-pixel tile - its slots plus a
// halo of W + 2 on either side, <= 448 rows of 128 bytes - is fetched ONCE per 64-channel block by LDS-DMA and all nine
// taps read it (row reads: each lane supplies the row of its own pixel + the tap's offset; pad slots are zero rows).
// Only the 8 KB weight tile of a tap streams (three DMA stages, counted vmcnt + raw barrier as in the ring kernel).
// 132 KB through the L1 per 256 pixels instead of 432.  Eight waves (4 x 2, 64 rows x 32 columns each), 80 KB of LDS:
// two workgroups per CU.  Raw inputs only (bf16 storage mode materialises BN + ReLU).  Epilogue: conv_epilogue<128, 64>
// by each half of the workgroup, as in the ring kernel.
#define THIN_WROWS 448
template <int EPI>
__global__ void __launch_bounds__(512)
conv3x3_bf16_thin_kernel(ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 256, BN = 64;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char* Awin = smem_raw;                          // [THIN_WROWS][128 B]
  unsigned char* Bst = smem_raw + THIN_WROWS * 128;        // [3][64][128 B]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int xb = blockIdx.x / (8 * a.tilesN), xr = blockIdx.x % (8 * a.tilesN);
  const int tile_m = xb * 8 + (xr & 7), tile_n = xr >> 3;
  if (tile_m * BM >= a.M) return;   // (M % 256 == 0: no ragged tile)
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int HW = a.H * a.W, PW = a.W + 1, PH = a.H + 1, PP = PH * PW, nB = a.M / HW;
  auto slot_of = [&](int p) { const int n = p / HW, r = p - n * HW; const int oh = r / a.W; return n * PP + oh * PW + (r - oh * a.W); };
  const int HALO = PW + 1;
  const int wbase = slot_of(m0) - HALO;                       // slot of window row 0 (may be negative)
  const int NR = slot_of(m0 + BM - 1) - slot_of(m0) + 1 + 2 * HALO;   // rows in use (<= THIN_WROWS: the launcher checks)

  const auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, (int)((int64_t)a.M * a.Cin * 2), 0x00020000);
  const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, a.Cout * 9 * a.Cin * 2, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;

  // ---- window fill: instruction i of wave w covers window rows (i*8 + w)*8 .. +7; lane -> row + (lane >> 3), stored
  // chunk lane & 7 = logical chunk ^ x(row), x(r) = (r >> 1) & 7.  The slot of a row advances by 64 from i to i + 1.
  struct Pos { int n, oh, ow; };
  const int drow = lane >> 3, dch = lane & 7;
  Pos w0;
  {
    const int s8 = wbase + wave * 8 + drow + 8 * PP;   // (+ 8 samples: the halo in front of the tensor is negative)
    w0.n = s8 / PP - 8; const int r = s8 % PP; w0.oh = r / PW; w0.ow = r - w0.oh * PW;
  }
  const int dn = 64 / PP, d_oh = (64 % PP) / PW, d_ow = (64 % PP) % PW;
  auto fill_window = [&](int cblk) {
    Pos q = w0;
#pragma unroll
    for (int i = 0; i < THIN_WROWS / 64; ++i) {
      const int r = (i * 8 + wave) * 8 + drow;
      const bool real = (r < NR) & ((unsigned)q.n < (unsigned)nB) & (q.oh < a.H) & (q.ow < a.W);
      const int p = (q.n * a.H + q.oh) * a.W + q.ow;
      const unsigned off = (unsigned)p * (unsigned)a.Cin * 2u + (unsigned)(cblk * KT + ((dch ^ ((r >> 1) & 7)) << 3)) * 2u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_in, (lds_ptr_t)(Awin + (i * 8 + wave) * 1024), 16, real ? off : OOB, 0, 0, 0);
      int ow = q.ow + d_ow; const bool c = ow >= PW; ow = c ? ow - PW : ow;
      int oh = q.oh + d_oh + (c ? 1 : 0); const bool c2 = oh >= PH; oh = c2 ? oh - PH : oh;
      q.ow = ow; q.oh = oh; q.n += dn + (c2 ? 1 : 0);
    }
  };
  // weight tile of K-step k = (channel block, tap): 64 rows x 128 B, one DMA instruction per wave
  const int brow = wave * 8 + drow;
  const unsigned w_off = (unsigned)((n0 + brow) * 9 * a.Cin + ((dch ^ ((brow >> 1) & 7)) << 3)) * 2u;
  const int nk = 9 * (a.Cin / KT);
  auto issue_B = [&](int k, int st) {
    const int kn = min(k, nk - 1);   // (requests past the end repeat the last tile into a stage nobody reads any more)
    const int cblk = kn / 9, tap = kn - cblk * 9;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(Bst + st * 8192 + wave * 1024), 16, w_off,
                                             (unsigned)(tap * a.Cin + cblk * KT) * 2u, 0, 0);
  };

  // ---- operand reads.  A: window row of this lane's pixel (centre tap) per 32-row block; B: row wn*32 + l31
  int wr[2];
#pragma unroll
  for (int im = 0; im < 2; ++im) wr[im] = slot_of(m0 + wm * 64 + im * 32 + l31) - wbase;
  const int x31 = (l31 >> 1) & 7;   // ((wn*32 + l31) >> 1) & 7
  const int b_rd = (wn * 32 + l31) * 128;

  f32x16 acc[2][1];
#pragma unroll
  for (int im = 0; im < 2; ++im)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[im][0][r] = 0.f;

  fill_window(0);
  issue_B(0, 0);
  issue_B(1, 1);
  int st = 0;
  for (int k = 0; k < nk; ++k) {
    const int cblk = k / 9, tap = k - cblk * 9;
    if (tap == 0) {
      if (k > 0) {   // next channel block: everybody is done with the old window, then refill it
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        fill_window(cblk);
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
      // weight tile k has landed when only tile k+1's instruction is outstanding; the barrier says so for every wave
      asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    int st2 = st + 2; st2 = st2 >= 3 ? st2 - 3 : st2;
    issue_B(k + 2, st2);
    const int sh = (tap / 3 - 1) * PW + (tap % 3 - 1);
    const unsigned char* Bb = Bst + st * 8192 + b_rd;
    int arow[2], ax[2];
#pragma unroll
    for (int im = 0; im < 2; ++im) { arow[im] = (wr[im] + sh) * 128; ax[im] = ((wr[im] + sh) >> 1) & 7; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(Bb + (((2 * ks + half) ^ x31) << 4));
#pragma unroll
      for (int im = 0; im < 2; ++im) {
        const bf16x8 afr = *reinterpret_cast<const bf16x8*>(Awin + arow[im] + (((2 * ks + half) ^ ax[im]) << 4));
        acc[im][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, bfr, acc[im][0], 0, 0, 0);
      }
    }
    st = st + 1 == 3 ? 0 : st + 1;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // tail requests still in flight towards LDS
  const int h = wm >> 1;
  conv_epilogue<128, BN, EPI, false>(a, acc, reinterpret_cast<float*>(smem_raw + h * 24576), tile_m * 2 + h, m0 + 128 * h, n0,
                                     wm & 1, wn, l31, half, tid & 255);
#endif
}

// --------------------------------------------------------------------------------------------- wgrad
// dW[co][tap][ci] = sum_p dy[p][co] * in[p + tap][ci].  Both operands are pixel-major in memory while
// the MFMA wants 8 consecutive k (pixels) of one row (channel) per lane, so the tiles are transposed
// while they are staged: a thread loads 4 pixels x 4 channels (four float4), and writes, per channel,
// the four pixels as one 8-byte bf16 pack into LDS[channel][pixel].  K-tile = 64 pixels.
// IO16: `in` and `dy` hold bf16 (8-byte loads of 4 channels; the transposition moves the 16-bit elements as they are)
template <int BM, int BN, bool IN_BN, bool IO16>
__global__ void __launch_bounds__(256, 2)   // two workgroups per CU (<= 256 registers): latency hiding first
conv3x3_wgrad_bf16_kernel(WgradArgs a) {
  constexpr unsigned ES = IO16 ? 2u : 4u;   // bytes per element of `in` / `dy`
  constexpr int WGN = 2;
  constexpr int WTM = BM / 2, WTN = BN / 2;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int ACH = BM / 4, BCH = BN / 4;            // float4 columns per pixel row
  constexpr int APG = 256 / ACH, BPG = 256 / BCH;      // 4-pixel groups handled per pass
  constexpr int APASS = KT / (4 * APG), BPASS = KT / (4 * BPG);
  static_assert(APASS >= 1 && BPASS >= 1, "tile too narrow for 256 threads");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* As = reinterpret_cast<__bf16*>(smem_raw);  // [2][BM][KTP]  dy^T
  __bf16* Bs = As + 2 * BM * KTP;                    // [2][BN][KTP]  in^T (shifted by the tap)

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wave / WGN, wn = wave % WGN;

  // (pixel chunk, co tile, ci tile, tap) from the workgroup id: the nine taps of a group share an XCD
  const int L = blockIdx.x;
  const int g = (L / 72) * 8 + (L % 8);
  const int tap = (L % 72) / 8;
  if (g >= a.groups) return;
  const int tiles = a.tilesCo * a.tilesCi;
  const int split = g / tiles, tl = g % tiles;
  const int tile_co = tl / a.tilesCi, tile_ci = tl % a.tilesCi;
  const int co0 = tile_co * BM, ci0 = tile_ci * BN;
  const int dh = tap / 3 - 1, dw = tap % 3 - 1;
  const int HW = a.H * a.W;
  const int p_lo = split * a.chunk;
  const int p_hi = min(p_lo + a.chunk, a.M);
  const int nk = (p_hi - p_lo + KT - 1) / KT;

  const int a_c4 = tid % ACH, a_pg = tid / ACH;
  const int b_c4 = tid % BCH, b_pg = tid / BCH;
  constexpr unsigned OOB = 0x80000000u;
  const int neg = (a.W + 1) * a.Cin;
  // the dy descriptor ENDS at this workgroup's last pixel: the ragged end of the range reads zeros
  const auto rsrc_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0,
                                                         (int)((int64_t)p_hi * a.Cout * ES), 0x00020000);
  const auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(const_cast<float*>(a.in)) - (int64_t)neg * ES, 0,
      (int)(((int64_t)a.M * a.Cin + 2 * neg) * ES), 0x00020000);
  const unsigned tap_shift = (unsigned)(((tap / 3) * a.W + (tap % 3)) * a.Cin) * ES;
  unsigned a_off[APASS], b_off[BPASS];
#pragma unroll
  for (int q = 0; q < APASS; ++q) a_off[q] = (unsigned)((q * 4 * APG + a_pg * 4) * a.Cout + co0 + a_c4 * 4) * ES;
#pragma unroll
  for (int q = 0; q < BPASS; ++q) b_off[q] = (unsigned)((q * 4 * BPG + b_pg * 4) * a.Cin + ci0 + b_c4 * 4) * ES;
  // (oh, ow) of the first pixel of each 4-pixel group this thread stages for B; advanced by KT pixels
  // per K-tile (tiles are requested strictly in order)
  int b_oh[BPASS], b_ow[BPASS];
#pragma unroll
  for (int q = 0; q < BPASS; ++q) {
    const int r = (p_lo + q * 4 * BPG + b_pg * 4) % HW;
    b_oh[q] = r / a.W;
    b_ow[q] = r % a.W;
  }
  const int adv_q = KT / a.W, adv_s = KT % a.W;
  f32x4 sc4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f};
  if (IN_BN) {
    sc4 = *reinterpret_cast<const f32x4*>(a.in_scale + ci0 + b_c4 * 4);
    sh4 = *reinterpret_cast<const f32x4*>(a.in_shift + ci0 + b_c4 * 4);
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int im = 0; im < TM; ++im)
#pragma unroll
    for (int in = 0; in < TN; ++in)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[im][in][r] = 0.f;

  // two register stages, as in conv3x3_bf16_kernel: the loads of pixel-tile kt+2 are in flight while
  // tile kt is multiplied and tile kt+1 is converted, transposed and stored
  typedef typename std::conditional<IO16, u32x2, f32x4>::type SReg;   // 4 channels of one pixel: 8 B of bf16 or 16 B of fp32
  struct Stage {
    SReg ra[APASS][4], rb[BPASS][4];
    unsigned okB;  // bit (q*4 + e): pixel e of group q has its tap inside the image
  };
  Stage S0, S1;

  int tiles_issued = 0;
  auto load_tile = [&](int kt_req, Stage& S) {
    const int kt = min(kt_req, nk - 1);           // clamped tail requests re-load the last tile (never stored)
    const bool advance = kt_req > 0 && kt_req < nk && kt_req == tiles_issued;
    if (kt_req == tiles_issued) ++tiles_issued;
    const int pbase = p_lo + kt * KT;
    const unsigned soff_a = (unsigned)pbase * (unsigned)a.Cout * ES;
    const unsigned soff_b = (unsigned)pbase * (unsigned)a.Cin * ES + tap_shift;
    auto ldx = [&](decltype(rsrc_dy) rs, unsigned voff, unsigned soff) -> SReg {
      if constexpr (IO16) {
        return __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
      } else {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
      }
    };
#pragma unroll
    for (int q = 0; q < APASS; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        S.ra[q][e] = ldx(rsrc_dy, a_off[q] + (unsigned)(e * a.Cout) * ES, soff_a);
    unsigned okB = 0;
#pragma unroll
    for (int q = 0; q < BPASS; ++q) {
      if (advance) {
        int ow = b_ow[q] + adv_s, oh = b_oh[q] + adv_q;
        if (ow >= a.W) { ow -= a.W; oh += 1; }
        while (oh >= a.H) oh -= a.H;
        b_ow[q] = ow;
        b_oh[q] = oh;
      }
      int oh = b_oh[q], ow = b_ow[q];
      const int pix0 = pbase + q * 4 * BPG + b_pg * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ih = oh + dh, iw = ow + dw;
        // (pixels past the tensor are refused too: their dy rows are zero, but 0 * garbage may be NaN)
        const bool ok = (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W && pix0 + e < a.M;
        okB |= ok ? (1u << (q * 4 + e)) : 0u;
        S.rb[q][e] = ldx(rsrc_in, ok ? b_off[q] + (unsigned)(e * a.Cin) * ES : OOB, soff_b);
        if (++ow == a.W) { ow = 0; if (++oh == a.H) oh = 0; }
      }
    }
    S.okB = okB;
  };
  auto store_tile = [&](const Stage& S, int buf) {
    __bf16* Ab = As + buf * BM * KTP;
    __bf16* Bb = Bs + buf * BN * KTP;
    // the 4 channels of pixel e of a stage register as floats (IO16: widened from the bf16 pack in its low half)
    auto wide = [&](const SReg& r) -> f32x4 {
      if constexpr (IO16) {
        const bf16x4 h = __builtin_bit_cast(bf16x4, r);
        return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
      } else {
        return r;
      }
    };
#pragma unroll
    for (int q = 0; q < APASS; ++q) {
      f32x4 y[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = wide(S.ra[q][e]);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4 v = {y[0][c], y[1][c], y[2][c], y[3][c]};   // (bf16 -> fp32 -> bf16 is exact)
        *reinterpret_cast<bf16x4*>(Ab + (a_c4 * 4 + c) * KTP + q * 4 * APG + a_pg * 4) = __builtin_convertvector(v, bf16x4);
      }
    }
#pragma unroll
    for (int q = 0; q < BPASS; ++q) {
      f32x4 x[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        x[e] = wide(S.rb[q][e]);
        if (IN_BN) {
#pragma unroll
          for (int c = 0; c < 4; ++c) x[e][c] = fmaxf(fmaf(x[e][c], sc4[c], sh4[c]), 0.f);
          if (!((S.okB >> (q * 4 + e)) & 1u)) x[e] = f32x4{0.f, 0.f, 0.f, 0.f};  // padding stays 0 after the transform
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4 v = {x[0][c], x[1][c], x[2][c], x[3][c]};
        *reinterpret_cast<bf16x4*>(Bb + (b_c4 * 4 + c) * KTP + q * 4 * BPG + b_pg * 4) = __builtin_convertvector(v, bf16x4);
      }
    }
  };
  auto compute = [&](int buf) {
    const __bf16* Ab = As + buf * BM * KTP + (wm * WTM + l31) * KTP + half * 8;
    const __bf16* Bb = Bs + buf * BN * KTP + (wn * WTN + l31) * KTP + half * 8;
#pragma unroll
    for (int ks = 0; ks < KT / 16; ++ks) {
      bf16x8 af[TM], bf[TN];
#pragma unroll
      for (int im = 0; im < TM; ++im) af[im] = *reinterpret_cast<const bf16x8*>(Ab + im * 32 * KTP + ks * 16);
#pragma unroll
      for (int in = 0; in < TN; ++in) bf[in] = *reinterpret_cast<const bf16x8*>(Bb + in * 32 * KTP + ks * 16);
#pragma unroll
      for (int im = 0; im < TM; ++im)
#pragma unroll
        for (int in = 0; in < TN; ++in)
          acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[im], bf[in], acc[im][in], 0, 0, 0);
    }
  };
  auto iteration = [&](int kt, const Stage& Sst, Stage& Sld) {
    const int cur = kt & 1;
    if (!(a.dbg & 64)) load_tile(kt + 2, Sld);
    __builtin_amdgcn_sched_barrier(0);
    if (!(a.dbg & 16)) compute(cur);
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < nk && !(a.dbg & 32)) store_tile(Sst, cur ^ 1);
    __syncthreads();
  };
  if (nk > 0) {
    load_tile(0, S0);
    load_tile(1, S1);
    store_tile(S0, 0);
  }
  __syncthreads();
  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {
    iteration(kt, S1, S0);
    iteration(kt + 1, S0, S1);
  }
  if (kt < nk) iteration(kt, S1, S0);

  float* slab = a.slabs + (size_t)split * a.Cout * 9 * a.Cin;
#pragma unroll
  for (int im = 0; im < TM; ++im)
#pragma unroll
    for (int in = 0; in < TN; ++in)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * WTM + im * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int ci = ci0 + wn * WTN + in * 32 + l31;
        slab[((size_t)co * 9 + tap) * a.Cin + ci] = acc[im][in][r];
      }
}

// ------------------------------------------------------------------- wgrad, bf16 storage (round 3)
// The kernel above, re-staged for bf16 tensors.  Measured with its ablation bits (tools/gpu_bf16_layers.py, MNIST
// B = 256): weight-gradient launches 1947 us in all, 1238 without the LDS staging, 1610 without the global loads,
// 324 with neither and no MFMA - the transposing stage, not the matrix core, set the time: its 8-byte LDS writes went
// to rows 4 apart at a 144-byte pitch, i.e. to 4 of the 32 banks per 16-lane group (8-way conflicts), and a thread
// fetched 8 bytes per load.  Here a thread loads PG pixels x 8 channels with 16-byte loads (PG = 4 for a 128-channel
// operand, 2 for a 64-channel one), transposes 16-bit elements with v_perm_b32 (no conversion; the BN+ReLU variant
// packs with v_cvt_pk_bf16_f32 instead) and writes PG pixels of one channel per store into LDS[channel][pixel] with
// a 160-byte pitch and the 16-byte chunk index XOR-ed with bits (3, 5, 6) of the row: 2-way on the stores (8 LDS
// cycles against the 6 a ds_write_b64 costs anyway), conflict-free on the ds_read_b128 operand reads.
#define KTW 80   // LDS pitch in elements of the swizzled image (72 measured 5 % slower: conflicts)
__device__ __forceinline__ int wg_swz(int row) { return ((row >> 3) & 1) | (((row >> 5) & 1) << 1) | (((row >> 6) & 1) << 2); }

template <int BM, int BN, bool IN_BN>
__global__ void __launch_bounds__(256, 2)
conv3x3_wgrad_bf16s_kernel(WgradArgs a) {
  constexpr int WGN = 2;
  constexpr int WTM = BM / 2, WTN = BN / 2;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int NCA = BM / 8, NCB = BN / 8;      // 16-byte chunks per pixel
  constexpr int PGA = NCA / 4, PGB = NCB / 4;    // pixels per thread: 256 threads cover KT = 64 pixels x NC chunks
  static_assert((PGA == 2 || PGA == 4) && (PGB == 2 || PGB == 4), "64- or 128-channel operands");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* As = reinterpret_cast<__bf16*>(smem_raw);  // [2][BM][KTW]  dy^T
  __bf16* Bs = As + 2 * BM * KTW;                    // [2][BN][KTW]  in^T (shifted by the tap)

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wave / WGN, wn = wave % WGN;

  const int L = blockIdx.x;
  const int g = (L / 72) * 8 + (L % 8);
  const int tap = (L % 72) / 8;
  if (g >= a.groups) return;
  const int tiles = a.tilesCo * a.tilesCi;
  const int split = g / tiles, tl = g % tiles;
  const int tile_co = tl / a.tilesCi, tile_ci = tl % a.tilesCi;
  const int co0 = tile_co * BM, ci0 = tile_ci * BN;
  const int dh = tap / 3 - 1, dw = tap % 3 - 1;
  const int HW = a.H * a.W;
  const int p_lo = split * a.chunk;
  const int p_hi = min(p_lo + a.chunk, a.M);
  const int nk = (p_hi - p_lo + KT - 1) / KT;

  const int a_c8 = tid % NCA, a_pg = tid / NCA;
  const int b_c8 = tid % NCB, b_pg = tid / NCB;
  constexpr unsigned OOB = 0x80000000u;
  const int neg = (a.W + 1) * a.Cin;
  // the dy descriptor ENDS at this workgroup's last pixel: the ragged end of the range reads zeros
  const auto rsrc_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0,
                                                         (int)((int64_t)p_hi * a.Cout * 2), 0x00020000);
  const auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(const_cast<float*>(a.in)) - (int64_t)neg * 2, 0,
      (int)(((int64_t)a.M * a.Cin + 2 * neg) * 2), 0x00020000);
  const unsigned tap_shift = (unsigned)(((tap / 3) * a.W + (tap % 3)) * a.Cin) * 2u;
  const unsigned a_off = (unsigned)((a_pg * PGA) * a.Cout + co0 + a_c8 * 8) * 2u;
  const unsigned b_off = (unsigned)((b_pg * PGB) * a.Cin + ci0 + b_c8 * 8) * 2u;
  // (oh, ow) of the first pixel this thread stages for B; advanced by KT pixels per K-tile (tiles are requested in order)
  int b_oh, b_ow;
  {
    const int r = (p_lo + b_pg * PGB) % HW;
    b_oh = r / a.W;
    b_ow = r % a.W;
  }
  // one K-tile further = KT mod HW pixels further in the image: adv_q rows (< H) and adv_s columns, one carry each
  const int adv_r = KT % HW;
  const int adv_q = adv_r / a.W, adv_s = adv_r % a.W;
  float sc8[8], sh8[8];
  if (IN_BN) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      sc8[c] = a.in_scale[ci0 + b_c8 * 8 + c];
      sh8[c] = a.in_shift[ci0 + b_c8 * 8 + c];
    }
  }
  // LDS element offsets of this thread's transposed stores (row 8*c8 + c adds c * KTW) and of its operand reads
  const int a_st = (8 * a_c8) * KTW + (PGA == 4 ? (((a_pg >> 1) ^ wg_swz(8 * a_c8)) * 8 + (a_pg & 1) * 4)
                                                : (((a_pg >> 2) ^ wg_swz(8 * a_c8)) * 8 + (a_pg & 3) * 2));
  const int b_st = (8 * b_c8) * KTW + (PGB == 4 ? (((b_pg >> 1) ^ wg_swz(8 * b_c8)) * 8 + (b_pg & 1) * 4)
                                                : (((b_pg >> 2) ^ wg_swz(8 * b_c8)) * 8 + (b_pg & 3) * 2));
  int a_rd[TM], a_sw[TM], b_rd[TN], b_sw[TN];
#pragma unroll
  for (int im = 0; im < TM; ++im) {
    const int row = wm * WTM + im * 32 + l31;
    a_rd[im] = row * KTW;
    a_sw[im] = wg_swz(row) ^ half;
  }
#pragma unroll
  for (int in = 0; in < TN; ++in) {
    const int row = wn * WTN + in * 32 + l31;
    b_rd[in] = row * KTW;
    b_sw[in] = wg_swz(row) ^ half;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int im = 0; im < TM; ++im)
#pragma unroll
    for (int in = 0; in < TN; ++in)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[im][in][r] = 0.f;

  struct Stage {
    u32x4 ra[PGA], rb[PGB];
    unsigned okB;  // bit e: pixel e of this thread's group has its tap inside the image
  };
  Stage S0, S1;

  int tiles_issued = 0;
  auto load_tile = [&](int kt_req, Stage& S) {
    const int kt = min(kt_req, nk - 1);           // clamped tail requests re-load the last tile (never stored)
    const bool advance = kt_req > 0 && kt_req < nk && kt_req == tiles_issued;
    if (kt_req == tiles_issued) ++tiles_issued;
    const int pbase = p_lo + kt * KT;
    const unsigned soff_a = (unsigned)pbase * (unsigned)a.Cout * 2u;
    const unsigned soff_b = (unsigned)pbase * (unsigned)a.Cin * 2u + tap_shift;
#pragma unroll
    for (int e = 0; e < PGA; ++e)
      S.ra[e] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_dy, a_off + (unsigned)(e * a.Cout) * 2u, soff_a, 0);
    {
      int ow = b_ow + adv_s, oh = b_oh + adv_q;
      const bool cw = ow >= a.W;
      ow = cw ? ow - a.W : ow;
      oh = cw ? oh + 1 : oh;
      oh = oh >= a.H ? oh - a.H : oh;
      b_ow = advance ? ow : b_ow;
      b_oh = advance ? oh : b_oh;
    }
    int oh = b_oh, ow = b_ow;
    const int pix0 = pbase + b_pg * PGB;
    unsigned okB = 0;
#pragma unroll
    for (int e = 0; e < PGB; ++e) {
      const int ih = oh + dh, iw = ow + dw;
      // (pixels past the tensor are refused too: their dy rows are zero, but 0 * garbage may be NaN)
      const bool ok = ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W) & (pix0 + e < a.M);   // (no branches)
      okB |= ok ? (1u << e) : 0u;
      S.rb[e] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, ok ? b_off + (unsigned)(e * a.Cin) * 2u : OOB, soff_b, 0);
      const bool we = ow + 1 == a.W;
      ow = we ? 0 : ow + 1;
      oh = we ? (oh + 1 == a.H ? 0 : oh + 1) : oh;
    }
    S.okB = okB;
  };
  // dword j of a 16-byte load holds channels 2j (low half) and 2j+1 of one pixel; a store takes PG pixels of one channel
  auto put = [&](__bf16* dst, unsigned d0, unsigned d1, bool four) {
    if (four) *reinterpret_cast<u32x2*>(dst) = u32x2{d0, d1};
    else *reinterpret_cast<unsigned*>(dst) = d0;
  };
  // one of the 16 stores of a tile: piece 0..7 = channel `piece` of this thread's dy chunk, 8..15 = of its `in` chunk
  auto store_piece = [&](const Stage& S, int buf, int piece) {
    __bf16* Ab = As + buf * BM * KTW + a_st;
    __bf16* Bb = Bs + buf * BN * KTW + b_st;
    const int j = (piece & 7) >> 1, hl = piece & 1;
    if (piece < 8) {
        const unsigned sel = hl ? 0x07060302u : 0x05040100u;
        const unsigned d0 = __builtin_amdgcn_perm(S.ra[1][j], S.ra[0][j], sel);
        const unsigned d1 = PGA == 4 ? __builtin_amdgcn_perm(S.ra[PGA - 1][j], S.ra[PGA - 2][j], sel) : 0u;
        put(Ab + (2 * j + hl) * KTW, d0, d1, PGA == 4);
    } else {
        unsigned d0, d1 = 0u;
        if (IN_BN) {
          // the saved tensor is the raw convolution output: BN + ReLU on the way; padding stays 0 AFTER the transform
          float v[PGB];
#pragma unroll
          for (int e = 0; e < PGB; ++e) {
            const float x = __builtin_bit_cast(float, hl ? (S.rb[e][j] & 0xffff0000u) : (S.rb[e][j] << 16));
            v[e] = ((S.okB >> e) & 1u) ? fmaxf(fmaf(x, sc8[2 * j + hl], sh8[2 * j + hl]), 0.f) : 0.f;
          }
          d0 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[0], v[1]}, bf16x2));
          if (PGB == 4) d1 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[PGB - 2], v[PGB - 1]}, bf16x2));
        } else {
          const unsigned sel = hl ? 0x07060302u : 0x05040100u;
          d0 = __builtin_amdgcn_perm(S.rb[1][j], S.rb[0][j], sel);
          if (PGB == 4) d1 = __builtin_amdgcn_perm(S.rb[PGB - 1][j], S.rb[PGB - 2][j], sel);
        }
        put(Bb + (2 * j + hl) * KTW, d0, d1, PGB == 4);
    }
  };
  auto store_tile = [&](const Stage& S, int buf) {
#pragma unroll
    for (int piece = 0; piece < 16; ++piece) store_piece(S, buf, piece);
  };
  // The MFMAs of tile kt with the 16 stores of tile kt+1 spread between them (16 / (4 TM TN) stores behind each MFMA):
  // with the phases kept apart (loads | MFMAs | stores, one sched_barrier between) the ablation bits showed their times
  // ADD UP (246 us fixed + 332 MFMA + 360 loads + 389 staging against 1196 measured for the MNIST launches) - two
  // workgroups per CU do not interleave by themselves.
  constexpr int NM = 4 * TM * TN, PER = 16 / NM;
  auto iteration = [&](int kt, const Stage& Sst, Stage& Sld, auto store) {
    constexpr bool ST = decltype(store)::value;   // (compile-time: a branch around every store would split the schedule)
    const int cur = kt & 1;
    if (ST) load_tile(kt + 2, Sld);
    __builtin_amdgcn_sched_barrier(0);
    const __bf16* Ab = As + cur * BM * KTW;
    const __bf16* Bb = Bs + cur * BN * KTW;
#pragma unroll
    for (int ks = 0; ks < KT / 16; ++ks) {
      bf16x8 af[TM], bf[TN];
#pragma unroll
      for (int im = 0; im < TM; ++im) af[im] = *reinterpret_cast<const bf16x8*>(Ab + a_rd[im] + ((2 * ks) ^ a_sw[im]) * 8);
#pragma unroll
      for (int in = 0; in < TN; ++in) bf[in] = *reinterpret_cast<const bf16x8*>(Bb + b_rd[in] + ((2 * ks) ^ b_sw[in]) * 8);
#pragma unroll
      for (int im = 0; im < TM; ++im)
#pragma unroll
        for (int in = 0; in < TN; ++in) {
          acc[im][in] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[im], bf[in], acc[im][in], 0, 0, 0);
          if (ST) {
            const int m = (ks * TM + im) * TN + in;
#pragma unroll
            for (int q = 0; q < PER; ++q) store_piece(Sst, cur ^ 1, m * PER + q);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();
  };
  if (nk > 0) {
    load_tile(0, S0);
    load_tile(1, S1);
    store_tile(S0, 0);
  }
  __syncthreads();
  // every iteration but the last stores the next tile (and requests the one after it: the tail re-loads the last tile)
  int kt = 0;
  for (; kt + 2 < nk; kt += 2) {
    iteration(kt, S1, S0, std::true_type{});
    iteration(kt + 1, S0, S1, std::true_type{});
  }
  if (kt + 2 == nk) {
    iteration(kt, S1, S0, std::true_type{});
    iteration(kt + 1, S0, S1, std::false_type{});
  } else if (kt < nk) {
    iteration(kt, S1, S0, std::false_type{});
  }

  float* slab = a.slabs + (size_t)split * a.Cout * 9 * a.Cin;
#pragma unroll
  for (int im = 0; im < TM; ++im)
#pragma unroll
    for (int in = 0; in < TN; ++in)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * WTM + im * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int ci = ci0 + wn * WTN + in * 32 + l31;
        slab[((size_t)co * 9 + tap) * a.Cin + ci] = acc[im][in][r];
      }
}

// ------------------------------------------------------------ wgrad, nine taps per workgroup (round 3)
// The kernels above give every tap its own workgroup: the dy tile is staged nine times and the input tile nine times
// shifted, 64 FLOP per byte that crosses the L1 - and every bf16 launch of this library settles at 7-10 TB/s of L1
// fills, whatever its kernel does in between.  Here ONE workgroup owns a 64 x 64 (co, ci) tile for ALL nine taps of a
// pixel range: 288 FLOP per byte.  What makes that possible:
//   * both operands stay in LDS the way they lie in memory, [pixel][channel] rows of 128 bytes, filled by LDS-DMA
//     (16 B per lane, no registers), and the MFMA operands are read with ds_read_b64_tr_b16 (the gfx950 transposing
//     read: a 16-lane group fetches 4 rows x 16 channels and each lane receives one channel's four pixels);
//   * a tap is then a ROW OFFSET of the input image, if the pixel index is made one-dimensional: K runs over SLOTS of a
//     padded image, (H+1) x (W+1) per sample with a zero column after every row and a zero row after every sample,
//     slot = (n (H+1) + oh)(W+1) + ow, so that the (dh, dw) neighbour of a slot is slot + dh (W+1) + dw and every
//     out-of-image neighbour lands on a zero slot.  dy rows of pad slots are zero too (OOB DMA), so they add nothing;
//   * the input rows live in a ring of 64-slot blocks (each block is fetched once; the halo of the next K-tile is the
//     body of this one), with the first 64 rows mirrored behind the last so that a 64-row window never wraps.
// K inflates by (H+1)(W+1)/(HW): 7 % at 28 x 28, 31 % at 7 x 7 (where there is little K anyway).
// Per K-tile and wave: 36 MFMAs (one 32 x 32 tile per tap), 8 + 72 transposing reads, 4 DMA instructions.
template <bool IN_BN>
__global__ void __launch_bounds__(256, 2)
conv3x3_wgrad9_bf16_kernel(WgradArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // LDS: A [2][64 rows][128 B] | B ring [R + 64 rows][128 B]
  const int PW = a.W + 1, PH = a.H + 1, PP = PH * PW;
  const int HALOB = (PW + 1 + 63) / 64 * 64;          // block-aligned halo in front of the first tile
  const int NB = (HALOB + 64 + PW) / 64;               // tile t reads blocks t .. t+NB
  const int NBLK = NB + 2;                              // + the block in flight
  const int R = NBLK * 64;
  unsigned char* Abuf = smem_raw;
  unsigned char* Bring = smem_raw + 2 * 64 * 128;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;

  // workgroup -> (pixel chunk, tile): the tiles of one chunk share an XCD (they read the same pixels)
  const int tiles = a.tilesCo * a.tilesCi;
  const int L = blockIdx.x;
  const int split = (L / (8 * tiles)) * 8 + (L % 8);
  const int tl = (L / 8) % tiles;
  if (split * a.chunk >= a.M) return;
  const int co0 = (tl / a.tilesCi) * 64, ci0 = (tl % a.tilesCi) * 64;
  const int HW = a.H * a.W;
  const int p_lo = split * a.chunk, p_hi = min(p_lo + a.chunk, a.M);
  auto slot_of = [&](int p) { const int n = p / HW, r = p - n * HW; const int oh = r / a.W; return n * PP + oh * PW + (r - oh * a.W); };
  const int s_lo = slot_of(p_lo), s_hi = slot_of(p_hi - 1) + 1;   // dy rows outside [s_lo, s_hi) are zero for this workgroup
  const int nk = (s_hi - s_lo + 63) / 64;

  const auto rsrc_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, (int)((int64_t)a.M * a.Cout * 2), 0x00020000);
  const auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, (int)((int64_t)a.M * a.Cin * 2), 0x00020000);
  constexpr unsigned OOB = 0x80000000u;

  // ---- staging: instruction i of wave w covers local rows (i*4 + w)*8 .. +7; lane -> row + (lane >> 3), stored chunk
  // lane & 7 = logical chunk ^ swz(row), swz(r) = ((r >> 1) & 1) << 2.  IN_BN register path: logical chunk tid & 7 of
  // rows (tid >> 3) + 32 i.
  struct Pos { int n, oh, ow; };
  auto pos_of = [&](int slot) {   // slot may be negative (halo in front of the tensor): shift by 8 samples
    const int s8 = slot + 8 * PP;
    Pos q; q.n = s8 / PP - 8; const int r = s8 % PP; q.oh = r / PW; q.ow = r - q.oh * PW; return q;
  };
  const int dn = 64 / PP, d_oh = (64 % PP) / PW, d_ow = (64 % PP) % PW;   // 64 slots further
  auto advance = [&](Pos& q) {
    int ow = q.ow + d_ow; const bool c = ow >= PW; ow = c ? ow - PW : ow;
    int oh = q.oh + d_oh + (c ? 1 : 0); const bool c2 = oh >= PH; oh = c2 ? oh - PH : oh;
    q.ow = ow; q.oh = oh; q.n += dn + (c2 ? 1 : 0);
  };
  const int nB = a.M / HW;
  auto pixel = [&](const Pos& q, bool& real) {
    real = ((unsigned)q.n < (unsigned)nB) & (q.oh < a.H) & (q.ow < a.W);
    return (q.n * a.H + q.oh) * a.W + q.ow;
  };
  int rowA[2], rowB[2];
  Pos pa[2], pb[2];
  unsigned cha[2], chb[2];   // byte offset of this lane's 16-byte chunk inside a pixel row
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    rowA[i] = (i * 4 + wave) * 8 + (lane >> 3);
    rowB[i] = IN_BN ? (tid >> 3) + 32 * i : rowA[i];
    pa[i] = pos_of(s_lo + rowA[i]);
    pb[i] = pos_of(s_lo - HALOB + rowB[i]);
    cha[i] = (unsigned)(co0 + (((lane & 7) ^ (((rowA[i] >> 1) & 1) << 2)) << 3)) * 2u;
    chb[i] = (unsigned)(ci0 + ((IN_BN ? (tid & 7) : ((lane & 7) ^ (((rowB[i] >> 1) & 1) << 2))) << 3)) * 2u;
  }
  float sc8[8], sh8[8];
  if (IN_BN) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      sc8[c] = a.in_scale[ci0 + (tid & 7) * 8 + c];
      sh8[c] = a.in_shift[ci0 + (tid & 7) * 8 + c];
    }
  }
  // tile t of dy -> Abuf[t & 1]; the slot state advances with every call
  int a_slot0 = s_lo;
  auto issue_A = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      bool real;
      const int p = pixel(pa[i], real);
      const int s = a_slot0 + rowA[i];
      real = real & (s < s_hi);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_dy, (lds_ptr_t)(Abuf + buf * 8192 + (i * 4 + wave) * 1024), 16,
                                               real ? (unsigned)p * (unsigned)a.Cout * 2u + cha[i] : OOB, 0, 0, 0);
      advance(pa[i]);
    }
    a_slot0 += 64;
  };
  // block b of the input -> ring rows (b mod NBLK)*64 .. +63 (+ the mirror behind the ring for the first 64 rows)
  struct BRegs { u32x4 r[2]; unsigned ok; };
  int b_blk = 0;   // ring position (block index mod NBLK) of the next block
  auto issue_B = [&](BRegs& Rg) {
    unsigned ok = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      bool real;
      const int p = pixel(pb[i], real);
      const unsigned off = real ? (unsigned)p * (unsigned)a.Cin * 2u + chb[i] : OOB;
      if (IN_BN) {
        ok |= real ? (1u << i) : 0u;
        Rg.r[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, off, 0, 0);
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_in, (lds_ptr_t)(Bring + b_blk * 8192 + (i * 4 + wave) * 1024), 16, off, 0, 0, 0);
        if (b_blk == 0)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_in, (lds_ptr_t)(Bring + R * 128 + (i * 4 + wave) * 1024), 16, off, 0, 0, 0);
      }
      advance(pb[i]);
    }
    Rg.ok = ok;
    if (!IN_BN) b_blk = b_blk + 1 == NBLK ? 0 : b_blk + 1;
  };
  // IN_BN: the block held in registers -> BN + ReLU -> ring (pad slots stay 0 AFTER the transform)
  auto put_B = [&](const BRegs& Rg) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bf16x8 h = __builtin_bit_cast(bf16x8, Rg.r[i]);
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (__bf16)fmaxf(fmaf((float)h[e], sc8[e], sh8[e]), 0.f);
      u32x4 raw = __builtin_bit_cast(u32x4, o);
      if (!((Rg.ok >> i) & 1u)) raw = u32x4{0u, 0u, 0u, 0u};
      const int row = rowB[i];
      const int off = row * 128 + (((tid & 7) ^ (((row >> 1) & 1) << 2)) << 4);
      *reinterpret_cast<u32x4*>(Bring + b_blk * 8192 + off) = raw;
      if (b_blk == 0) *reinterpret_cast<u32x4*>(Bring + R * 128 + off) = raw;
    }
    b_blk = b_blk + 1 == NBLK ? 0 : b_blk + 1;
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // ---- transposing operand reads (cdna_hip_programming.md T10; tools/micro/tr_read_probe.hip checks this addressing):
  // lane = 16 g + 4 q + pp supplies row q of its group's 4 x 16 block, channels 4 pp .. 4 pp + 3, and receives channel
  // 16 (g & 1) + (lane & 15) of the four rows.  A k-step is 16 slots: lanes 0-31 take slots 0-7 of it, lanes 32-63 slots
  // 8-15, two reads of four rows each.  Row offsets that are multiples of 4 keep bit 1 of the row, i.e. the swizzle, so
  // the eight reads of a (tile, tap) share one address register and differ in the immediate offset.
  typedef __bf16 bf16x4v __attribute__((__vector_size__(4 * sizeof(__bf16))));
  typedef __attribute__((address_space(3))) bf16x4v* lds_v4;
  const int g = lane >> 4, q4 = (lane >> 2) & 3, pp = lane & 3;
  const int lrow = 8 * (g >> 1) + q4;
  const int chA = wm * 4 + 2 * (g & 1) + (pp >> 1), chBf = wn * 4 + 2 * (g & 1) + (pp >> 1);
  const int subb = 8 * (pp & 1);
  auto frag = [&](const unsigned char* base, int ks) -> bf16x8 {
    const bf16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(base + (16 * ks) * 128));
    const bf16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(base + (16 * ks + 4) * 128));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  };

  BRegs Rg;
  // prologue: dy tile 0 and input blocks 0 .. NB
  issue_A(0);
  for (int b = 0; b <= NB; ++b) {
    issue_B(Rg);
    if (IN_BN) put_B(Rg);
  }
  int tbase = HALOB % R;   // ring row of slot s_lo + 64 t
  for (int t = 0; t < nk; ++t) {
    __syncthreads();   // tile t and its blocks have landed (the fence drains the DMA); tile t-1 has been read by every wave
    const bool more = t + 1 < nk && !(a.dbg & 64);   // (ablation bits of tools/gpu_bf16_layers.py: 64 no loads, 16 no MFMAs / reads)
    if (more) {
      issue_A((t + 1) & 1);
      issue_B(Rg);
    }
    if (a.dbg & 16) continue;
    const unsigned char* Ab = Abuf + (t & 1) * 8192 + lrow * 128 + ((chA ^ (((lrow >> 1) & 1) << 2)) << 4) + subb;
    bf16x8 af[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) af[ks] = frag(Ab, ks);
    // the input operand of tap+1 (eight reads) is requested before the four MFMAs of tap: hipcc otherwise keeps two
    // fragments in flight and every MFMA pair waits for its LDS round trip
    bf16x8 bfr[2][4];
    auto load_tap = [&](int tap, bf16x8 (&dst)[4]) {
      int r0 = tbase + (tap / 3 - 1) * PW + (tap % 3 - 1);   // first ring row of this tap's 64-row window (uniform)
      r0 = r0 < 0 ? r0 + R : (r0 >= R ? r0 - R : r0);
      const int row = r0 + lrow;
      const unsigned char* Bb = Bring + row * 128 + ((chBf ^ (((row >> 1) & 1) << 2)) << 4) + subb;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) dst[ks] = frag(Bb, ks);
    };
    load_tap(0, bfr[0]);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap + 1 < 9) load_tap(tap + 1, bfr[(tap + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], bfr[tap & 1][ks], acc[tap], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (IN_BN && more) put_B(Rg);
    tbase = tbase + 64 >= R ? tbase + 64 - R : tbase + 64;
  }

  float* slab = a.slabs + (size_t)split * a.Cout * 9 * a.Cin;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      const int ci = ci0 + wn * 32 + l31;
      slab[((size_t)co * 9 + tap) * a.Cin + ci] = acc[tap][r];
    }
#endif
}

// ------------------------------------------------------------------------------------------ packing
// OIHW fp32 -> bf16 forward pack [Cout][9][Cin] and dgrad pack [Cin][9 mirrored][Cout]; one launch for
// all units of a network (TdxPackBatch, destinations reinterpreted as bf16)
__global__ void pack_conv3x3_batch_bf16_kernel(TdxPackBatch b) {
  int u = 0;
  while (u + 1 < b.count && (int)blockIdx.x >= b.chunk_start[u + 1]) ++u;
  const int64_t n = (int64_t)b.cout[u] * b.cin[u] * 9;
  const int cin = b.cin[u], cin_real = b.cin_real[u], cout = b.cout[u];
  const float* __restrict__ w = b.w[u];
  __bf16* __restrict__ wf = reinterpret_cast<__bf16*>(b.wf[u]);
  __bf16* __restrict__ wd = reinterpret_cast<__bf16*>(b.wd[u]);
  const int64_t base = (int64_t)(blockIdx.x - b.chunk_start[u]) * 1024;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t i = base + k * 256 + threadIdx.x;
    if (i >= n) break;
    const int ci = (int)(i % cin);
    const int tap = (int)((i / cin) % 9);
    const int co = (int)(i / ((int64_t)9 * cin));
    const float v = ci < cin_real ? w[((size_t)co * cin_real + ci) * 9 + tap] : 0.f;
    if (wf) wf[i] = (__bf16)v;
    if (wd) wd[((size_t)ci * 9 + (8 - tap)) * cout + co] = (__bf16)v;
  }
}

int tdx_pack_conv3x3_batch_bf16(TdxPackBatch* b, tdx_stream_t stream) {
  if (!b || b->count <= 0 || b->count > TDX_PACK_MAX) return TDX_E_BADARG;
  int chunks = 0;
  for (int u = 0; u < b->count; ++u) {
    if (!b->w[u] || (!b->wf[u] && !b->wd[u]) || b->cin_real[u] <= 0 || b->cin_real[u] > b->cin[u]) return TDX_E_BADARG;
    b->chunk_start[u] = chunks;
    chunks += cdiv((int64_t)b->cout[u] * b->cin[u] * 9, 1024);
  }
  pack_conv3x3_batch_bf16_kernel<<<chunks, 256, 0, to_stream(stream)>>>(*b);
  TDX_CHECK_LAUNCH();
  return 0;
}

extern "C" int tdx_pack_conv3x3_bf16(const float* w_oihw, void* w_fwd_bf16, void* w_dgrad_bf16, int cout, int cin,
                                     tdx_stream_t stream) {
  if (!w_oihw || cout <= 0 || cin <= 0) return TDX_E_BADARG;
  TdxPackBatch b;
  b.count = 1;
  b.w[0] = w_oihw; b.wf[0] = static_cast<float*>(w_fwd_bf16); b.wd[0] = static_cast<float*>(w_dgrad_bf16);
  b.cout[0] = cout; b.cin[0] = cin; b.cin_real[0] = cin;
  return tdx_pack_conv3x3_batch_bf16(&b, stream);
}

// ----------------------------------------------------------------------------------------- dispatch
extern "C" int tdx_conv3x3_bf16_stat_tile_rows(void) { return 128; }

template <int BM, int BN, bool IO16>
static int launch_bf16(const ConvArgs& a, int flags, hipStream_t st) {
  const size_t lds = (size_t)2 * (BM + BN) * KTP * sizeof(__bf16);
  const int grid = (cdiv(a.M, BM) + 7) / 8 * 8 * a.tilesN;
  const bool in_bn = flags & TDX_CONV_IN_BNRELU;
  const int epi = (flags & TDX_CONV_OUT_BNRELU) ? EPI_BNRELU : (flags & TDX_CONV_OUT_STATS) ? EPI_STATS : EPI_PLAIN;
#define TDX_LAUNCH_BF16(INBN, EPI_)                                                                  \
  do {                                                                                               \
    auto kern = conv3x3_bf16_kernel<BM, BN, INBN, EPI_, IO16>;                                       \
    static bool attr_set = false;                                                                    \
    if (lds > 65536 && !attr_set) {                                                                  \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                        \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);      \
      if (e != hipSuccess) return (int)e;                                                            \
      attr_set = true;                                                                               \
    }                                                                                                \
    kern<<<grid, 256, lds, st>>>(a);                                                                 \
  } while (0)
  if (in_bn) {
    if (epi == EPI_BNRELU) TDX_LAUNCH_BF16(true, EPI_BNRELU);
    else if (epi == EPI_STATS) TDX_LAUNCH_BF16(true, EPI_STATS);
    else TDX_LAUNCH_BF16(true, EPI_PLAIN);
  } else {
    if (epi == EPI_BNRELU) TDX_LAUNCH_BF16(false, EPI_BNRELU);
    else if (epi == EPI_STATS) TDX_LAUNCH_BF16(false, EPI_STATS);
    else TDX_LAUNCH_BF16(false, EPI_PLAIN);
  }
#undef TDX_LAUNCH_BF16
  TDX_CHECK_LAUNCH();
  return 0;
}

// conv3x3_bf16_ring_kernel: 256-row tiles, 512 threads, three DMA stages (+ 8 KB for the BN scale / shift copy)
// Off by default: isolated, the ring kernel is 6-9 % faster over the MNIST launches (forward 1109 -> 1043 us, input
// gradient 955 -> 865), but one 155 KB workgroup per CU shares a CU with nothing, and inside the three-stream step it
// measured SLOWER (MNIST bf16 4.50 vs 4.40 ms, LAION 64x64 10.29 vs 10.12).  tdx_tune_set("bf16_ring", 1) selects it.
int g_tdx_bf16_ring = 0;
template <int BN>
static int launch_bf16_ring(const ConvArgs& a, int flags, hipStream_t st) {
  const size_t lds = (size_t)3 * (256 + BN) * 64 * sizeof(__bf16) + 8192;
  const int grid = (cdiv(a.M, 256) + 7) / 8 * 8 * a.tilesN;
  const bool in_bn = flags & TDX_CONV_IN_BNRELU;
  const int epi = (flags & TDX_CONV_OUT_BNRELU) ? EPI_BNRELU : (flags & TDX_CONV_OUT_STATS) ? EPI_STATS : EPI_PLAIN;
#define TDX_LAUNCH_RING(INBN, EPI_)                                                                  \
  do {                                                                                               \
    auto kern = conv3x3_bf16_ring_kernel<BN, INBN, EPI_>;                                            \
    static bool attr_set = false;                                                                    \
    if (!attr_set) {                                                                                 \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                        \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);      \
      if (e != hipSuccess) return (int)e;                                                            \
      attr_set = true;                                                                               \
    }                                                                                                \
    kern<<<grid, 512, lds, st>>>(a);                                                                 \
  } while (0)
  if (in_bn) {
    if (epi == EPI_BNRELU) TDX_LAUNCH_RING(true, EPI_BNRELU);
    else if (epi == EPI_STATS) TDX_LAUNCH_RING(true, EPI_STATS);
    else TDX_LAUNCH_RING(true, EPI_PLAIN);
  } else {
    if (epi == EPI_BNRELU) TDX_LAUNCH_RING(false, EPI_BNRELU);
    else if (epi == EPI_STATS) TDX_LAUNCH_RING(false, EPI_STATS);
    else TDX_LAUNCH_RING(false, EPI_PLAIN);
  }
#undef TDX_LAUNCH_RING
  TDX_CHECK_LAUNCH();
  return 0;
}

// conv3x3_bf16_thin_kernel: 256-row x 64-column tiles, 512 threads, 80 KB of LDS.  tdx_tune_set("bf16_thin", v):
// 0 off | 1 layers with 64 output channels | 2 also layers with 64 input channels | 3 (default) every raw-input launch
// whose rows are a multiple of 256 (wider layers run one workgroup per 64 output channels, each with its own copy of
// the window).  Measured, B = 256: isolated (LAION 64x64) 64 -> 64 forward 201 -> 129 us, its input gradient 151-193 ->
// 104-144, 192 -> 64 forward 428 -> 273, its input gradient (64 -> 192) 473 -> 335; steps in ms, knob 0 / 1 / 2 / 3:
// LAION 64x64 9.34 / 8.84 / 8.65 / 8.49, LAION 32x32 3.24 / 3.11 / 3.08 / 2.96, MNIST 4.27 / 4.16 / 4.13 / 3.98.  Still
// 3-4x the thin layers' HBM time: a workgroup's 72 MFMAs per wave (1 us) wait for nine weight tiles one L2 round trip
// each - keeping the 72 KB of weights resident across tiles is the next step.
int g_tdx_bf16_thin = 3;
static bool thin_window_fits(int H, int W) {   // the largest window of any 256-pixel tile: its slots + the halo, exactly
  const int PW = W + 1, PP = (H + 1) * PW, HW = H * W;
  auto slot_of = [&](int p) { const int n = p / HW, r = p - n * HW; const int oh = r / W; return n * PP + oh * PW + (r - oh * W); };
  int worst = 0;
  for (int j = 0; j < HW; ++j) {   // tile starts 256 j cover every phase of a sample (period HW / gcd(HW, 256) tiles)
    const int m0 = 256 * j;
    worst = std::max(worst, slot_of(m0 + 255) - slot_of(m0) + 1 + 2 * (PW + 1));
    if ((m0 + 256) % HW == 0) break;
  }
  return worst <= THIN_WROWS;
}
static int launch_bf16_thin(const ConvArgs& a, int flags, hipStream_t st) {
  const size_t lds = (size_t)THIN_WROWS * 128 + 3 * 8192;
  const int grid = (cdiv(a.M, 256) + 7) / 8 * 8 * a.tilesN;
  const int epi = (flags & TDX_CONV_OUT_BNRELU) ? EPI_BNRELU : (flags & TDX_CONV_OUT_STATS) ? EPI_STATS : EPI_PLAIN;
#define TDX_LAUNCH_THIN(EPI_)                                                                        \
  do {                                                                                               \
    auto kern = conv3x3_bf16_thin_kernel<EPI_>;                                                      \
    static bool attr_set = false;                                                                    \
    if (!attr_set) {                                                                                 \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                        \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);      \
      if (e != hipSuccess) return (int)e;                                                            \
      attr_set = true;                                                                               \
    }                                                                                                \
    kern<<<grid, 512, lds, st>>>(a);                                                                 \
  } while (0)
  if (epi == EPI_BNRELU) TDX_LAUNCH_THIN(EPI_BNRELU);
  else if (epi == EPI_STATS) TDX_LAUNCH_THIN(EPI_STATS);
  else TDX_LAUNCH_THIN(EPI_PLAIN);
#undef TDX_LAUNCH_THIN
  TDX_CHECK_LAUNCH();
  return 0;
}

extern "C" int tdx_conv3x3_fwd_bf16(const float* in, const void* wpk_bf16, const float* bias, float* out,
                                    int B, int H, int W, int cin, int cout, int flags,
                                    const float* in_scale, const float* in_shift,
                                    const float* out_scale, const float* out_shift,
                                    float* stats_partial, tdx_stream_t stream) {
  return tdx_conv3x3_fwd_bf16_io(in, wpk_bf16, bias, out, B, H, W, cin, cout, flags, in_scale, in_shift, out_scale, out_shift,
                                 stats_partial, 0, stream);
}

// io16: `in` and `out` hold bf16 (bf16 storage mode; the C-ABI entry above is the fp32-tensor form)
extern "C" int tdx_conv3x3_fwd_bf16_io(const void* in_, const void* wpk_bf16, const float* bias, void* out_, int B, int H, int W,
                            int cin, int cout, int flags, const float* in_scale, const float* in_shift,
                            const float* out_scale, const float* out_shift, float* stats_partial, int io16,
                            tdx_stream_t stream) {
  const float* in = static_cast<const float*>(in_);
  float* out = static_cast<float*>(out_);
  if (!in || !wpk_bf16 || !out || B <= 0 || H <= 0 || W <= 0) return TDX_E_BADARG;
  if (cin % KT || cout % 64) return TDX_E_SHAPE;
  if ((flags & TDX_CONV_IN_BNRELU) && (!in_scale || !in_shift)) return TDX_E_BADARG;
  if ((flags & TDX_CONV_OUT_BNRELU) && (!out_scale || !out_shift)) return TDX_E_BADARG;
  if ((flags & TDX_CONV_OUT_STATS) && !stats_partial) return TDX_E_BADARG;
  if ((flags & TDX_CONV_OUT_STATS) && (flags & TDX_CONV_OUT_BNRELU)) return TDX_E_BADARG;
  if (!tdx_conv3x3_shape_ok(B, H, W, cin, cout)) return TDX_E_SHAPE;
  ConvArgs a{};
  a.in = in; a.w = static_cast<const float*>(wpk_bf16); a.bias = bias; a.out = out;
  a.in_scale = in_scale; a.in_shift = in_shift; a.out_scale = out_scale; a.out_shift = out_shift;
  a.stats = stats_partial;
  a.B = B; a.H = H; a.W = W; a.Cin = cin; a.Cout = cout; a.M = (int)((int64_t)B * H * W);
  a.splits = 1; a.kt_per_split = 0; a.dbg = tdx_conv_dbg_get(); a.stamps = nullptr;
  a.out_bf16 = io16 ? 1 : 0;
  hipStream_t st = to_stream(stream);
  if (io16 && !(flags & TDX_CONV_IN_BNRELU) && a.M % 256 == 0 && thin_window_fits(H, W) &&
      ((g_tdx_bf16_thin >= 1 && cout == 64) || (g_tdx_bf16_thin >= 2 && cin == 64) || g_tdx_bf16_thin >= 3)) {
    a.tilesN = cout / 64;
    return launch_bf16_thin(a, flags, st);
  }
  if (io16 && g_tdx_bf16_ring && a.M % 256 == 0 && cin <= 1024) {
    a.tilesN = cout % 128 == 0 ? cout / 128 : cout / 64;
    return cout % 128 == 0 ? launch_bf16_ring<128>(a, flags, st) : launch_bf16_ring<64>(a, flags, st);
  }
  // memory-bound: the widest column tile re-reads the input least often
  if (cout % 128 == 0) {
    a.tilesN = cout / 128;
    return io16 ? launch_bf16<128, 128, true>(a, flags, st) : launch_bf16<128, 128, false>(a, flags, st);
  }
  a.tilesN = cout / 64;
  return io16 ? launch_bf16<128, 64, true>(a, flags, st) : launch_bf16<128, 64, false>(a, flags, st);
}

template <int BM, int BN, bool IO16>
static int launch_wgrad_bf16(const WgradArgs& a, bool in_bn, hipStream_t st) {
  const size_t lds = (size_t)2 * (BM + BN) * KTP * sizeof(__bf16);
  dim3 grid((unsigned)(((int64_t)a.groups + 7) / 8 * 72));
#define TDX_LAUNCH_WG(INBN)                                                                          \
  do {                                                                                               \
    auto kern = conv3x3_wgrad_bf16_kernel<BM, BN, INBN, IO16>;                                       \
    static bool attr_set = false;                                                                    \
    if (lds > 65536 && !attr_set) {                                                                  \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                        \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);      \
      if (e != hipSuccess) return (int)e;                                                            \
      attr_set = true;                                                                               \
    }                                                                                                \
    kern<<<grid, 256, lds, st>>>(a);                                                                 \
  } while (0)
  if (in_bn) TDX_LAUNCH_WG(true);
  else TDX_LAUNCH_WG(false);
#undef TDX_LAUNCH_WG
  TDX_CHECK_LAUNCH();
  return 0;
}

// bf16-storage form (conv3x3_wgrad_bf16s_kernel); tdx_tune_set("bf16_wgrad_swz", 0) goes back to the IO16 variant above
int g_tdx_wgrad_bf16s = 1;
template <int BM, int BN>
static int launch_wgrad_bf16s(const WgradArgs& a, bool in_bn, hipStream_t st) {
  const size_t lds = (size_t)2 * (BM + BN) * KTW * sizeof(__bf16);   // 80 KB for 128 x 128: two workgroups fill the CU's 160 KB
  dim3 grid((unsigned)(((int64_t)a.groups + 7) / 8 * 72));
#define TDX_LAUNCH_WG(INBN)                                                                          \
  do {                                                                                               \
    auto kern = conv3x3_wgrad_bf16s_kernel<BM, BN, INBN>;                                            \
    static bool attr_set = false;                                                                    \
    if (lds > 65536 && !attr_set) {                                                                  \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                        \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);      \
      if (e != hipSuccess) return (int)e;                                                            \
      attr_set = true;                                                                               \
    }                                                                                                \
    kern<<<grid, 256, lds, st>>>(a);                                                                 \
  } while (0)
  if (in_bn) TDX_LAUNCH_WG(true);
  else TDX_LAUNCH_WG(false);
#undef TDX_LAUNCH_WG
  TDX_CHECK_LAUNCH();
  return 0;
}

// conv3x3_wgrad9_bf16_kernel: one workgroup per (pixel chunk, 64 x 64 tile), all nine taps; same chunks and slabs as
// the per-tap kernels (tdx_wgrad_plan), 2.25-4.5x fewer workgroups.  tdx_tune_set("bf16_wgrad9", 0) switches it off.
int g_tdx_wgrad9 = 1;
static int launch_wgrad9_bf16(WgradArgs a, int splits, bool in_bn, hipStream_t st) {
  a.tilesCo = a.Cout / 64;
  a.tilesCi = a.Cin / 64;
  const int PW = a.W + 1;
  const int HALOB = (PW + 1 + 63) / 64 * 64, NB = (HALOB + 64 + PW) / 64, R = (NB + 2) * 64;   // as in the kernel
  const size_t lds = (size_t)2 * 64 * 128 + (size_t)(R + 64) * 128;
  dim3 grid((unsigned)((splits + 7) / 8 * 8 * a.tilesCo * a.tilesCi));
#define TDX_LAUNCH_WG9(INBN)                                                                         \
  do {                                                                                               \
    auto kern = conv3x3_wgrad9_bf16_kernel<INBN>;                                                    \
    static bool attr_set = false;                                                                    \
    if (!attr_set) {                                                                                 \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                        \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 98304);         \
      if (e != hipSuccess) return (int)e;                                                            \
      attr_set = true;                                                                               \
    }                                                                                                \
    kern<<<grid, 256, lds, st>>>(a);                                                                 \
  } while (0)
  if (in_bn) TDX_LAUNCH_WG9(true);
  else TDX_LAUNCH_WG9(false);
#undef TDX_LAUNCH_WG9
  TDX_CHECK_LAUNCH();
  return 0;
}

extern "C" int tdx_conv3x3_wgrad_bf16(const float* in, const float* dy, float* dw_slabs, int B, int H,
                                      int W, int cin, int cout, int flags, const float* in_scale,
                                      const float* in_shift, tdx_stream_t stream) {
  return tdx_conv3x3_wgrad_bf16_io(in, dy, dw_slabs, B, H, W, cin, cout, flags, in_scale, in_shift, 0, stream);
}

extern "C" int tdx_conv3x3_wgrad_bf16_io(const void* in_, const void* dy_, float* dw_slabs, int B, int H, int W, int cin, int cout,
                              int flags, const float* in_scale, const float* in_shift, int io16, tdx_stream_t stream) {
  const float* in = static_cast<const float*>(in_);
  const float* dy = static_cast<const float*>(dy_);
  if (!in || !dy || !dw_slabs || B <= 0 || H <= 0 || W <= 0) return TDX_E_BADARG;
  if (cin % 64 || cout % 64) return TDX_E_SHAPE;
  const bool in_bn = flags & TDX_CONV_IN_BNRELU;
  if (in_bn && (!in_scale || !in_shift)) return TDX_E_BADARG;
  if (!tdx_conv3x3_shape_ok(B, H, W, cin, cout)) return TDX_E_SHAPE;
  const int64_t M64 = (int64_t)B * H * W;
  int bm, bn, splits, chunk;
  tdx_wgrad_plan(M64, cin, cout, &bm, &bn, &splits, &chunk, true);  // same slab layout and reduce as the fp32 path, own tiles / splits
  WgradArgs a;
  a.in = in; a.dy = dy; a.slabs = dw_slabs; a.in_scale = in_scale; a.in_shift = in_shift;
  a.B = B; a.H = H; a.W = W; a.Cin = cin; a.Cout = cout; a.M = (int)M64;
  a.tilesCi = cin / bn; a.tilesCo = cout / bm; a.chunk = chunk;
  a.groups = a.tilesCi * a.tilesCo * splits;
  a.adv_q = 0; a.adv_s = 0; a.dbg = tdx_conv_dbg_get();
  hipStream_t st = to_stream(stream);
  if (io16 && g_tdx_wgrad9 && W <= 95) return launch_wgrad9_bf16(a, splits, in_bn, st);   // (ring of <= 7 blocks: 96 KB of LDS)
#define TDX_WG(BM_, BN_)                                                                        \
  (io16 ? (g_tdx_wgrad_bf16s ? launch_wgrad_bf16s<BM_, BN_>(a, in_bn, st) : launch_wgrad_bf16<BM_, BN_, true>(a, in_bn, st)) \
        : launch_wgrad_bf16<BM_, BN_, false>(a, in_bn, st))
  if (bm == 128 && bn == 128) return TDX_WG(128, 128);
  if (bm == 128 && bn == 64) return TDX_WG(128, 64);
  if (bm == 64 && bn == 128) return TDX_WG(64, 128);
  return TDX_WG(64, 64);
#undef TDX_WG
}

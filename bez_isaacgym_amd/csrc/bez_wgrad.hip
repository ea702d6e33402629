// bez_wgrad.hip -- the weight gradients of a PPO minibatch step as ONE split-K MFMA kernel + one deterministic reduction.
//
// rl_games' backward pass (torch autograd through the actor-critic MLP, a2c_common.py calc_gradients [ext] via train.py:89-113)
// computes dW_L = dY_L^T X_L for the five Linear layers: outputs of 400x54 ... 1x100 elements with a reduction over the
// 32768 rows of the minibatch.  As GEMM-library calls these are either 7-workgroup launches (108 us each) or, reshaped into a
// 32-way batched GEMM + a sum (round 2), ~95 us of a 243 us minibatch step.  Here:
//   * both operands are row-major with the REDUCTION index as the row (dY is (K, out), X is (K, in)), while an MFMA lane wants 8
//     consecutive k of ONE column: the tiles are staged row-major into LDS with coalesced 16-byte loads and read back through
//     gfx950's transposing LDS read (ds_read_b64_tr_b16: a 4-row x 16-column block delivered column-major), so neither operand
//     is ever transposed in memory;
//   * a workgroup owns one output block (<= 4 x 8 tiles of 32 x 32) of one layer for one K-chunk and keeps it in registers.  The
//     launch is ONE round of ~250 workgroups on 256 CUs: every block is split along K in proportion to the bytes it has to
//     stream, so all workgroups carry the same load, and each keeps two stages of its operand stream in flight in registers
//     (the kernel is a stream of ~150 MB through LDS; what has to be hidden is HBM latency, not MFMA time);
//   * every workgroup writes its fp32 partial block once; a second small kernel adds a block's splits IN FIXED ORDER straight into
//     the fp32 master gradient (deterministic -- no float atomics).
// LDS row strides are chosen so that the four rows of a transposed-read block fall into disjoint bank ranges (stride in bytes
// = 64 or 192 mod 256, cdna_hip_programming.md T10 / MI355X_MICROARCH.md LDS).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <atomic>
#include <cstdlib>
#include <cstring>

#include "../../include/bez_sim.h"

namespace {

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

#ifndef BEZ_WGRAD_WAVES
#define BEZ_WGRAD_WAVES 8
#endif
constexpr int WG_WAVES = BEZ_WGRAD_WAVES;   // 8: output blocks of <= 4 x 8 tiles (216 VGPRs); 16: <= 8 x 8 tiles, each operand column read by fewer blocks (<= 128 VGPRs)
constexpr int WG_THREADS = 64 * WG_WAVES;
#ifndef BEZ_WGRAD_TPW
#define BEZ_WGRAD_TPW 4
#endif
constexpr int MAX_TPW = BEZ_WGRAD_TPW;      // output tiles a wave keeps in registers (16 accumulator registers each): 4, or 8 (blocks of <= 8 x 8 tiles on 8 waves)
constexpr int MAX_MT = WG_WAVES * MAX_TPW / 8;   // tile rows of a block: 4 (8 waves x 4 tiles), 8 (16 waves x 4 tiles, or 8 waves x 8 tiles)
constexpr int KT = 64;            // rows of the reduction staged per step
constexpr int MAX_PARTS = 24;
constexpr int MAX_UNITS = 2048 / WG_THREADS;   // 16-byte slots a thread prefetches per operand and stage: 64 rows x 256 columns per operand at most

struct Part {
  const _Float16* g; int ldg, g0, gcols;   // dY (K, ldg): columns [g0, g0 + gcols) = output rows of this block
  const _Float16* x; int ldx, x0, xcols;   // X  (K, ldx): columns [x0, x0 + xcols) = output columns of this block
  int gu, xu;                               // load unit (halfs per load: 8, 4, 2 or 1) of the two operands
  int mt, nt;                               // tiles of 32 along the output rows (1, 2 or 4) and columns (<= 8 * 4 / mt)
  int gs, xs;                               // LDS row strides in halfs
  int splits, wg_begin;                     // K-splits of this block, index of its first workgroup
  long long partial_off;                    // floats: this block's [splits][gcols][xcols] partial images
  float* dst; int dst_ld;                   // the layer's fp32 gradient (out, in) and its row length
};
struct Args {
  Part part[MAX_PARTS];
  int nparts, nstage_total;                 // stages of KT rows in the whole reduction
  float* partial;
  int wg_total, max_block, lds_bytes, grid; // launch geometry (read by bez_ppo_wgrad_run on the host copy); grid = workgroups launched (>= wg_total)
  // XCD-aware placement (round 6): workgroup id -> (part, split).  The hardware deals workgroup ids round-robin over the 8 XCDs (id % 8), each
  // with its own L2; the blocks of one layer re-read each other's operand columns (1.6 x the unique bytes), so the workgroups of a layer
  // that stream the SAME rows at the same time are given ids of one XCD -- the second reader of a slice hits that XCD's L2.  0xff = no work.
  unsigned char map_part[256], map_split[256];
};
static_assert(sizeof(Args) <= BEZ_PPO_WGRAD_PLAN_BYTES, "plan buffer of the C ABI too small");

// one load unit of U halfs (16, 8, 4 or 2 bytes): the width is a template parameter of the kernel body, so the staging code is
// straight-line -- with a run-time width every load sat behind a branch and the compiler waited for it (vmcnt(0)) at the join
template <int U> struct Unit;
template <> struct Unit<8> { using T = uint4; };
template <> struct Unit<4> { using T = uint2; };
template <> struct Unit<2> { using T = uint32_t; };
template <> struct Unit<1> { using T = uint32_t; };   // a single-column dY (the value head): kept in a full register (16-bit register arrays end up in scratch)
template <int U> __device__ __forceinline__ typename Unit<U>::T load_unit(const _Float16* p) {
  if constexpr (U == 1) return (uint32_t)*reinterpret_cast<const uint16_t*>(p);
  else return *reinterpret_cast<const typename Unit<U>::T*>(p);
}
template <int U> __device__ __forceinline__ void store_unit(_Float16* p, typename Unit<U>::T v) {
  if constexpr (U == 1) *reinterpret_cast<uint16_t*>(p) = (uint16_t)v;
  else *reinterpret_cast<typename Unit<U>::T*>(p) = v;
}

// 8 consecutive k (rows kb .. kb+7 of the LDS image) of column `col` for lane (r, h) of an MFMA operand: two transposed reads.
// Lane 4q + p of each 16-lane group supplies the address of row q, columns 4p .. 4p+3 of its block; lane i receives column i.
__device__ __forceinline__ half8 frag_tr(const _Float16* img, int stride, int kb, int colbase, int lane) {
  const int h = lane >> 5, g1 = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
  const _Float16* a = img + (size_t)(kb + 8 * h + q) * stride + colbase + 16 * g1 + 4 * p;
  const fp16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)a);
  const fp16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(a + 4 * (size_t)stride));
  // (bit copies: __fp16 and _Float16 are the same 16 bits; a value conversion would go through fp32, 16 VALU instructions per fragment)
  struct Pair { fp16x4 lo, hi; } pr = {lo, hi};
  return __builtin_bit_cast(half8, pr);
}

// The body for a block whose waves keep TPW tiles each: branch-free inner loops (a tile slot beyond the block's last column
// recomputes that last column and is dropped at the end), so that the compiler can issue every LDS read of a k-step ahead of its
// MFMAs.  Pipeline per stage t: [global loads of t+2 -> registers] [registers of t+1 -> LDS buffer (t+1) & 1] [MFMAs on buffer t & 1]
// [ONE barrier].
template <int TPW, int GU, int XU>
__device__ __forceinline__ void wgrad_body(const Args& A, const Part& P, int split, _Float16* lds) {  // A, P: the plan in global memory (uniform: scalar loads)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int st_begin = (int)((long long)split * A.nstage_total / P.splits), st_end = (int)((long long)(split + 1) * A.nstage_total / P.splits);
  const int buf_halfs = KT * (P.gs + P.xs);
  // tiles of this wave: the waves form an mt x (8 / mt) grid; a wave keeps ONE row of tiles (its A fragment is shared) and every
  // (8 / mt)-th tile column
  const int wm = wave % P.mt, wn = wave / P.mt, ncol = WG_WAVES / P.mt;
  int ncolbase[TPW];
#pragma unroll
  for (int s = 0; s < TPW; ++s) { const int n = wn + s * ncol; ncolbase[s] = 32 * (n < P.nt ? n : P.nt - 1); }
  f32x16 acc[TPW];
#pragma unroll
  for (int s = 0; s < TPW; ++s)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[s][e] = 0.f;
  const int gupr = P.gcols / GU, xupr = P.xcols / XU;   // load units per row
  const int gunits = KT * gupr, xunits = KT * xupr;
  // the (row, unit) of every slot of this thread is the same in every stage: global / LDS offsets once, outside the loop.  A slot
  // beyond the operand's last unit re-reads unit 0 and stores into a per-thread scrap word behind the two buffers: no branches
  const int scrap = 2 * buf_halfs + tid * 8;
  int g_src[MAX_UNITS], g_dst[MAX_UNITS], x_src[MAX_UNITS], x_dst[MAX_UNITS];
#pragma unroll
  for (int i = 0; i < MAX_UNITS; ++i) {
    const int idx = tid + i * WG_THREADS;
    int row = idx / gupr, c = idx - row * gupr;
    g_src[i] = idx < gunits ? row * P.ldg + P.g0 + c * GU : P.g0; g_dst[i] = idx < gunits ? row * P.gs + c * GU : scrap - 0;
    row = idx / xupr; c = idx - row * xupr;
    x_src[i] = idx < xunits ? row * P.ldx + P.x0 + c * XU : P.x0; x_dst[i] = idx < xunits ? KT * P.gs + row * P.xs + c * XU : scrap;
  }
  // two stages of the operand stream live in registers (ring of depth 2) and two in LDS: while stage t is multiplied out of one
  // LDS buffer, stage t+1 is being written into the other and the loads of stage t+2 are in flight
  constexpr int RING = TPW > 4 ? 1 : 2;
  typename Unit<GU>::T gr[RING][MAX_UNITS];
  typename Unit<XU>::T xr[RING][MAX_UNITS];
  auto prefetch = [&](int st, typename Unit<GU>::T (&g4)[MAX_UNITS], typename Unit<XU>::T (&x4)[MAX_UNITS]) {
    const long long k0 = (long long)st * KT;
    const _Float16* gb = P.g + k0 * P.ldg;
    const _Float16* xb = P.x + k0 * P.ldx;
#pragma unroll
    for (int i = 0; i < MAX_UNITS; ++i) { g4[i] = load_unit<GU>(gb + g_src[i]); x4[i] = load_unit<XU>(xb + x_src[i]); }
  };
  auto stage = [&](_Float16* buf, const typename Unit<GU>::T (&g4)[MAX_UNITS], const typename Unit<XU>::T (&x4)[MAX_UNITS]) {
    // (buffer 1 lies buf_halfs behind buffer 0; the scrap offsets were computed relative to buffer 0)
    const int shift = (int)(buf - lds);
#pragma unroll
    for (int i = 0; i < MAX_UNITS; ++i) {
      store_unit<GU>(lds + (g_dst[i] >= 2 * buf_halfs ? g_dst[i] : g_dst[i] + shift), g4[i]);
      store_unit<XU>(lds + (x_dst[i] >= 2 * buf_halfs ? x_dst[i] : x_dst[i] + shift), x4[i]);
    }
  };
  auto multiply = [&](const _Float16* buf) {
    const _Float16* G = buf;
    const _Float16* X = buf + (size_t)KT * P.gs;
#pragma unroll
    for (int kk = 0; kk < KT / 16; ++kk) {
      const half8 a = frag_tr(G, P.gs, kk * 16, 32 * wm, lane);   // every lane takes part (the transposed read needs EXEC = all ones)
      if constexpr (TPW <= 4) {
        half8 b[TPW];
#pragma unroll
        for (int s = 0; s < TPW; ++s) b[s] = frag_tr(X, P.xs, kk * 16, ncolbase[s], lane);
#pragma unroll
        for (int s = 0; s < TPW; ++s) acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[s], acc[s], 0, 0, 0);
      } else {
        // eight tiles per wave (128 accumulator registers): the B fragments in groups of four, and no LDS read of the next k-step above
        // this one's MFMAs -- hoisted, four k-steps' fragments cost 144 registers the accumulators have left no room for
#pragma unroll
        for (int h = 0; h < TPW; h += 4) {
          half8 b[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) b[s] = frag_tr(X, P.xs, kk * 16, ncolbase[h + s], lane);
#pragma unroll
          for (int s = 0; s < 4; ++s) acc[h + s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[s], acc[h + s], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  };
  _Float16* buf0 = lds;
  _Float16* buf1 = lds + buf_halfs;
  if constexpr (TPW > 4) {
    // eight tiles per wave: the 128 accumulator registers leave room for ONE stage of the operand stream in registers -- its loads are
    // issued before the MFMAs of the stage in LDS and land behind them (64 rows x <= 512 columns in flight per CU)
    if (st_begin < st_end) { prefetch(st_begin, gr[0], xr[0]); stage(buf0, gr[0], xr[0]); }
    __syncthreads();
    for (int st = st_begin; st < st_end; ++st) {
      _Float16* cur = ((st - st_begin) & 1) ? buf1 : buf0;
      _Float16* nxt = ((st - st_begin) & 1) ? buf0 : buf1;
      if (st + 1 < st_end) prefetch(st + 1, gr[0], xr[0]);
      multiply(cur);
      if (st + 1 < st_end) stage(nxt, gr[0], xr[0]);
      __syncthreads();
    }
  } else {
  if (st_begin < st_end) prefetch(st_begin, gr[0], xr[0]);
  if (st_begin + 1 < st_end) prefetch(st_begin + 1, gr[1], xr[1]);
  if (st_begin < st_end) stage(buf0, gr[0], xr[0]);
  if (st_begin + 2 < st_end) prefetch(st_begin + 2, gr[0], xr[0]);
  __syncthreads();
  for (int st = st_begin; st < st_end; st += 2) {   // unrolled by the ring depth: register set and LDS buffer of a stage are compile-time choices
    if (st + 1 < st_end) stage(buf1, gr[1], xr[1]);
    if (st + 3 < st_end) prefetch(st + 3, gr[1], xr[1]);
    multiply(buf0);
    __syncthreads();
    if (st + 1 < st_end) {
      if (st + 2 < st_end) stage(buf0, gr[0], xr[0]);
      if (st + 4 < st_end) prefetch(st + 4, gr[0], xr[0]);
      multiply(buf1);
      __syncthreads();
    }
  }
  }
  // partial block: acc register e of lane l = dW[g0 + 32 wm + (e & 3) + 8 (e >> 2) + 4 (l >> 5)][x0 + 32 n + (l & 31)]
  float* out = A.partial + P.partial_off + (long long)split * P.gcols * P.xcols;
#pragma unroll
  for (int s = 0; s < TPW; ++s) {
    const int n = wn + s * ncol;
    if (n >= P.nt) continue;
    const int ci = 32 * n + (lane & 31);
    if (ci >= P.xcols) continue;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ri = 32 * wm + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
      if (ri < P.gcols) out[(size_t)ri * P.xcols + ci] = acc[s][e];
    }
  }
}

// The plan (output blocks, their K-splits, LDS strides) lives in DEVICE memory, written once when the caller's tensors are known: as a
// by-value kernel argument the table had to be indexed dynamically, and with more than ~20 instantiated variants the compiler
// copied all 2.5 KB of it into scratch first.
__global__ void __launch_bounds__(WG_THREADS) wgrad_kernel(const Args* __restrict__ Ap) {
  extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
  const Args& A = *Ap;
  const int pi = A.map_part[blockIdx.x];
  if (pi == 0xff) return;   // (an id the placement left empty)
  const Part& P = A.part[pi];
  const int split = A.map_split[blockIdx.x];
  const int tpw = (P.nt + WG_WAVES / P.mt - 1) / (WG_WAVES / P.mt);   // tile columns per wave of this block
  const int variant = (tpw <= 1 ? 0 : (tpw <= 4 ? 2 : 3)) * 16 + (P.gu == 8 ? 0 : (P.gu == 4 ? 1 : (P.gu == 2 ? 2 : 3))) * 4 + (P.xu == 8 ? 0 : (P.xu == 4 ? 1 : 2));
  // every (tiles per wave, dY unit, X unit) combination is its own straight-line instantiation
#define WG_CASE(T, TI, GUV, GI, XUV, XI) case (TI * 16 + GI * 4 + XI): wgrad_body<T, GUV, XUV>(A, P, split, lds); break;
#define WG_X(T, TI, GUV, GI) WG_CASE(T, TI, GUV, GI, 8, 0) WG_CASE(T, TI, GUV, GI, 4, 1) WG_CASE(T, TI, GUV, GI, 2, 2)
#define WG_G(T, TI) WG_X(T, TI, 8, 0) WG_X(T, TI, 4, 1) WG_X(T, TI, 2, 2)
#if BEZ_WGRAD_TPW > 4
  switch (variant) { WG_G(1, 0) WG_G(4, 2) WG_G(8, 3) WG_CASE(1, 0, 1, 3, 8, 0) WG_CASE(1, 0, 1, 3, 4, 1) WG_CASE(1, 0, 1, 3, 2, 2) default: break; }
#else
  switch (variant) { WG_G(1, 0) WG_G(4, 2) WG_CASE(1, 0, 1, 3, 8, 0) WG_CASE(1, 0, 1, 3, 4, 1) WG_CASE(1, 0, 1, 3, 2, 2) default: break; }
#endif
#undef WG_G
#undef WG_X
#undef WG_CASE
}

// the splits of every block, added in fixed order into the layer's gradient: grid = (elements of the largest block / 256, blocks)
__global__ void wgrad_reduce_kernel(const Args* __restrict__ Ap, int accumulate) {
  const Args& A = *Ap;
  const Part& P = A.part[blockIdx.y];
  const int i = blockIdx.x * blockDim.x + threadIdx.x, n = P.gcols * P.xcols;
  if (i >= n) return;
  const int r = i / P.xcols, c = i - r * P.xcols;
  float* d = P.dst + (size_t)(P.g0 + r) * P.dst_ld + P.x0 + c;
  const float* p = A.partial + P.partial_off + i;
  float s = accumulate ? *d : 0.f;
  int k = 0;
  for (; k + 8 <= P.splits; k += 8) {   // eight loads in flight, added in their fixed order: deterministic
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(long long)(k + u) * n];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; k < P.splits; ++k) s += p[(long long)k * n];
  *d = s;
}

// ---- every second-stage reduction of a minibatch step's gradient in ONE launch (round 3: three): the split-K partial images of the
// five weight gradients (wgrad_kernel above), the per-workgroup column sums of the bias gradients (policy_backward_kernel,
// csrc/bez_policy.hip) and the per-workgroup sums of the loss kernel (d loss / d log-sigma, the five loss / KL / entropy sums:
// ppo_loss_kernel, csrc/bez_ppo.hip).  All three are fixed-order sums of partials -- no float atomics, bit-reproducible -- and each
// output element has exactly one writer, so with accumulate == 0 the launch WRITES the whole flat gradient (weights, biases, log-sigma)
// and the statistics: the caller needs no clear in front of the step.
constexpr int RA_MAXL = 8;
// the bias-gradient blocks of the launch: 256 threads as RA_BROWS row lanes x RA_BCOLS columns.  512 per-workgroup partials (32768 rows) on 4 row
// lanes were 128 dependent-latency loads per thread, 16 rounds of 8 in flight -- the longest workgroups of the launch by far.
#ifndef BEZ_RA_BCOLS
#define BEZ_RA_BCOLS 16
#endif
constexpr int RA_BCOLS = BEZ_RA_BCOLS, RA_BROWS = 256 / RA_BCOLS;
static_assert(RA_BROWS % 4 == 0 && RA_BROWS * RA_BCOLS == 256, "row lanes in fours");
struct BiasReduce { const float* partial; int prow, ptotal, nhid, num_actions, nwg; int poff[RA_MAXL]; float* bgrad[RA_MAXL]; float* bmu; float* bv; };
struct LossReduce { const float* scratch; int A; unsigned int nblocks; float* grad_logstd; float* stats; };
__global__ __launch_bounds__(256) void grad_reduce_all_kernel(const Args* __restrict__ Ap, int accumulate, int wx, int nwb, BiasReduce B, int nbb, LossReduce Ls,
                                                              float* __restrict__ normpart) {
  __shared__ float sh[RA_BROWS][RA_BCOLS];
  __shared__ float nrm[2][4];
  const int b = blockIdx.x, tid = threadIdx.x;
  float mine = 0.f;      // the gradient element this thread wrote (0: none), for the block's share of the squared gradient norm
  if (b < nwb) {   // weight gradients: block (b % wx) of part (b / wx)
    const Args& A = *Ap;
    const Part& P = A.part[b / wx];
    const int i = (b % wx) * 256 + tid, n = P.gcols * P.xcols;
    if (i < n) {
      const int r = i / P.xcols, c = i - r * P.xcols;
      float* d = P.dst + (size_t)(P.g0 + r) * P.dst_ld + P.x0 + c;
      const float* p = A.partial + P.partial_off + i;
      float s = accumulate ? *d : 0.f;
      int k = 0;
      for (; k + 8 <= P.splits; k += 8) {   // eight loads in flight, added in their fixed order (16 / 32 in flight: no faster)
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(long long)(k + u) * n];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
      }
      for (; k < P.splits; ++k) s += p[(long long)k * n];
      *d = s;
      mine = s;
    }
  } else if (b < nwb + nbb) {   // bias gradients: RA_BROWS row lanes x RA_BCOLS columns; a row lane sums every RA_BROWS-th workgroup's partial
    const int l = tid % RA_BCOLS, rl = tid / RA_BCOLS;
    const int c = (b - nwb) * RA_BCOLS + l;
    const int ncol = B.ptotal + B.num_actions + 1;
    float s = 0.f;
    if (c < ncol) {
#pragma unroll 8
      for (int w = rl; w < B.nwg; w += RA_BROWS) s += B.partial[(size_t)w * B.prow + c];
    }
    sh[rl][l] = s;
    __syncthreads();
    if (rl == 0 && c < ncol) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < RA_BROWS; q += 4) t += (sh[q][l] + sh[q + 1][l]) + (sh[q + 2][l] + sh[q + 3][l]);
      float* d;
      if (c >= B.ptotal) { const int k = c - B.ptotal; d = k < B.num_actions ? &B.bmu[k] : &B.bv[0]; }
      else {
        int L = 0;
        while (L + 1 < B.nhid && c >= B.poff[L + 1]) ++L;
        d = &B.bgrad[L][c - B.poff[L]];
      }
      if (accumulate) t += *d;
      *d = t;
      mine = t;
    }
  } else {   // loss sums: one wave per column; lane l adds workgroups l, l + 64, ..., then the butterfly
    const int c = b - nwb - nbb;
    if (tid < 64) {
      const float* col = Ls.scratch + 2 + (size_t)c * Ls.nblocks;
      float acc = 0.f;
#pragma unroll 8
      for (unsigned int q = tid; q < Ls.nblocks; q += 64) acc += col[q];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
      if (tid == 0) {
        float* d = c < Ls.A ? &Ls.grad_logstd[c] : &Ls.stats[c - Ls.A];
        if (accumulate) acc += *d;
        *d = acc;
        if (c < Ls.A) mine = acc;   // (the statistics are not part of the gradient)
      }
    }
  }
  if (!normpart) return;
  // this block's share of sum g^2 and of the non-finite count (of the STILL-SCALED gradient), fixed order: bez_ppo_adam_step adds the
  // blocks' shares in order instead of reading the whole gradient again
  float s2 = mine * mine, bad = fabsf(mine) <= 3.4028234e38f ? 0.f : 1.f;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s2 += __shfl_xor(s2, o, 64); bad += __shfl_xor(bad, o, 64); }
  if ((tid & 63) == 0) { nrm[0][tid >> 6] = s2; nrm[1][tid >> 6] = bad; }
  __syncthreads();
  if (tid == 0) {
    normpart[2 * b] = (nrm[0][0] + nrm[0][1]) + (nrm[0][2] + nrm[0][3]);
    normpart[2 * b + 1] = (nrm[1][0] + nrm[1][1]) + (nrm[1][2] + nrm[1][3]);
  }
}

int lds_stride(int cols) {   // halfs: >= cols rounded up to whole tiles, stride bytes = 64 or 192 (mod 256)
  int c = (cols + 31) / 32 * 32;
  return ((c / 32) & 1) ? c : c + 32;
}
int load_unit_of(const void* base, int ld, int c0, int cols) {
  for (int u = 8; u > 1; u >>= 1)
    if (cols % u == 0 && c0 % u == 0 && ld % u == 0 && (reinterpret_cast<uintptr_t>(base) % (2 * u)) == 0) return u;
  return 1;
}

}  // namespace

extern "C" {

/* Plan: the output blocks of `nlayers` weight gradients dW_L (+)= dY_L^T X_L (dY_L (rows, out_L), X_L (rows, in_L) fp16 row-major,
 * dW_L (out_L, in_L) fp32), their K-splits in proportion to the bytes each streams (~250 workgroups in all, within a scratch of
 * nsplit * sum(out_L * in_L) floats at partial_dev), LDS strides and load widths -- written into plan_host (BEZ_PPO_WGRAD_PLAN_BYTES
 * bytes).  The caller copies the plan to device memory ONCE and then calls bez_ppo_wgrad_run per step.  rows % 64 must be 0;
 * -3 = shapes the kernel does not take (the caller keeps its GEMM path). */
int bez_ppo_wgrad_plan(const void* const* dy_f16_dev, const void* const* x_f16_dev, const int32_t* out_features, const int32_t* in_features,
                       float* const* dw_dev, int32_t nlayers, int64_t rows, int32_t nsplit, float* partial_dev, void* plan_host) {
  if (!plan_host || nlayers < 1 || nlayers > 8 || nsplit < 1 || rows <= 0 || rows % KT != 0) return -3;
  Args A;
  std::memset(&A, 0, sizeof(A));
  int np = 0;
  size_t lds_bytes = 0;
  long long weight[MAX_PARTS], total_weight = 0, budget = 0;
  int layer_of[MAX_PARTS];
  // fixed part of a stage (barrier, load latency) in column equivalents: measured, 32768 rows of bez_kickPPO.yaml's five layers:
  // 0 -> 61 us (the heads' few workgroups run 46 short stages each while the wide blocks finish 13), 64 -> 48, 128 -> 46, 512 -> 46
  const int stage_cost = 128;
  for (int L = 0; L < nlayers; ++L) {
    const int O = out_features[L], I = in_features[L];
    if (O < 1 || I < 1) return -3;
    budget += (long long)nsplit * O * I;                           // the caller's scratch: nsplit images of every gradient
    const int mtiles = (O + 31) / 32, ntiles = (I + 31) / 32;
    const int mt = (mtiles > 4 && MAX_MT >= 8) ? 8 : (mtiles >= 4 ? 4 : (mtiles >= 2 ? 2 : 1));       // tile rows per block
    int ntmax = (WG_WAVES / mt) * MAX_TPW;                         // tile columns a block can keep in registers ...
    if (ntmax > 8) ntmax = 8;                                      // ... and stage with MAX_UNITS 16-byte slots per thread (256 columns)
    const int nblocks_n = (ntiles + ntmax - 1) / ntmax;
    const int nt_even = (ntiles + nblocks_n - 1) / nblocks_n;      // balanced column blocks
    for (int m0 = 0; m0 < mtiles; m0 += mt) {
      for (int n0 = 0; n0 < ntiles; n0 += nt_even) {
        if (np >= MAX_PARTS) return -3;
        Part& P = A.part[np];
        P.g = static_cast<const _Float16*>(dy_f16_dev[L]); P.ldg = O; P.g0 = 32 * m0; P.gcols = (O - P.g0 < 32 * mt) ? O - P.g0 : 32 * mt;
        P.x = static_cast<const _Float16*>(x_f16_dev[L]); P.ldx = I; P.x0 = 32 * n0;
        const int ntp = (ntiles - n0 < nt_even) ? ntiles - n0 : nt_even;
        P.xcols = (I - P.x0 < 32 * ntp) ? I - P.x0 : 32 * ntp;
        P.mt = mt; P.nt = ntp;
        P.gu = load_unit_of(P.g, P.ldg, P.g0, P.gcols); P.xu = load_unit_of(P.x, P.ldx, P.x0, P.xcols);
        // single-half units: only for dY, and only for a block of one tile row whose waves keep one tile each (the value head)
        if (P.xu == 1 || (P.gu == 1 && (mt != 1 || (ntp + WG_WAVES - 1) / WG_WAVES > 1))) return -3;
        if (KT * (P.gcols / P.gu) > MAX_UNITS * WG_THREADS || KT * (P.xcols / P.xu) > MAX_UNITS * WG_THREADS) return -3;
        P.gs = lds_stride(32 * mt); P.xs = lds_stride(32 * ntp);
        P.dst = dw_dev[L]; P.dst_ld = I;
        const size_t need = 2 * (size_t)KT * (P.gs + P.xs) * sizeof(_Float16) + WG_THREADS * 16;   // two LDS buffers + a scrap slot per thread
        if (need > lds_bytes) lds_bytes = need;
        layer_of[np] = L;
        weight[np] = P.gcols + P.xcols + stage_cost;               // time of one stage of this block: columns streamed + a fixed part (barrier, load latency)
        total_weight += weight[np];
        ++np;
      }
    }
  }
  if (lds_bytes > 160 * 1024) return -3;
  // K-splits per block in proportion to its stream, ~250 workgroups in all (one round on 256 CUs), within the caller's scratch
  const int nstage = (int)(rows / KT);
  // 254: with the rounding below 255 workgroups for bez_kickPPO.yaml's five layers -- every workgroup fewer is ~0.2 us more for the others (248 / 250 /
  // 255 workgroups: 41.1 / 40.2-40.6 / 39.7-39.8 us on one box, tools/wgrad_target_ab.sh); a 257th would be a second round: trimmed below
  static const int target_wgs = [] { const char* e = std::getenv("BEZ_WGRAD_TARGET_WGS"); const int v = e ? std::atoi(e) : 0; return v > 0 ? v : 254; }();   // (A/B knob)
  constexpr int max_wgs = 256;   // one workgroup per CU, ONE round
  long long used = 0;
  int wg = 0, max_block = 0;
  for (int i = 0; i < np; ++i) {
    Part& P = A.part[i];
    long long sp = (weight[i] * target_wgs + total_weight / 2) / total_weight;
    if (sp < 1) sp = 1;
    if (sp > nstage) sp = nstage;
    if (sp > nsplit) sp = nsplit;   // the caller's scratch holds nsplit images of every gradient
    P.splits = (int)sp; P.wg_begin = wg; P.partial_off = used;
    wg += P.splits;
    used += (long long)P.splits * P.gcols * P.xcols;
    if (P.gcols * P.xcols > max_block) max_block = P.gcols * P.xcols;
  }
  while (wg > max_wgs) {   // the roundings added up to more than one round: take the split back where a workgroup has the fewest stages to gain
    int k = -1;
    for (int i = 0; i < np; ++i) if (A.part[i].splits > 1 && (k < 0 || A.part[i].splits > A.part[k].splits)) k = i;
    if (k < 0) break;
    --A.part[k].splits; --wg;
  }
  used = 0; wg = 0; max_block = 0;
  for (int i = 0; i < np; ++i) {
    Part& P = A.part[i];
    P.wg_begin = wg; P.partial_off = used;
    wg += P.splits;
    used += (long long)P.splits * P.gcols * P.xcols;
    if (P.gcols * P.xcols > max_block) max_block = P.gcols * P.xcols;
  }
  if (used > budget) return -3;
  A.nparts = np; A.nstage_total = nstage; A.partial = partial_dev;
  A.wg_total = wg; A.max_block = max_block; A.lds_bytes = (int)lds_bytes;
  {  // placement: per layer the workgroups in the order of the rows they start at (ties: by block), four consecutive ones to one XCD
    struct W { int part, split; long long start; };
    W order[256];
    int n = 0;
    for (int L = 0; L < nlayers; ++L) {
      const int first = n;
      for (int i = 0; i < np; ++i) if (layer_of[i] == L)
        for (int k = 0; k < A.part[i].splits; ++k) order[n++] = W{i, k, (long long)k * nstage / A.part[i].splits};
      for (int a = first + 1; a < n; ++a) {   // insertion sort (<= 256 entries): by first stage, then block
        const W w = order[a];
        int b = a - 1;
        while (b >= first && (order[b].start > w.start || (order[b].start == w.start && order[b].part > w.part))) { order[b + 1] = order[b]; --b; }
        order[b + 1] = w;
      }
    }
    static const int xcd_aware = [] { const char* e = std::getenv("BEZ_WGRAD_XCD"); const int v = e ? std::atoi(e) : 4; return (v == 0 || v == 1 || v == 2 || v == 4 || v == 8 || v == 16 || v == 32) ? v : 4; }();   // (A/B knob: consecutive workgroups per XCD; 0 = ids in plan order)
    std::memset(A.map_part, 0xff, sizeof(A.map_part)); std::memset(A.map_split, 0, sizeof(A.map_split));
    int grid = 0;
    for (int l = 0; l < n; ++l) {
      int id = l;
      if (xcd_aware) { const int G = xcd_aware, g = l / G, q = l % G, x = g % 8, slot = (g / 8) * G + q; id = slot * 8 + x; }   // G consecutive ones to one XCD
      if (id >= 256) return -3;
      A.map_part[id] = (unsigned char)order[l].part; A.map_split[id] = (unsigned char)order[l].split;
      if (id + 1 > grid) grid = id + 1;
    }
    A.grid = grid;
  }
  std::memset(plan_host, 0, BEZ_PPO_WGRAD_PLAN_BYTES);
  std::memcpy(plan_host, &A, sizeof(A));
  return 0;
}

/* Run a plan: plan_host = the buffer bez_ppo_wgrad_plan filled (launch geometry), plan_dev = its copy in device memory (read by the
 * kernels).  accumulate != 0 adds to the gradients.  Two launches: the split-K MFMA kernel and the fixed-order reduction. */
int bez_ppo_wgrad_run(const void* plan_host, const void* plan_dev, int32_t accumulate, void* stream_) {
  if (!plan_host || !plan_dev) return -1;
  const Args* H = static_cast<const Args*>(plan_host);
  const Args* D = static_cast<const Args*>(plan_dev);
  hipStream_t stream = (hipStream_t)stream_;
  // the attribute is per device (a process may drive several) and cheap to set: per device once, guarded by an atomic bit mask
  static std::atomic<uint64_t> attr_devices{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) return -2;
  const uint64_t bit = dev < 64 ? (1ull << dev) : 0;  // beyond 64 devices: set it every time
  if (!bit || !(attr_devices.load(std::memory_order_acquire) & bit)) {
    if (hipFuncSetAttribute((const void*)wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -2;
    attr_devices.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL(wgrad_kernel, dim3(H->grid), dim3(WG_THREADS), (size_t)H->lds_bytes, stream, D);
  if (accumulate != 2)  // 2: the partial images only -- bez_ppo_grad_reduce_all adds them together with the step's other reductions
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((H->max_block + 255) / 256, H->nparts), dim3(256), 0, stream, D, (int)accumulate);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

static int reduce_all_geometry(const Args* H, int32_t num_hidden, const int32_t* hidden_width, int32_t num_actions, int* wx, int* nwb, int* nbb, int* nlb) {
  if (!H || num_hidden <= 0 || num_hidden > RA_MAXL || !hidden_width || num_actions <= 0 || num_actions > 31) return -1;
  int tot = 0;
  for (int i = 0; i < num_hidden; ++i) tot += hidden_width[i];
  *wx = (H->max_block + 255) / 256; *nwb = *wx * H->nparts; *nbb = (tot + num_actions + 1 + RA_BCOLS - 1) / RA_BCOLS; *nlb = num_actions + 5;
  return 0;
}
/* workgroups of bez_ppo_grad_reduce_all for this plan / network = pairs of floats it writes to norm_parts_dev */
int bez_ppo_grad_reduce_blocks(const void* plan_host, int32_t num_hidden, const int32_t* hidden_width, int32_t num_actions) {
  int wx, nwb, nbb, nlb;
  if (reduce_all_geometry(static_cast<const Args*>(plan_host), num_hidden, hidden_width, num_actions, &wx, &nwb, &nbb, &nlb)) return -1;
  return nwb + nbb + nlb;
}
int bez_ppo_grad_reduce_all(const void* plan_host, const void* plan_dev, const float* bias_partial_dev, int64_t rows, int32_t num_hidden,
                            const int32_t* hidden_width, int32_t num_actions, float* const* bias_grad_dev, float* mu_bias_grad_dev,
                            float* value_bias_grad_dev, const float* loss_scratch_dev, int64_t loss_rows, float* grad_logstd_dev, float* stats_dev,
                            int32_t accumulate, float* norm_parts_dev, void* stream) {
  if (!plan_host || !plan_dev || !bias_partial_dev || rows <= 0 || num_hidden <= 0 || num_hidden > RA_MAXL || !hidden_width || num_actions <= 0 ||
      num_actions > 31 || !bias_grad_dev || !mu_bias_grad_dev || !value_bias_grad_dev || !loss_scratch_dev || loss_rows <= 0 || !grad_logstd_dev || !stats_dev) return -1;
  const Args* H = static_cast<const Args*>(plan_host);
  BiasReduce B{};
  B.partial = bias_partial_dev; B.nhid = num_hidden; B.num_actions = num_actions; B.nwg = (int)((rows + 63) / 64);  // policy_backward_kernel: 64 rows per workgroup
  int off = 0;
  for (int i = 0; i < num_hidden; ++i) { if (!bias_grad_dev[i]) return -1; B.poff[i] = off; B.bgrad[i] = bias_grad_dev[i]; off += hidden_width[i]; }
  B.ptotal = off; B.prow = off + 32; B.bmu = mu_bias_grad_dev; B.bv = value_bias_grad_dev;
  LossReduce Ls{loss_scratch_dev, (int)num_actions, (unsigned int)((loss_rows + 63) / 64), grad_logstd_dev, stats_dev};  // ppo_loss_kernel: 64 rows per workgroup
  int wx, nwb, nbb, nlb;
  if (reduce_all_geometry(H, num_hidden, hidden_width, num_actions, &wx, &nwb, &nbb, &nlb)) return -1;
  hipLaunchKernelGGL(grad_reduce_all_kernel, dim3(nwb + nbb + nlb), dim3(256), 0, (hipStream_t)stream, static_cast<const Args*>(plan_dev), (int)(accumulate != 0),
                     wx, nwb, B, nbb, Ls, accumulate ? nullptr : norm_parts_dev);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // extern "C"

// bez_policy.hip -- the rollout's policy forward pass as ONE gfx950 kernel (C ABI: bez_ppo_policy_forward, include/bez_sim.h).
//
// What it replaces: per env step of the PPO rollout, rl_games' get_action_values runs the observation normaliser, five
// Linear layers and three ELUs as ~14 launches of 5-25 us each on a (4096 x 54) batch -- tiny GEMMs that leave the chip idle.
// Here one workgroup (8 waves) carries 64 envs through the whole MLP: activations live in LDS (two 64 x 416 fp16 buffers),
// the fp16 weights (247 KB for 54-400-200-100-(18+1), L2-resident) stream from global memory straight into MFMA B fragments,
// products are v_mfma_f32_32x32x16_f16 with fp32 accumulators, bias + fp16 rounding + ELU + fp16 rounding in the epilogue (the
// arithmetic of torch's fp16 Linear / ELU on the explicit-fp16 path of a2c_continuous.py).  The same kernel serves the rollout
// (mode 1: sampling and the rollout-buffer rows fused behind it) and the forward half of a PPO minibatch step (mode 2: every ELU
// output and the normalised input are also written to HBM, which is all the backward pass needs); the backward pass stays
// PyTorch-ROCm GEMMs + the reduction kernels of bez_ppo.hip.
//
// Fragment maps (MI355X guide, checked by tests/test_gpu_ppo_fused.py against torch on asymmetric data):
//   A (32 x 16): lane l holds A[row l & 31][k = 8 (l >> 5) + j], j = 0..7      -> the activations of 32 envs
//   B (16 x 32): lane l holds B[k = 8 (l >> 5) + j][col l & 31] = W[col][k]    -> 8 consecutive fp16 of one weight row
//   C (32 x 32): lane l holds C[row (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col l & 31], r = 0..15
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>
#include <cstring>

#include <type_traits>

#include "../../include/bez_sim.h"
#include "bez_dr_noise.h"
#include "bez_dr_step.h"
#include "bez_ppo_loss.h"

namespace {

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using half2v = __attribute__((ext_vector_type(2))) _Float16;
using half4v = __attribute__((ext_vector_type(4))) _Float16;

constexpr int PF_ROWS = 64;         // envs per workgroup
constexpr int PF_MAXW = 416;        // widest layer the LDS buffers hold (13 blocks of 32 columns)
constexpr int PF_LD = PF_MAXW + 8;  // row stride in halfs: 848 B = 53 x 16 B (16-byte fragment reads, conflict-free row walk)
constexpr int PF_MAXL = 6;          // hidden layers supported
constexpr int PF_WAVES = 8;         // waves per workgroup: the column blocks of a layer are dealt round-robin

struct PolicyArgs {
  const float* obs; int64_t n; int d_in;
  const double* mean; const double* var; float eps;  // RunningMeanStd of the observations (mean == null: obs are used as they are)
  int nhid;
  const _Float16* w[PF_MAXL]; const _Float16* b[PF_MAXL]; int width[PF_MAXL];  // hidden layers: (width[i], width[i-1] or d_in)
  const _Float16* w_mu; const _Float16* b_mu; int num_actions;                // mu head (num_actions, width[nhid-1])
  const _Float16* w_val; const _Float16* b_val;                               // value head (1, width[nhid-1])
  float* mu; float* value;
  // rollout step (ROLL): everything between the forward pass and the env step, fused behind it (bez_ppo_rollout_pre's work)
  const float* logstd; const float* noise; const float* dones; const double* vmean; const double* vvar; float veps;
  float* mb_obs; float* mb_dones; float* mb_mu; float* mb_val; float* act; float* act_env; float* neglogp; float* sigma;
  // ... and, in front of it, the bookkeeping of the PREVIOUS env step (bez_ppo_rollout_post's work; post.rew == null: none)
  BezPpoRolloutPost post;
  // ... and, behind it, the env's action-noise lambda of the domain randomisation (vec_task.py:586-592; an.snap_dev == null: none)
  BezPpoActionNoise an;
  // ... and, as ONE EXTRA workgroup of the launch, the randomisation of the coming env step (bez_sim_dr_step_args; dr_on == 0: none)
  int dr_on; bez::dr::DrArgs dr;
  // row strides (floats) of the rollout rows this launch writes: mb_obs; mb_mu / act / sigma; neglogp (BezPpoRolloutLayout; contiguous: d_in, A, 1)
  int64_t ld_obs, ld_act, ld_one;
  // training forward (mode 2): what the backward pass needs -- the fp16 input of the first Linear and every ELU output, row-major
  _Float16* x0_out; _Float16* act_out[PF_MAXL];
  int packed;  // weights are fragment-major copies (see gemm_packed): w[L] and w_mu (= the packed [mu; value] block)
#ifdef BEZ_PF_STAMPS
  unsigned long long* stamps;  // diagnostic build only: s_memtime of workgroup 0 / thread 0 at the phase boundaries
#endif
};
#ifdef BEZ_PF_STAMPS
#define PF_STAMP(k) do { if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0) a.stamps[k] = __builtin_amdgcn_s_memtime(); } while (0)
__device__ unsigned long long* g_pf_wave_stamps = nullptr;   // (layer tag, wave, phase) stamps of workgroup 0: entries [tag * 64 + wave * 4 + phase]
#define PF_WSTAMP(tag, wave, ph) do { if (g_pf_wave_stamps && blockIdx.x == 0 && (threadIdx.x & 63) == 0) g_pf_wave_stamps[(tag) * 64 + (wave) * 4 + (ph)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PF_STAMP(k) do { } while (0)
#define PF_WSTAMP(tag, wave, ph) do { } while (0)
#endif

// rows [0, nrow) x columns [0, cols) of an LDS activation tile -> global (row-major, `cols` halfs per row, cols even): half2 per
// lane, consecutive lanes on consecutive columns (256 B per wave instruction)
template <int LD>
__device__ __forceinline__ void store_tile(const _Float16 (*src)[LD], _Float16* dst, int64_t row0, int nrow, int cols, int tid, int nw = PF_WAVES) {
  // waves over rows, lanes over columns: no index division (a flat index over (row, column pair) costs an integer division by a run-time
  // width per element -- ~1.9 k of the training forward's 3.4 k vector instructions per wave were that); 8 bytes per lane where the width allows
  const int lane = tid & 63, wave = tid >> 6;
  if ((cols & 3) == 0) {
    const int c4 = cols >> 2;
    for (int rr = wave; rr < nrow; rr += nw) {
      _Float16* drow = dst + (row0 + rr) * cols;
      for (int c = lane; c < c4; c += 64) *reinterpret_cast<uint2*>(drow + 4 * c) = *reinterpret_cast<const uint2*>(&src[rr][4 * c]);
    }
  } else {
    const int c2 = cols >> 1;
    for (int rr = wave; rr < nrow; rr += nw) {
      _Float16* drow = dst + (row0 + rr) * cols;
      for (int c = lane; c < c2; c += 64) *reinterpret_cast<uint32_t*>(drow + 2 * c) = *reinterpret_cast<const uint32_t*>(&src[rr][2 * c]);
    }
  }
}

struct __attribute__((packed, aligned(4))) H8 { _Float16 v[8]; };

// 8 consecutive fp16 of a weight row starting at column k0; columns >= cols read as zero; rows are only 4-byte aligned in general
__device__ __forceinline__ half8 load_w8(const _Float16* row, int k0, int cols) {
  half8 f;
  if (k0 + 8 <= cols && (cols & 1) == 0) {
    const H8 t = *reinterpret_cast<const H8*>(row + k0);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = t.v[j];
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (k0 + j < cols) ? row[k0 + j] : (_Float16)0.f;
  }
  return f;
}

// acc0 / acc1 (rows 0..31 / 32..63 of the workgroup) += src[:, 0:in] * wrow[0:in] for this lane's column (wrow = its weight row,
// live = the column exists).  ROW-MAJOR weights: the fallback of callers without fragment-major copies (the compiler sinks these loads next to
// their MFMAs whatever the grouping below says -- gemm_packed shows what it takes to keep them ahead).
template <int LD>
__device__ __forceinline__ void gemm_col_block(const _Float16 (*src)[LD], const _Float16* wrow, bool live, int in, int r, int h, f32x16& acc0, f32x16& acc1) {
  const int full = ((in & 1) == 0) ? (in >> 4) : 0;  // k-steps whose 16 columns all exist (and whose rows are 4-byte aligned)
  const int ksteps = (in + 15) >> 4;
  int ks = 0;
  auto group = [&](auto G) {
    constexpr int g = decltype(G)::value;
    half8 bf[g];
#pragma unroll
    for (int u = 0; u < g; ++u) {
      if (live) {
        const H8 t = *reinterpret_cast<const H8*>(wrow + (ks + u) * 16 + 8 * h);
#pragma unroll
        for (int j = 0; j < 8; ++j) bf[u][j] = t.v[j];
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) bf[u][j] = (_Float16)0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < g; ++u) {
      const int k0 = (ks + u) * 16 + 8 * h;
      const half8 a0 = *reinterpret_cast<const half8*>(&src[r][k0]);
      const half8 a1 = *reinterpret_cast<const half8*>(&src[32 + r][k0]);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bf[u], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bf[u], acc1, 0, 0, 0);
    }
    ks += g;
  };
  while (ks + 12 <= full) group(std::integral_constant<int, 12>{});
  while (ks + 4 <= full) group(std::integral_constant<int, 4>{});
  while (ks + 1 <= full) group(std::integral_constant<int, 1>{});
  for (; ks < ksteps; ++ks) {  // the remaining (possibly partial) k-steps
    const int k0 = ks * 16 + 8 * h;
    const half8 bf = load_w8(wrow, k0, live ? in : 0);
    const half8 a0 = *reinterpret_cast<const half8*>(&src[r][k0]);
    const half8 a1 = *reinterpret_cast<const half8*>(&src[32 + r][k0]);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bf, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bf, acc1, 0, 0, 0);
  }
}

// The same product on FRAGMENT-MAJOR weights: block nb / k-step ks of a Linear (out, in) is one 1 KB chunk in which lane l = 32 h + r
// holds W[32 nb + r][16 ks + 8 h + 0..7] (zero beyond out / in), i.e. exactly its B fragment -- one fully coalesced 16-byte load per lane.
// Read from the row-major matrix the same fragment touches 64 different cache lines per wave instruction (a row per lane), eight waves
// thrash the 16 KB L1 and every prefetch group costs ~2 k cycles of L2 round trips: the packed copy (kept current by one scatter of the
// flat fp16 working copy per optimiser step) is what makes the weight stream cheap.
// Software pipeline of the fragment-major weight stream.  Left to itself the compiler sinks every weight load next to the MFMA that uses
// it (two loads in flight, `s_waitcnt vmcnt(0..1)` before each pair of MFMAs): a k-step then costs one L2 round trip, ~200 cycles against 64
// of matrix work.  Here the fragments of the NEXT PF_G k-steps are requested before the MFMAs of the current PF_G, with scheduling barriers
// that keep the requests where they are written; the two fragment sets alternate (no register copies).
// T: the TRANSPOSED product (weights as the A operand, activations as B): the same fragments and the same sums, but a lane then holds 16
// output COLUMNS (4 groups of 4 consecutive ones) of its row r instead of 16 rows of its column -- see layer().
// NACC = 2: both 32-row halves of the tile (acc[0], acc[1]); NACC = 1: the half `half` only (acc[0]).
struct alignas(4) BiasQuad { half2v lo, hi; };
struct Bias16 { BiasQuad q[4]; };
// the 16 bias values of a lane's output columns as four groups of four consecutive ones: unconditional loads, issued BEFORE the product so
// that they ride behind its weight stream (a guarded load per element compiles to 16 branches, each waiting for its own load).  `out` a
// multiple of 4 and B 4-byte aligned (every Linear of the flat fp16 working copy): a group is either complete or absent, one request each.
__device__ __forceinline__ Bias16 load_bias16(const _Float16* B, int cb, int out) {
  Bias16 b;
  const half2v z = half2v{(_Float16)0.f, (_Float16)0.f};
  if (!B) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { b.q[j].lo = z; b.q[j].hi = z; }
  } else if (((out & 3) | (int)(reinterpret_cast<uintptr_t>(B) & 3)) == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = cb + 8 * j;
      b.q[j] = *reinterpret_cast<const BiasQuad*>(B + (c < out ? c : 0));   // (columns >= out: zeroed in the epilogue)
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = cb + 8 * j;
      b.q[j].lo = half2v{B[c < out ? c : 0], B[c + 1 < out ? c + 1 : 0]};
      b.q[j].hi = half2v{B[c + 2 < out ? c + 2 : 0], B[c + 3 < out ? c + 3 : 0]};
    }
  }
  return b;
}
constexpr int PF_G = 4;
// RP = ksteps mod 2 PF_G as a compile-time constant (the callers switch on it): pairs of full groups in a branch-free loop, then RP k-steps
// of straight-line code -- no accumulator ever meets a branch (each guarded MFMA is a phi the register allocator answers with copies of the
// 16-register accumulators and out-of-place MFMAs).
// The accumulators start at `bias` (T only: the lane's 16 output columns; null = 0) -- set AFTER the first group's requests are out, so the
// bias loads ride behind them and the epilogue has no bias add.
template <int LD, bool T, int NACC, int RP>
__device__ __forceinline__ void gemm_packed(const _Float16 (*src)[LD], const _Float16* wblk, int ksteps, int r, int h, int half, f32x16* acc,
                                            const Bias16* bias = nullptr) {
  const _Float16* p = wblk + (h * 32 + r) * 8;   // this lane's 16 bytes of chunk 0; chunk k is 512 halfs further
  auto fetch = [&](half8* f, int g) {             // group g: k-steps g PF_G ..; past the end: harmless repeats of the last chunk, never multiplied
#pragma unroll
    for (int u = 0; u < PF_G; ++u) {
      const int k = g * PF_G + u < ksteps ? g * PF_G + u : ksteps - 1;
      f[u] = *reinterpret_cast<const half8*>(p + (size_t)k * 512);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto mma = [&](const half8& w, int ks) {
    const int k0 = ks * 16 + 8 * h;
    if (NACC == 2) {
      const half8 a0 = *reinterpret_cast<const half8*>(&src[r][k0]);
      const half8 a1 = *reinterpret_cast<const half8*>(&src[32 + r][k0]);
      acc[0] = T ? __builtin_amdgcn_mfma_f32_32x32x16_f16(w, a0, acc[0], 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, w, acc[0], 0, 0, 0);
      acc[NACC - 1] = T ? __builtin_amdgcn_mfma_f32_32x32x16_f16(w, a1, acc[NACC - 1], 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, w, acc[NACC - 1], 0, 0, 0);
    } else {
      const half8 a0 = *reinterpret_cast<const half8*>(&src[32 * half + r][k0]);
      acc[0] = T ? __builtin_amdgcn_mfma_f32_32x32x16_f16(w, a0, acc[0], 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, w, acc[0], 0, 0, 0);
    }
  };
  auto stage = [&](const half8* f, int ks, auto CNT) {   // CNT k-steps, straight-line
#pragma unroll
    for (int u = 0; u < decltype(CNT)::value; ++u) mma(f[u], ks + u);
    __builtin_amdgcn_sched_barrier(0);
  };
  using Full = std::integral_constant<int, PF_G>;
  const int npair = ksteps / (2 * PF_G);
  half8 fa[PF_G], fb[PF_G];
  __builtin_amdgcn_sched_barrier(0);
  fetch(fa, 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f32x2 lo = {0.f, 0.f}, hi = {0.f, 0.f};
    if (T && bias) { lo = __builtin_convertvector(bias->q[j].lo, f32x2); hi = __builtin_convertvector(bias->q[j].hi, f32x2); }
#pragma unroll
    for (int c = 0; c < NACC; ++c) { acc[c][4 * j] = lo.x; acc[c][4 * j + 1] = lo.y; acc[c][4 * j + 2] = hi.x; acc[c][4 * j + 3] = hi.y; }
  }
  __builtin_amdgcn_sched_barrier(0);
  for (int i = 0; i < npair; ++i) {   // branch-free body: the wait counters stay exact (the next group's four requests outstanding)
    fetch(fb, 2 * i + 1);
    stage(fa, 2 * i * PF_G, Full{});
    fetch(fa, 2 * i + 2);
    stage(fb, (2 * i + 1) * PF_G, Full{});
  }
  if constexpr (RP > PF_G) fetch(fb, 2 * npair + 1);
  if constexpr (RP > 0) stage(fa, 2 * npair * PF_G, std::integral_constant<int, (RP < PF_G ? RP : PF_G)>{});
  if constexpr (RP > PF_G) stage(fb, (2 * npair + 1) * PF_G, std::integral_constant<int, RP - PF_G>{});
}
// run(RP) with RP = ksteps mod 2 PF_G as a compile-time constant
template <class F>
__device__ __forceinline__ void dispatch_ksteps(int ksteps, F&& run) {
  static_assert(PF_G == 4, "eight remainders");
  switch (ksteps & 7) {
    case 0: run(std::integral_constant<int, 0>{}); break;
    case 1: run(std::integral_constant<int, 1>{}); break;
    case 2: run(std::integral_constant<int, 2>{}); break;
    case 3: run(std::integral_constant<int, 3>{}); break;
    case 4: run(std::integral_constant<int, 4>{}); break;
    case 5: run(std::integral_constant<int, 5>{}); break;
    case 6: run(std::integral_constant<int, 6>{}); break;
    default: run(std::integral_constant<int, 7>{}); break;
  }
}
// Epilogue of the transposed product (fragment-major weights): this lane's 16 results are the output columns cb + 8 j + 0..3 (j = 0..3,
// cb = 32 nb + 4 h) of ONE row.  Bias, the fp16 rounding of the Linear and ELU run two outputs per packed instruction and every group of four
// leaves as one 8-byte LDS store (the direct product's lane holds 16 rows of one column: 16 two-byte stores and scalar math per 16 results --
// that epilogue, not the MFMAs, was what a column block cost: ~2 k cycles against 0.4-1.6 k of matrix work, tools/policy_stamp_probe.py).
// (Columns >= out of the last block get the bias of column 0: finite values in the K padding of the next layer, where the fragment-major
// weights are zero -- as harmless as the elu(0) they used to hold, and two guards per pair cheaper.)
template <bool ELU, int LDD>
__device__ __forceinline__ void store_row_groups(const f32x16& acc, _Float16 (*dst)[LDD], int row, int cb, int npad) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c0 = cb + 8 * j;
    uint32_t o[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int i = 4 * j + 2 * q;
      const half2v hq = __builtin_convertvector(f32x2{acc[i], acc[i + 1]}, half2v);  // the fp16 output of the Linear (the bias is in the accumulator)
      if (ELU) {
        // elu on the fp16 value, computed in fp32 as torch does: h > 0 ? h : exp(h) - 1.  The select works on the sign bits of the packed pair
        // (h = +0 and exp(+0) - 1 are the same number, NaN stays NaN either way): one packed shift and one bit-field insert per pair
        const f32x2 e = f32x2{__builtin_amdgcn_exp2f(__builtin_fmaf((float)hq.x, 1.4426950408889634f, 0.f)),
                              __builtin_amdgcn_exp2f(__builtin_fmaf((float)hq.y, 1.4426950408889634f, 0.f))} - f32x2{1.f, 1.f};   // (v_fma_mix_f32: the conversion rides in the multiply)
        const half2v eh = __builtin_convertvector(e, half2v);
        const uint32_t hb = __builtin_bit_cast(uint32_t, hq), eb = __builtin_bit_cast(uint32_t, eh);
        uint32_t neg;   // 0xffff in every half whose sign bit is set (the compiler turns the C form of this into two compares and selects)
        asm("v_pk_ashrrev_i16 %0, %1, %2" : "=v"(neg) : "v"(0x000f000fu), "v"(hb));   // (an inline 15 would only reach the low half)
        o[q] = (eb & neg) | (hb & ~neg);
      } else {
        o[q] = __builtin_bit_cast(uint32_t, hq);
      }
    }
    if (LDD >= PF_LD || c0 < npad) *reinterpret_cast<uint2*>(&dst[row][c0]) = make_uint2(o[0], o[1]);
  }
}

// dst[:, 0:Npad] = act(src[:, 0:K] W^T + b) for the 64 rows of the workgroup; this wave takes column blocks wave, wave + PF_WAVES, ...
template <bool ELU, bool PK, bool ONEHALF = false, int LDS, int LDD>
__device__ __forceinline__ void layer(const _Float16 (*src)[LDS], _Float16 (*dst)[LDD], const _Float16* W, const _Float16* B, int in, int out, int wave,
                                      int lane, int tag = 0, int nw = PF_WAVES) {   // nw: waves of the workgroup (8, or 16 in the 4096-row forward kernels)
  const int r = lane & 31, h = lane >> 5;
  const int nblk = (out + 31) >> 5, npad = (out + 15) & ~15;  // (a narrow tile only covers the width padded to 16 columns)
  if constexpr (PK && ONEHALF) {
    // a 32-row tile (the forward-only / rollout kernels of <= 8192 rows): one MFMA chain per column block, the blocks dealt over the waves
    const int ksteps = (in + 15) >> 4;
    dispatch_ksteps(ksteps, [&](auto RP) {
      for (int nb = wave; nb < nblk; nb += nw) {
        const int cb = nb * 32 + 4 * h;
        const Bias16 bias = load_bias16(B, cb, out);
        f32x16 acc;
        if (nb == wave) PF_WSTAMP(tag, wave, 0);
        gemm_packed<LDS, true, 1, decltype(RP)::value>(src, W + (size_t)nb * ksteps * 512, ksteps, r, h, 0, &acc, &bias);
        if (nb == wave) PF_WSTAMP(tag, wave, 1);
        store_row_groups<ELU>(acc, dst, r, cb, npad);
        if (nb == wave) PF_WSTAMP(tag, wave, 2);
      }
    });
    return;
  }
  if constexpr (PK) {
    const int ksteps = (in + 15) >> 4;
    // One 32-row half of a column block per wave where a layer has <= 4 column blocks (the 100-wide one: 4): half or more of the eight waves
    // sat idle while the others ran two MFMA chains each; with (column block, row half) as the unit of work every wave gets one chain.
    // (<= 4 blocks whatever the wave count: each unit streams its block's weights, and two half units stream them twice -- the 200-wide layer
    // as 14 half units on 16 waves pulled 350 KB through the CU's 64 B/clk L1 port, 5.5 k cycles, where 7 full units take 2.7 k)
    if (nblk * 2 <= PF_WAVES) {
      if (wave < 2 * nblk) {
        const int nb = wave < nblk ? wave : wave - nblk, half = wave < nblk ? 0 : 1, cb = nb * 32 + 4 * h;   // (no integer division)
        dispatch_ksteps(ksteps, [&](auto RP) {
          const Bias16 bias = load_bias16(B, cb, out);
          f32x16 acc;
          PF_WSTAMP(tag, wave, 0);
          gemm_packed<LDS, true, 1, decltype(RP)::value>(src, W + (size_t)nb * ksteps * 512, ksteps, r, h, half, &acc, &bias);
          PF_WSTAMP(tag, wave, 1);
          store_row_groups<ELU>(acc, dst, 32 * half + r, cb, npad);
          PF_WSTAMP(tag, wave, 2);
        });
      }
      return;
    }
    dispatch_ksteps(ksteps, [&](auto RP) {
      for (int nb = wave; nb < nblk; nb += nw) {
        const int cb = nb * 32 + 4 * h;
        const Bias16 bias = load_bias16(B, cb, out);
        f32x16 acc[2];
        if (nb == wave) PF_WSTAMP(tag, wave, 0);
        gemm_packed<LDS, true, 2, decltype(RP)::value>(src, W + (size_t)nb * ksteps * 512, ksteps, r, h, 0, acc, &bias);
        if (nb == wave) PF_WSTAMP(tag, wave, 1);
        store_row_groups<ELU>(acc[0], dst, r, cb, npad);
        store_row_groups<ELU>(acc[1], dst, 32 + r, cb, npad);
        if (nb == wave) PF_WSTAMP(tag, wave, 2); else PF_WSTAMP(tag, wave, 3);
      }
    });
    return;
  }
  for (int nb = wave; nb < nblk; nb += nw) {
    const int n = nb * 32 + r;
    const _Float16* wrow = W + (size_t)(n < out ? n : 0) * in;
    const float bias = (B && n < out) ? (float)B[n] : 0.f;
    f32x16 acc0, acc1;   // (the accumulators start at the bias, as in the fragment-major path: the two stay bit-identical)
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = bias; acc1[i] = bias; }
    gemm_col_block(src, wrow, n < out, in, r, h, acc0, acc1);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
      float v0 = (float)(_Float16)acc0[i], v1 = (float)(_Float16)acc1[i];  // the fp16 output of the Linear
      if (ELU) { v0 = v0 > 0.f ? v0 : __expf(v0) - 1.f; v1 = v1 > 0.f ? v1 : __expf(v1) - 1.f; }  // (v_exp_f32: the result is rounded to fp16 anyway)
      if (LDD >= PF_LD || n < npad) {
        dst[row][n] = (_Float16)v0;
        dst[32 + row][n] = (_Float16)v1;
      }
    }
  }
}

// the two heads on the last hidden activations `src`: one column block (num_actions + 1 <= 32 columns), wave 0; results as fp32 of the
// fp16 outputs, to global memory or (rollout mode) into `tile`, the free activation tile viewed as 33-float rows
template <bool ROLL, bool PK, int LD>
__device__ __forceinline__ void heads(const PolicyArgs& a, const _Float16 (*src)[LD], int in, float* tile, int64_t row0, int nrow, int lane) {
  const int r = lane & 31, h = lane >> 5, A = a.num_actions;
  const _Float16* wrow = r < A ? a.w_mu + (size_t)r * in : a.w_val;
  const bool live = r <= A;
  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  static_assert(!PK, "the fragment-major instantiations use heads_half");
  gemm_col_block(src, wrow, live, in, r, h, acc0, acc1);
  const float bias = r < A ? (float)a.b_mu[r] : (r == A ? (float)a.b_val[0] : 0.f);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
    const float v0 = (float)(_Float16)(acc0[i] + bias), v1 = (float)(_Float16)(acc1[i] + bias);
    if (ROLL) {
      if (r <= A) { tile[row * 33 + r] = v0; tile[(32 + row) * 33 + r] = v1; }
    } else {
      if (row < nrow) { if (r < A) a.mu[(row0 + row) * A + r] = v0; else if (r == A) a.value[row0 + row] = v0; }
      if (32 + row < nrow) { if (r < A) a.mu[(row0 + 32 + row) * A + r] = v1; else if (r == A) a.value[row0 + 32 + row] = v1; }
    }
  }
}

__device__ __forceinline__ float head_bias(const PolicyArgs& a, int r) {
  return r < a.num_actions ? (float)a.b_mu[r] : (r == a.num_actions ? (float)a.b_val[0] : 0.f);
}
template <bool ROLL, int LD>
__device__ __forceinline__ void heads_half(const PolicyArgs& a, const _Float16 (*src)[LD], int in, float* tile, int64_t row0, int nrow, int lane, int half) {
  const int r = lane & 31, h = lane >> 5, A = a.num_actions, ksteps = (in + 15) >> 4;
  const float bias = head_bias(a, r);   // requested ahead of the weight stream (behind the product it was a round trip of its own)
  dispatch_ksteps(ksteps, [&](auto RP) {
    f32x16 acc;
    gemm_packed<LD, false, 1, decltype(RP)::value>(src, a.w_mu, ksteps, r, h, half, &acc);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = 32 * half + (i & 3) + 8 * (i >> 2) + 4 * h;
      const float v = (float)(_Float16)(acc[i] + bias);
      if (ROLL) {
        if (r <= A) tile[row * 33 + r] = v;
      } else if (row < nrow) {
        if (r < A) a.mu[(row0 + row) * A + r] = v; else if (r == A) a.value[row0 + row] = v;
      }
    }
  });
}

// MODE 0: forward only; 1: rollout step (ROLL); 2: training forward (activations kept for the backward pass).  LD0 / LD1: row strides
// (halfs) of the two LDS activation tiles -- tile 0 holds the staged input and the outputs of layers 1, 3, 5, tile 1 those of layers
// 0, 2, 4.  (424, 424) fits every supported width; (216, 424) is 80 KB, so that two workgroups share a CU (training forward of
// 54-400-200-100: 512 workgroups, each one's MFMAs cover the other's weight-fetch latency).
// Waves per workgroup.  Training forward (MODE 2): 8, and two workgroups per CU (4 waves per SIMD, <= 128 VGPRs) where the
// two tiles fit twice into the CU's LDS.  Forward-only / rollout (4096 rows = 64 workgroups on 256 CUs, one per CU): 16 waves -- every
// layer of 54-400-200-100 is then ONE round of units (13 column blocks; 14, 8 and 2 (block, row half) units) instead of two rounds of
// double work on 8 waves; the kernel is a chain of per-layer latencies, and this halves the two longest links.
constexpr int PF_FWD_WAVES = 16;
template <int MODE, int LD0, int LD1, bool PK, int ROWS = PF_ROWS>
__global__ __launch_bounds__(MODE == 2 ? PF_WAVES * 64 : PF_FWD_WAVES * 64, (MODE != 2 || (LD0 + LD1) * PF_ROWS * 2 <= 80 * 1024) ? 4 : 2) void policy_forward_kernel(PolicyArgs a) {
  constexpr bool ROLL = MODE == 1, TRAIN = MODE == 2;
  // (32-row tiles for the TRAINING forward, 1024 workgroups at three per CU: 35.5-36.3 us against 31.3 -- every workgroup streams the whole weight
  // set, and the stream doubles)
  static_assert(ROWS == PF_ROWS || (ROWS == 32 && PK && MODE != 2), "32-row tiles: fragment-major weights, forward-only / rollout kernels");
  constexpr bool ONEHALF = ROWS == 32;
  __shared__ __attribute__((aligned(16))) _Float16 t0[ROWS][LD0];
  __shared__ __attribute__((aligned(16))) _Float16 t1[ROWS][LD1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int nw = MODE == 2 ? PF_WAVES : PF_FWD_WAVES, nt = nw * 64;   // (= blockDim.x: the launches below)
  if (ROLL && a.dr_on && blockIdx.x == gridDim.x - 1) {
    // the extra workgroup: the coming env step's domain randomisation (bez_dr_step.h), beside the forward pass of the others.  It reads the
    // envs' reset flags / episode counters and writes the randomisation's own state and the per-env parameter rows -- nothing the other
    // workgroups touch (their action noise comes from the snapshot the last step kernel left, not from the state updated here).
    int* scratch = reinterpret_cast<int*>(&t0[0][0]);
    constexpr int cap = (int)(sizeof(t0) / sizeof(int)) - 1;
    bez::dr::dr_step(a.dr, scratch + 1, cap, scratch);
    return;
  }
  const int64_t row0 = (int64_t)blockIdx.x * ROWS;
  const int nrow = (int)((a.n - row0) < (int64_t)ROWS ? (a.n - row0) : (int64_t)ROWS);
  PF_STAMP(0);
  // ROLL with the env's action noise (BezPpoActionNoise): the samples do not depend on the forward pass, so they are drawn HERE, in the
  // shadow of the observation loads: one Philox block + Box-Muller pair = four samples per thread (the block's 64 x A / 4 quads on its first
  // threads), parked in LDS until the epilogue (this mode's tiles leave 54 KB free).  Every element used to recompute its whole quad:
  // 4 x the generator work, +1.3 us on the launch.  (row0 * A is a multiple of 4: a workgroup's block starts on a quad.)
  __shared__ float nzs[ROLL ? ROWS * 32 : 1];
  if (ROLL && a.an.snap_dev) {
    const bez::DrSnap sn = *static_cast<const bez::DrSnap*>(a.an.snap_dev);
    const unsigned long long frame = (unsigned long long)sn.frame_hi << 32 | sn.frame_lo;
    const int nel = nrow * a.num_actions;
    for (int q = tid; 4 * q < nel; q += nt) {
      float z4[4];
      bez::dr_noise_quad(a.an.seed, a.an.env_id_offset, frame, 1, ((row0 * a.num_actions) >> 2) + q, z4);
#pragma unroll
      for (int c = 0; c < 4; ++c) nzs[4 * q + c] = fmaf(z4[c], sn.sd, sn.mean);
    }
  }
  // stage the (normalised) observations as fp16, zero-padded to a multiple of 16 columns: lanes over columns (the column's mean and
  // standard deviation once per lane), waves over rows -- no index division, no fp64 in the row loop
  const int kpad0 = (a.d_in + 15) & ~15;
  for (int k = lane; k < kpad0; k += 64) {
    const bool kin = k < a.d_in;
    float mk = 0.f, sk = 1.f;
    if (a.mean && kin) { mk = (float)a.mean[k]; sk = sqrtf((float)a.var[k] + a.eps); }
#pragma unroll 4
    for (int rr = wave; rr < ROWS; rr += nw) {
      float v = 0.f;
      if (rr < nrow && kin) {
        v = a.obs[(row0 + rr) * a.d_in + k];
        if (ROLL) a.mb_obs[(row0 + rr) * a.ld_obs + k] = v;  // the rollout buffer keeps the raw observation
        if (a.mean) {
          v = (v - mk) / sk;
          v = fminf(fmaxf(v, -5.0f), 5.0f);
        }
      }
      t0[rr][k] = (_Float16)v;
    }
  }
  __syncthreads();
  PF_STAMP(1);
  __shared__ float ls_s[1];   // ROLL: the sum of log sigma (one number for every row)
  [[maybe_unused]] double ep_c = 0.0, ep_r = 0.0, ep_l = 0.0;   // the bookkeeping wave: this lane's share of the finished episodes' count / return / length
  // What does not depend on the forward pass runs on the two last waves, which the layers of 54-400-200-100 leave idle (13, 7 and 4 column blocks on
  // 16 waves), in their shadow (it used to follow the heads: three dependent rounds of load latency at the end of every workgroup): the sum of log
  // sigma in the reference's order on wave 14; on wave 15 the bookkeeping of the env step BEFORE this one, in two parts -- loads, shaping, stores behind
  // the staging barrier; the three fp64 butterflies and the episode sums behind the first layer's barrier (all of it in the first window held that
  // barrier up by 1.3 k cycles).
  auto log_sigma_sum = [&]() {
    const int A = a.num_actions;
    const float lj = lane < A ? a.logstd[lane] : 0.f;
    float ls = 0.f;
    for (int j = 0; j < A; ++j) ls += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lj), j));
    if (lane == 0) ls_s[0] = ls;
  };
  auto bookkeeping = [&]() {
    // as ppo_rollout_post_kernel (csrc/bez_ppo.hip): shaped reward with the time-out bootstrap on the value THAT step's policy launch stored,
    // done flags as floats (also this step's rollout row), episode return / length.  The env's reward / reset / time-out buffers still hold that
    // step's results: the next env step runs behind this launch.
    const BezPpoRolloutPost& q = a.post;
    const bool ok = lane < nrow;
    const int64_t i = row0 + (ok ? lane : 0);
    double c = 0.0, r = 0.0, l = 0.0;
    if (ok) {
      const float rw = q.rew[i];
      float sh = rw * q.reward_scale;
      if (q.bootstrap) sh += q.gamma * q.prev_values[i] * (float)q.timeouts[i];
      q.shaped[i] = sh;
      const float d = (float)q.reset[i];
      q.dones_f[i] = d;
      a.mb_dones[i] = d;
      const float cr = q.cur_rew[i] + rw, cl = q.cur_len[i] + 1.0f;
      c = d; r = cr * d; l = cl * d;
      q.cur_rew[i] = cr * (1.0f - d); q.cur_len[i] = cl * (1.0f - d);
    }
    ep_c = c; ep_r = r; ep_l = l;
  };
  auto episode_sums = [&]() {
    double c = ep_c, r = ep_r, l = ep_l;
#pragma unroll
    for (int o = (ROWS < 64 ? ROWS : 64) / 2; o > 0; o >>= 1) { c += __shfl_xor(c, o, 64); r += __shfl_xor(r, o, 64); l += __shfl_xor(l, o, 64); }   // (lanes >= ROWS hold zeros)
    if (lane == 0 && c != 0.0) {
      if (a.post.ep_parts) {   // this workgroup's own slot: a plain read-modify-write (launches are stream-ordered, nobody else touches it)
        double* p = a.post.ep_parts + 4 * (size_t)blockIdx.x;
        p[0] += c; p[1] += r; p[2] += l;
      } else {
        atomicAdd(&a.post.ep_stats[0], c); atomicAdd(&a.post.ep_stats[1], r); atomicAdd(&a.post.ep_stats[2], l);
      }
    }
  };
  // window w = behind the barrier of layer w - 1 (0: behind the staging barrier); with one hidden layer both parts run in window 1 at the latest
  auto shadow_work = [&](int w) {
    if (!ROLL) return;
    if (w == 0 && wave == nw - 2) log_sigma_sum();
    if (a.post.rew && wave == nw - 1) {
      if (w == 0) bookkeeping();
      if (w == 1) episode_sums();
    }
  };
  shadow_work(0);
  if (TRAIN) store_tile(t0, a.x0_out, row0, nrow, a.d_in, tid, nw);
  int in = a.d_in;
  for (int L = 0; L < a.nhid; L += 2) {
    layer<true, PK, ONEHALF>(t0, t1, a.w[L], a.b[L], in, a.width[L], wave, lane, L, nw);
    PF_STAMP(2 + 2 * L);
    // the next layer reads K padded to 16: columns width..pad16(width) were written as elu(0 + 0) = 0 by the padded column block
    __syncthreads();
    PF_STAMP(3 + 2 * L);
    shadow_work(L + 1);
    in = a.width[L];
    if (TRAIN) store_tile(t1, a.act_out[L], row0, nrow, in, tid, nw);  // (moving these stores behind the next layer's product changed nothing: 31.5 -> 31.3 us)
    if (L + 1 < a.nhid) {
      layer<true, PK, ONEHALF>(t1, t0, a.w[L + 1], a.b[L + 1], in, a.width[L + 1], wave, lane, L + 1, nw);
      PF_STAMP(4 + 2 * L);
      __syncthreads();
      PF_STAMP(5 + 2 * L);
      in = a.width[L + 1];
      if (TRAIN) store_tile(t0, a.act_out[L + 1], row0, nrow, in, tid, nw);
    }
  }
  const bool in_t1 = (a.nhid & 1) != 0;  // where the last hidden activations are; the other tile is free
  float* tile = in_t1 ? reinterpret_cast<float*>(&t0[0][0]) : reinterpret_cast<float*>(&t1[0][0]);
  // ROLL: the sampling pass's operands are requested here, in front of the heads (one (env, action) element per thread and pass)
  constexpr int NIT = ROLL ? (ROWS * 31 + nt - 1) / nt : 1;
  [[maybe_unused]] float ez[NIT], el[NIT];
  if (ROLL) {
    const int A = a.num_actions;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + it * nt;
      ez[it] = el[it] = 0.f;
      if (idx < nrow * A) { ez[it] = a.noise[row0 * A + idx]; el[it] = a.logstd[idx % A]; }
    }
  }
  if constexpr (PK) {
    if (wave < (ONEHALF ? 1 : 2)) {   // two waves, one 32-row half each (a 32-row tile: one)
      if (in_t1) heads_half<ROLL>(a, t1, in, tile, row0, nrow, lane, wave);
      else heads_half<ROLL>(a, t0, in, tile, row0, nrow, lane, wave);
    }
  } else if (wave == 0) {
    if (in_t1) heads<ROLL, PK>(a, t1, in, tile, row0, nrow, lane);
    else heads<ROLL, PK>(a, t0, in, tile, row0, nrow, lane);
  }
  PF_STAMP(14);
  if (ROLL) {
    // sampling, neglogp, clamp, rollout-buffer rows: one thread per (env, action), the per-env sum through the same LDS tile
    __syncthreads();
    const int A = a.num_actions;
    float* zz = tile + ROWS * 33;  // (rows, 36) squared standardised actions: 16-byte rows a bank group apart (stride 32 was a 32-way bank conflict per read)
    static_assert(sizeof(float) * ROWS * (33 + 36) <= sizeof(_Float16) * ROWS * (LD0 < LD1 ? LD0 : LD1), "the free tile holds both");
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + it * nt;
      if (idx >= nrow * A) break;
      const int rr = idx / A, j = idx - rr * A;
      const float m = tile[rr * 33 + j];
      const float l = el[it], sg = expf(l), z = ez[it];
      const float x = fmaf(sg, z, m);
      const int64_t o = (row0 + rr) * A + j;
      float xe = fminf(fmaxf(x, -1.0f), 1.0f);   // what rl_games hands to env.step
      if (a.an.snap_dev) xe = xe + nzs[idx];
      const int64_t om = (row0 + rr) * a.ld_act + j;   // (the rollout rows may be strided: written straight into the env-major dataset)
      a.mb_mu[om] = m; a.act[om] = x; a.act_env[o] = xe; a.sigma[om] = sg;
      const float q = (x - m) / sg;  // as the reference computes it from the stored action
      zz[rr * 36 + j] = q * q;
    }
    __syncthreads();
    PF_STAMP(13);
    if (tid < nrow) {
      float acc = 0.f;   // (the additions in the order j = 0, 1, ...: what the separate rollout kernel computes)
      const float4* z4 = reinterpret_cast<const float4*>(zz + tid * 36);
      for (int j = 0; j < A; j += 4) {
        const float4 v = z4[j >> 2];
        acc += v.x;
        if (j + 1 < A) acc += v.y;
        if (j + 2 < A) acc += v.z;
        if (j + 3 < A) acc += v.w;
      }
      a.neglogp[(row0 + tid) * a.ld_one] = 0.5f * acc + 0.5f * 1.8378770664093453f * (float)A + ls_s[0];
      float v = tile[tid * 33 + A];
      if (a.vmean) v = sqrtf((float)a.vvar[0] + a.veps) * fminf(fmaxf(v, -5.0f), 5.0f) + (float)a.vmean[0];
      a.mb_val[row0 + tid] = v;
      if (!a.post.rew) a.mb_dones[row0 + tid] = a.dones[row0 + tid];
    }
    PF_STAMP(12);
  }
  PF_STAMP(15);
}

// ---- the input-gradient half of the backward pass of one minibatch as ONE kernel (bez_ppo_policy_backward): per 64-row tile, from
// d loss / d mu and d loss / d value down to d loss / d (pre-activation) of every hidden layer.  What it replaces per layer: the ELU
// derivative + bias-gradient pass over (rows, width) and the input-gradient GEMM of torch's Linear backward -- the tile's gradient
// never leaves LDS between them.  The weight gradients stay split-K GEMMs on the gz tensors written here.
// B operands are TRANSPOSED weights (8 consecutive k per lane must be contiguous): wt[L] = W_L^T (width[L-1], width[L]) and the two
// heads as one (width[last], 32) matrix; a scatter of the fp16 working copy refreshes them once per optimiser step.
struct BackwardArgs {
  int64_t n; int nhid; int width[PF_MAXL]; int num_actions;
  const float* gmu; const float* gval;
  const _Float16* act[PF_MAXL];
  const _Float16* wt[PF_MAXL];
  const _Float16* wht;
  _Float16* gz[PF_MAXL];
  _Float16* gmu16; _Float16* gv16;
  float* bgrad[PF_MAXL]; float* bmu_grad; float* bv_grad;
  int packed;  // wt[L] / wht are fragment-major (packed as Linear(out = width[L-1], in = width[L]) and Linear(out = width[last], in = 32))
  float* partial; int ptotal; int poff[PF_MAXL];  // per-workgroup column sums of gz (ceil(n / 64) rows of prow floats; ptotal = sum of the widths, layer L at poff[L])
  int prow;                                       // = ptotal + 32: the heads' column sums [d/d mu | d/d value] follow the hidden layers' in a row
  bez_loss::LossArgs loss;                        // the fused-loss instantiations (LA > 0): the minibatch's loss operands; gmu / gval are then not read
};

// gz = g * elu'(y) of a full 64-row tile whose width is a multiple of 4, in place in tile A and out to HBM, and its per-workgroup column
// sums.  Waves over rows, lanes over groups of four columns (8-byte accesses; a narrow layer puts 64 / P rows side by side in a wave,
// P = the power of two >= W / 4): every address is a row base plus a lane offset, all of a thread's loads are in flight before its first
// store, and the arithmetic runs on mixed-precision FMAs -- elu'(y) = min(y, 0) + 1 exactly as fp32 of the fp16 ELU output, the product
// with g in fp32, one rounding to fp16 (what torch's elu_backward does under autocast), the column sums over the rounded values.
template <int LA, int LB>
__device__ __forceinline__ void backward_elementwise_x4(const BackwardArgs& a, _Float16 (*A)[LA], _Float16 (*B)[LB], int L, int64_t row0, int tid, int wave,
                                                        int lane) {
  const int W = a.width[L], g = W >> 2;
  int P = 64, sh = 6;
  while (P > 1 && (P >> 1) >= g) { P >>= 1; --sh; }
  const int rp = 64 >> sh, lp = lane & (P - 1), sub = lane >> sh;   // rp rows side by side in a wave
  const int npass = (g + 63) >> 6;                                  // column-group passes (2 only where g > 64: then P = 64, rp = 1)
  float* red = reinterpret_cast<float*>(&B[0][0]);                  // [8 rp][W] partial column sums (tile B is free until the GEMM writes it)
  static_assert(sizeof(_Float16) * PF_ROWS * LB >= sizeof(float) * 8 * PF_MAXW, "the free tile holds the column-sum partials");
  const _Float16* act = a.act[L] + row0 * W;
  _Float16* gz = a.gz[L] + row0 * W;
  const int r0 = wave * rp + sub, rstep = PF_WAVES * rp, niter = PF_ROWS / rstep;   // (uniform: 8, 4, 2 or 1 rows per thread)
  for (int j = 0; j < npass; ++j) {
    const int cg = lp + 64 * j;
    f32x2 s01 = {0.f, 0.f}, s23 = {0.f, 0.f};
    if (cg < g) {
      constexpr int NB = 8;   // rows per batch (64 / rstep <= 8 iterations in all)
      uint2 yy[NB], gg[NB];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int rr = r0 + i * rstep;
        if (i < niter) {
          yy[i] = *reinterpret_cast<const uint2*>(act + (size_t)rr * W + 4 * cg);
          gg[i] = *reinterpret_cast<const uint2*>(&A[rr][4 * cg]);
        }
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int rr = r0 + i * rstep;
        if (i < niter) {
          uint32_t zz[2];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const half2v y = __builtin_bit_cast(half2v, q ? yy[i].y : yy[i].x), gq = __builtin_bit_cast(half2v, q ? gg[i].y : gg[i].x);
            const half2v yn = __builtin_elementwise_min(y, half2v{(_Float16)0.f, (_Float16)0.f});
            // g (min(y, 0) + 1) = g min(y, 0) + g: one fused multiply-add on fp16 operands, exact before its single fp32 rounding -- the same
            // number as the fp32 product of g with the (exact) fp32 sum
            const half2v z = __builtin_convertvector(f32x2{__builtin_fmaf((float)gq.x, (float)yn.x, (float)gq.x), __builtin_fmaf((float)gq.y, (float)yn.y, (float)gq.y)}, half2v);
            zz[q] = __builtin_bit_cast(uint32_t, z);
            f32x2& sacc = q ? s23 : s01;   // the bias gradient sums what the GEMMs see
            sacc = f32x2{__builtin_fmaf((float)z.x, 1.f, sacc.x), __builtin_fmaf((float)z.y, 1.f, sacc.y)};
          }
          *reinterpret_cast<uint2*>(&A[rr][4 * cg]) = make_uint2(zz[0], zz[1]);
          *reinterpret_cast<uint2*>(gz + (size_t)rr * W + 4 * cg) = make_uint2(zz[0], zz[1]);
        }
      }
      *reinterpret_cast<float4*>(red + (size_t)r0 * W + 4 * cg) = make_float4(s01.x, s01.y, s23.x, s23.y);
    }
  }
  __syncthreads();
  if (tid < W) {
    float sum = 0.f;
    const int parts = PF_WAVES * rp;
    for (int q = 0; q < parts; ++q) sum += red[q * W + tid];
    // (512 workgroups adding to the same 700 addresses cost 20 us of the kernel: per-workgroup partials + one small reduction instead)
    a.partial[(size_t)blockIdx.x * a.prow + a.poff[L] + tid] = sum;
  }
}

// one layer of the chain: gz = g * elu'(y) in place in tile A (and out to HBM), its per-workgroup column sums, then d/d h_{L-1} = gz W_L
// into tile B.  The reduction scratch lives in tile B, which is free until the GEMM writes it.
template <bool PK, int LA, int LB>
__device__ __forceinline__ void backward_stage(const BackwardArgs& a, _Float16 (*A)[LA], _Float16 (*B)[LB], int L, int64_t row0, int nrow, int tid,
                                               int wave, int lane) {
  constexpr int NT = PF_WAVES * 64;
  const int W = a.width[L];
  if ((W & 3) == 0 && nrow == PF_ROWS) {
    backward_elementwise_x4(a, A, B, L, row0, tid, wave, lane);
  } else {   // a ragged last tile, or a width that is only even: column pairs, guarded rows
    float2* red = reinterpret_cast<float2*>(&B[0][0]);
    static_assert(sizeof(_Float16) * PF_ROWS * LB >= sizeof(float2) * NT, "the free tile holds the reduction scratch");
    const int ncp = W >> 1, nparts = NT / ncp;
    const int part = tid / ncp, cp = tid - part * ncp;
    float2 acc = make_float2(0.f, 0.f);
    if (part < nparts) {  // consecutive threads on consecutive column pairs of a row
      for (int rb = part; rb < nrow; rb += 8 * nparts) {
        uint32_t yy[8], gy[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int rr = rb + u * nparts;
          yy[u] = 0u; gy[u] = 0u;
          if (rr < nrow) {
            yy[u] = *reinterpret_cast<const uint32_t*>(a.act[L] + (row0 + rr) * W + 2 * cp);
            gy[u] = *reinterpret_cast<const uint32_t*>(&A[rr][2 * cp]);
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int rr = rb + u * nparts;
          if (rr < nrow) {
            const _Float16 g0 = reinterpret_cast<const _Float16*>(&gy[u])[0], g1 = reinterpret_cast<const _Float16*>(&gy[u])[1];
            const float y0 = (float)reinterpret_cast<const _Float16*>(&yy[u])[0], y1 = (float)reinterpret_cast<const _Float16*>(&yy[u])[1];
            _Float16 z[2];
            z[0] = (_Float16)((float)g0 * (y0 > 0.f ? 1.f : y0 + 1.f));
            z[1] = (_Float16)((float)g1 * (y1 > 0.f ? 1.f : y1 + 1.f));
            const uint32_t zz = *reinterpret_cast<const uint32_t*>(z);
            *reinterpret_cast<uint32_t*>(&A[rr][2 * cp]) = zz;
            *reinterpret_cast<uint32_t*>(a.gz[L] + (row0 + rr) * W + 2 * cp) = zz;
            acc.x += (float)z[0]; acc.y += (float)z[1];  // the bias gradient sums what the GEMMs see
          }
        }
      }
    }
    red[tid] = acc;
    __syncthreads();
    if (tid < ncp) {
      float sx = 0.f, sy = 0.f;
      for (int q = 0; q < nparts; ++q) { sx += red[q * ncp + tid].x; sy += red[q * ncp + tid].y; }
      float* dst = a.partial + (size_t)blockIdx.x * a.prow + a.poff[L] + 2 * tid;
      dst[0] = sx; dst[1] = sy;
    }
  }
  if (L > 0) {
    __syncthreads();  // the scratch has been read: tile B may be overwritten
    layer<false, PK>(A, B, a.wt[L], nullptr, W, a.width[L - 1], wave, lane);  // d/d h_{L-1} = gz W_L
    __syncthreads();
  }
}

// LD0 / LD1: row strides of the two LDS tiles.  Tile 0 holds the (64, 32) head-gradient tile and d/d h_L for L = nhid-2, nhid-4, ...;
// tile 1 holds d/d h_L for L = nhid-1, nhid-3, ...  (216, 424) = 80 KB: two workgroups per CU for 54-400-200-100.
// LA > 0: the PPO loss of the tile (bez_ppo_loss.h, action width LA as its compile-time constant) runs IN FRONT of the head stage, in tile 1:
// d loss / d mu and d loss / d value go from LDS straight into the fp16 head-gradient tile -- one launch (11 us of latencies) and their
// round trip through HBM less per minibatch step.  Same arithmetic, same bits as bez_ppo_loss + bez_ppo_policy_backward.
template <int LD0, int LD1, bool PK, int LA = 0>
// 4 waves per SIMD = two workgroups per CU (<= 128 VGPRs) only where the two tiles fit twice into the CU's LDS: the (424, 424)
// fallback for wider networks holds 106 KB and runs one workgroup per CU, so it does not ask for an occupancy it cannot have
__global__ __launch_bounds__(PF_WAVES * 64, ((LD0 + LD1) * PF_ROWS * 2 <= 80 * 1024) ? 4 : 2) void policy_backward_kernel(BackwardArgs a) {
  __shared__ __attribute__((aligned(16))) _Float16 t0[PF_ROWS][LD0];
  __shared__ __attribute__((aligned(16))) _Float16 t1[PF_ROWS][LD1];
  constexpr int NT = PF_WAVES * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row0 = (int64_t)blockIdx.x * PF_ROWS;
  const int nrow = (int)((a.n - row0) < (int64_t)PF_ROWS ? (a.n - row0) : (int64_t)PF_ROWS);
  const int A = a.num_actions;
  constexpr int LOSS_FLOATS = LA > 0 ? bez_loss::loss_lds_floats(LA > 0 ? LA : 1) : 0;
  static_assert(sizeof(float) * (LOSS_FLOATS + NT) <= sizeof(_Float16) * PF_ROWS * LD1, "tile 1 holds the loss tile and the head stage's scratch");
  static_assert(bez_loss::LOSS_TB == PF_ROWS, "the loss tile is the backward tile");
  float* ltile = reinterpret_cast<float*>(&t1[0][0]);   // (tile 1 is free until the first GEMM)
  if constexpr (LA > 0) bez_loss::ppo_loss_tile<LA, true>(a.loss, ltile, tid);   // ends behind a barrier: ltile[row * (LA + 1) + k], gvalue[row]
  {  // heads: the (64, 32) tile [d/d mu | d/d value | 0] as fp16 (= the cast nodes of autocast), its copies for the head weight-gradient
     // GEMMs, and its column sums = the head bias gradients.  Thread -> column tid & 31, rows tid >> 5, + 16, ...
    float* red = ltile + LOSS_FLOATS;
    const float* lgv = ltile + 4 * PF_ROWS * (LA + 1) + 3 * bez_loss::LOSS_CW * PF_ROWS;
    const int k = tid & 31, rsub = tid >> 5;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < PF_ROWS / (NT / 32); ++j) {
      const int rr = rsub + j * (NT / 32);
      float v = 0.f;
      if (rr < nrow) {
        if constexpr (LA > 0) v = k < A ? ltile[rr * (LA + 1) + k] : (k == A ? lgv[rr] : 0.f);
        else v = k < A ? a.gmu[(row0 + rr) * A + k] : (k == A ? a.gval[row0 + rr] : 0.f);
      }
      const _Float16 hv = (_Float16)v;
      t0[rr][k] = hv;
      if (rr < nrow) {
        if (k < A) a.gmu16[(row0 + rr) * A + k] = hv; else if (k == A) a.gv16[row0 + rr] = hv;
        acc += (float)hv;
      }
    }
    red[tid] = acc;
    __syncthreads();
    if (tid <= A) {   // per-workgroup partial, added in fixed order by policy_bias_reduce_kernel (no float atomics: bit-reproducible)
      float sum = 0.f;
      for (int q = 0; q < NT / 32; ++q) sum += red[q * 32 + tid];
      a.partial[(size_t)blockIdx.x * a.prow + a.ptotal + tid] = sum;
    }
    __syncthreads();
  }
  layer<false, PK>(t0, t1, a.wht, nullptr, 32, a.width[a.nhid - 1], wave, lane);  // d/d h_last = [gmu | gval] [Wmu; Wv]
  __syncthreads();
  bool in1 = true;
  for (int L = a.nhid - 1; L >= 0; --L) {
    if (in1) backward_stage<PK>(a, t1, t0, L, row0, nrow, tid, wave, lane);
    else backward_stage<PK>(a, t0, t1, L, row0, nrow, tid, wave, lane);
    in1 = !in1;
  }
}

// bias gradients of the hidden layers and of the two heads: column c += sum over the workgroups' partials.  1024 threads = 16 row lanes
// x 64 columns; a row lane sums every 16th partial (eight loads in flight), the lanes meet in LDS in a fixed order (deterministic)
__global__ __launch_bounds__(1024) void policy_bias_reduce_kernel(BackwardArgs a, int nwg) {
  __shared__ float sh[16][64];
  const int l = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + l;
  const int ncol = a.ptotal + a.num_actions + 1;
  float s = 0.f;
  if (c < ncol) {
#pragma unroll 8
    for (int w = rl; w < nwg; w += 16) s += a.partial[(size_t)w * a.prow + c];
  }
  sh[rl][l] = s;
  __syncthreads();
  if (rl == 0 && c < ncol) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += sh[q][l];
    if (c >= a.ptotal) {
      const int k = c - a.ptotal;
      if (k < a.num_actions) a.bmu_grad[k] += t; else a.bv_grad[0] += t;
    } else {
      int L = 0;
      while (L + 1 < a.nhid && c >= a.poff[L + 1]) ++L;
      a.bgrad[L][c - a.poff[L]] += t;
    }
  }
}

// two destinations per source element (forward and backward copies of the same weight): dst[map_a[i]] = dst[map_b[i]] = src[i]
__global__ void scatter2_f16_kernel(const _Float16* __restrict__ src, const int32_t* __restrict__ ma, const int32_t* __restrict__ mb, int64_t n,
                                    _Float16* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const _Float16 v = src[i]; const int32_t a = ma[i], b = mb[i]; if (a >= 0) dst[a] = v; if (b >= 0) dst[b] = v; }
}

// dst[map[i]] = src[i] for the entries with map[i] >= 0: refreshes the transposed weight copies from the flat fp16 working copy
__global__ void scatter_f16_kernel(const _Float16* __restrict__ src, const int32_t* __restrict__ map, int64_t n, _Float16* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const int32_t m = map[i]; if (m >= 0) dst[m] = src[i]; }
}

}  // namespace

#ifdef BEZ_PF_STAMPS
static unsigned long long* g_pf_stamps = nullptr;  // diagnostic build: device buffer of 16 stamps (tools/policy_stamp_probe.py)
extern "C" void bez_ppo_policy_debug_stamps(unsigned long long* dev) {   // 16 phase stamps, then (layer, wave, phase) stamps: 16 + 64 * PF_MAXL entries
  g_pf_stamps = dev;
  unsigned long long* w = dev ? dev + 16 : nullptr;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pf_wave_stamps), &w, sizeof(w));
}
#endif
static int fill_args(PolicyArgs& a, const float* obs_dev, int64_t n, int32_t num_obs, const double* obs_mean_dev, const double* obs_var_dev, float obs_eps,
                     int32_t num_hidden, const void* const* hidden_w_f16_dev, const void* const* hidden_b_f16_dev, const int32_t* hidden_width,
                     const void* mu_w_f16_dev, const void* mu_b_f16_dev, int32_t num_actions, const void* value_w_f16_dev, const void* value_b_f16_dev) {
  if (!obs_dev || n <= 0 || num_obs <= 0 || num_obs > PF_MAXW || num_hidden <= 0 || num_hidden > PF_MAXL || !hidden_w_f16_dev || !hidden_b_f16_dev ||
      !hidden_width || !mu_w_f16_dev || !mu_b_f16_dev || !value_w_f16_dev || !value_b_f16_dev || num_actions <= 0 || num_actions > 31 ||
      (obs_mean_dev && !obs_var_dev)) return -1;
  a.obs = obs_dev; a.n = n; a.d_in = num_obs; a.mean = obs_mean_dev; a.var = obs_var_dev; a.eps = obs_eps; a.nhid = num_hidden;
  for (int i = 0; i < PF_MAXL; ++i) { a.w[i] = nullptr; a.b[i] = nullptr; a.width[i] = 0; }
  for (int i = 0; i < num_hidden; ++i) {
    if (!hidden_w_f16_dev[i] || !hidden_b_f16_dev[i] || hidden_width[i] <= 0 || hidden_width[i] > PF_MAXW) return -1;
    a.w[i] = (const _Float16*)hidden_w_f16_dev[i]; a.b[i] = (const _Float16*)hidden_b_f16_dev[i]; a.width[i] = hidden_width[i];
  }
  a.w_mu = (const _Float16*)mu_w_f16_dev; a.b_mu = (const _Float16*)mu_b_f16_dev; a.num_actions = num_actions;
  a.w_val = (const _Float16*)value_w_f16_dev; a.b_val = (const _Float16*)value_b_f16_dev;
  a.mu = nullptr; a.value = nullptr;
  a.logstd = a.noise = a.dones = nullptr; a.vmean = a.vvar = nullptr; a.veps = 0.f;
  a.mb_obs = a.mb_dones = a.mb_mu = a.mb_val = a.act = a.act_env = a.neglogp = a.sigma = nullptr;
  a.post = BezPpoRolloutPost{};
  a.an = BezPpoActionNoise{};
  a.dr_on = 0;
  a.ld_obs = num_obs; a.ld_act = num_actions; a.ld_one = 1;
  a.x0_out = nullptr;
  for (int i = 0; i < PF_MAXL; ++i) a.act_out[i] = nullptr;
  a.packed = 0;
#ifdef BEZ_PF_STAMPS
  a.stamps = g_pf_stamps;
#endif
  return 0;
}

// Forward-only / rollout launches of at most 8192 rows (<= 128 tiles of 64 on 256 CUs) run 32-row tiles: twice the workgroups, every
// column block one MFMA chain.  BEZ_PF_ROWS=64 (read once) keeps the 64-row tiles for A/B runs.
static bool small_batch(int64_t n) {
  static const bool off = [] { const char* e = std::getenv("BEZ_PF_ROWS"); return e && std::atoi(e) == 64; }();
  return !off && n <= 8192;
}

extern "C" int bez_ppo_policy_forward(const float* obs_dev, int64_t n, int32_t num_obs, const double* obs_mean_dev, const double* obs_var_dev, float obs_eps,
                                      int32_t num_hidden, const void* const* hidden_w_f16_dev, const void* const* hidden_b_f16_dev, const int32_t* hidden_width,
                                      const void* mu_w_f16_dev, const void* mu_b_f16_dev, int32_t num_actions, const void* value_w_f16_dev,
                                      const void* value_b_f16_dev, float* mu_dev, float* value_dev, int32_t weights_packed, void* stream) {
  PolicyArgs a;
  if (!mu_dev || !value_dev || fill_args(a, obs_dev, n, num_obs, obs_mean_dev, obs_var_dev, obs_eps, num_hidden, hidden_w_f16_dev, hidden_b_f16_dev, hidden_width,
                                         mu_w_f16_dev, mu_b_f16_dev, num_actions, value_w_f16_dev, value_b_f16_dev)) return -1;
  a.mu = mu_dev; a.value = value_dev; a.packed = weights_packed;
  if (weights_packed && small_batch(n)) hipLaunchKernelGGL((policy_forward_kernel<0, PF_LD, PF_LD, true, 32>), dim3((unsigned)((n + 31) / 32)), dim3(PF_FWD_WAVES * 64), 0, (hipStream_t)stream, a);
  else if (weights_packed) hipLaunchKernelGGL((policy_forward_kernel<0, PF_LD, PF_LD, true>), dim3((unsigned)((n + PF_ROWS - 1) / PF_ROWS)), dim3(PF_FWD_WAVES * 64), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((policy_forward_kernel<0, PF_LD, PF_LD, false>), dim3((unsigned)((n + PF_ROWS - 1) / PF_ROWS)), dim3(PF_FWD_WAVES * 64), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int bez_ppo_policy_rollout_step(const float* obs_dev, int64_t n, int32_t num_obs, const double* obs_mean_dev, const double* obs_var_dev, float obs_eps,
                                           int32_t num_hidden, const void* const* hidden_w_f16_dev, const void* const* hidden_b_f16_dev,
                                           const int32_t* hidden_width, const void* mu_w_f16_dev, const void* mu_b_f16_dev, int32_t num_actions,
                                           const void* value_w_f16_dev, const void* value_b_f16_dev, const float* logstd_dev, const float* noise_dev,
                                           const float* dones_dev, const double* value_mean_dev, const double* value_var_dev, float value_eps, float* mb_obs_dev,
                                           float* mb_dones_dev, float* mb_mu_dev, float* mb_val_dev, float* actions_dev, float* env_actions_dev,
                                           float* neglogp_dev, float* sigma_dev, int32_t weights_packed, const BezPpoRolloutPost* prev_post, const BezPpoActionNoise* action_noise,
                                           const void* dr_step, const BezPpoRolloutLayout* layout, void* stream) {
  PolicyArgs a;
  if (prev_post && (!prev_post->rew || !prev_post->reset || !prev_post->timeouts || !prev_post->prev_values || !prev_post->shaped || !prev_post->dones_f ||
                    !prev_post->cur_rew || !prev_post->cur_len || !prev_post->ep_stats)) return -1;
  if (!logstd_dev || !noise_dev || !dones_dev || !mb_obs_dev || !mb_dones_dev || !mb_mu_dev || !mb_val_dev || !actions_dev || !env_actions_dev || !neglogp_dev ||
      !sigma_dev || (value_mean_dev && !value_var_dev) ||
      fill_args(a, obs_dev, n, num_obs, obs_mean_dev, obs_var_dev, obs_eps, num_hidden, hidden_w_f16_dev, hidden_b_f16_dev, hidden_width, mu_w_f16_dev,
                mu_b_f16_dev, num_actions, value_w_f16_dev, value_b_f16_dev)) return -1;
  a.logstd = logstd_dev; a.noise = noise_dev; a.dones = dones_dev; a.vmean = value_mean_dev; a.vvar = value_var_dev; a.veps = value_eps;
  a.mb_obs = mb_obs_dev; a.mb_dones = mb_dones_dev; a.mb_mu = mb_mu_dev; a.mb_val = mb_val_dev; a.act = actions_dev; a.act_env = env_actions_dev;
  a.neglogp = neglogp_dev; a.sigma = sigma_dev; a.packed = weights_packed;
  if (prev_post) a.post = *prev_post;
  if (action_noise && action_noise->snap_dev) a.an = *action_noise;
  if (layout) {
    if (layout->obs_row_stride < num_obs || layout->act_row_stride < num_actions || layout->scalar_stride < 1) return -1;
    a.ld_obs = layout->obs_row_stride; a.ld_act = layout->act_row_stride; a.ld_one = layout->scalar_stride;
  }
  static_assert(sizeof(bez::dr::DrArgs) <= BEZ_DR_STEP_BYTES, "BezPpoDrStep blob of the C ABI too small");
  if (dr_step) { std::memcpy(&a.dr, dr_step, sizeof(a.dr)); a.dr_on = 1; if (a.dr.n <= 0 || !a.dr.st || a.dr.first) return -1; }
  const bool small = weights_packed && small_batch(n);
  const int rows = small ? 32 : PF_ROWS;
  const unsigned grid = (unsigned)((n + rows - 1) / rows) + (dr_step ? 1u : 0u);   // + the randomisation workgroup
  if (small) hipLaunchKernelGGL((policy_forward_kernel<1, PF_LD, PF_LD, true, 32>), dim3(grid), dim3(PF_FWD_WAVES * 64), 0, (hipStream_t)stream, a);
  else if (weights_packed) hipLaunchKernelGGL((policy_forward_kernel<1, PF_LD, PF_LD, true>), dim3(grid), dim3(PF_FWD_WAVES * 64), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((policy_forward_kernel<1, PF_LD, PF_LD, false>), dim3(grid), dim3(PF_FWD_WAVES * 64), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int bez_ppo_policy_forward_train(const float* obs_dev, int64_t n, int32_t num_obs, const double* obs_mean_dev, const double* obs_var_dev, float obs_eps,
                                            int32_t num_hidden, const void* const* hidden_w_f16_dev, const void* const* hidden_b_f16_dev,
                                            const int32_t* hidden_width, const void* mu_w_f16_dev, const void* mu_b_f16_dev, int32_t num_actions,
                                            const void* value_w_f16_dev, const void* value_b_f16_dev, void* x0_f16_dev, void* const* act_f16_dev,
                                            float* mu_dev, float* value_dev, int32_t weights_packed, void* stream) {
  PolicyArgs a;
  if (!mu_dev || !value_dev || !x0_f16_dev || !act_f16_dev || (num_obs & 1) ||
      fill_args(a, obs_dev, n, num_obs, obs_mean_dev, obs_var_dev, obs_eps, num_hidden, hidden_w_f16_dev, hidden_b_f16_dev, hidden_width, mu_w_f16_dev,
                mu_b_f16_dev, num_actions, value_w_f16_dev, value_b_f16_dev)) return -1;
  for (int i = 0; i < num_hidden; ++i) {
    if (!act_f16_dev[i] || (hidden_width[i] & 1)) return -1;  // half2 stores: even widths
    a.act_out[i] = (_Float16*)act_f16_dev[i];
  }
  a.x0_out = (_Float16*)x0_f16_dev; a.mu = mu_dev; a.value = value_dev; a.packed = weights_packed;
  // tile 0 holds the input and the outputs of the odd layers, tile 1 those of the even layers: with strides (216, 424) the tiles are
  // 80 KB and two workgroups share a CU
  int w0 = num_obs, w1 = 0;
  for (int i = 0; i < num_hidden; ++i) { int& w = (i & 1) ? w0 : w1; if (hidden_width[i] > w) w = hidden_width[i]; }
  const dim3 grid((unsigned)((n + PF_ROWS - 1) / PF_ROWS)), block(PF_WAVES * 64);
  const bool narrow = ((w0 + 15) & ~15) + 8 <= 216;
  if (narrow && weights_packed) hipLaunchKernelGGL((policy_forward_kernel<2, 216, PF_LD, true>), grid, block, 0, (hipStream_t)stream, a);
  else if (narrow) hipLaunchKernelGGL((policy_forward_kernel<2, 216, PF_LD, false>), grid, block, 0, (hipStream_t)stream, a);
  else if (weights_packed) hipLaunchKernelGGL((policy_forward_kernel<2, PF_LD, PF_LD, true>), grid, block, 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((policy_forward_kernel<2, PF_LD, PF_LD, false>), grid, block, 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

constexpr int PF_FUSED_LOSS_ACTIONS = 18;   // the action width the fused-loss backward kernel is instantiated for (bez: 18 joints)
static int policy_backward_impl(const BezPpoLossOperands* loss, const float* grad_mu_dev, const float* grad_value_dev, int64_t n, int32_t num_hidden, const int32_t* hidden_width,
                                int32_t num_actions, const void* const* act_f16_dev, const void* const* wt_f16_dev, const void* heads_t_f16_dev,
                                void* const* gz_f16_dev, void* grad_mu_f16_dev, void* grad_value_f16_dev, float* const* bias_grad_dev,
                                float* mu_bias_grad_dev, float* value_bias_grad_dev, float* partial_dev, int32_t weights_packed_flags, void* stream) {
  const int32_t weights_packed = weights_packed_flags & 1;
  if (!partial_dev) return -1;
  if ((!loss && (!grad_mu_dev || !grad_value_dev)) || n <= 0 || num_hidden <= 0 || num_hidden > PF_MAXL || !hidden_width || num_actions <= 0 || num_actions > 31 ||
      !act_f16_dev || !wt_f16_dev || !heads_t_f16_dev || !gz_f16_dev || !grad_mu_f16_dev || !grad_value_f16_dev || !bias_grad_dev || !mu_bias_grad_dev ||
      !value_bias_grad_dev) return -1;
  BackwardArgs a;
  a.n = n; a.nhid = num_hidden; a.num_actions = num_actions; a.gmu = grad_mu_dev; a.gval = grad_value_dev;
  for (int i = 0; i < PF_MAXL; ++i) { a.width[i] = 0; a.act[i] = nullptr; a.wt[i] = nullptr; a.gz[i] = nullptr; a.bgrad[i] = nullptr; }
  for (int i = 0; i < num_hidden; ++i) {
    // even widths (half2 rows), at most one column pair per thread, and at least 32 columns of K for the 16-byte fragment reads
    if (hidden_width[i] < 32 || hidden_width[i] > PF_MAXW || (hidden_width[i] & 1) || !act_f16_dev[i] || !gz_f16_dev[i] || !bias_grad_dev[i] ||
        (i > 0 && !wt_f16_dev[i])) return -1;
    a.width[i] = hidden_width[i]; a.act[i] = (const _Float16*)act_f16_dev[i]; a.wt[i] = (const _Float16*)wt_f16_dev[i];
    a.gz[i] = (_Float16*)gz_f16_dev[i]; a.bgrad[i] = bias_grad_dev[i];
  }
  a.wht = (const _Float16*)heads_t_f16_dev; a.gmu16 = (_Float16*)grad_mu_f16_dev; a.gv16 = (_Float16*)grad_value_f16_dev;
  a.bmu_grad = mu_bias_grad_dev; a.bv_grad = value_bias_grad_dev;
  a.partial = partial_dev; a.ptotal = 0; a.packed = weights_packed;
  for (int i = 0; i < PF_MAXL; ++i) a.poff[i] = 0;
  for (int i = 0; i < num_hidden; ++i) { a.poff[i] = a.ptotal; a.ptotal += hidden_width[i]; }
  a.prow = a.ptotal + 32;
  const unsigned nwg = (unsigned)((n + PF_ROWS - 1) / PF_ROWS);
  int w0 = 32, w1 = 0;  // widest tenant of each tile (see policy_backward_kernel)
  for (int i = 0; i < num_hidden; ++i) { int& w = ((num_hidden - 1 - i) & 1) ? w0 : w1; if (hidden_width[i] > w) w = hidden_width[i]; }
  const bool narrow = ((w0 + 15) & ~15) + 8 <= 216;
  if (loss) {
    // the fused-loss kernel exists for the production shape only: bez's action width, fragment-major weights, the two-workgroups-per-CU
    // tiles, deferred (fixed-order) reductions.  -3: the caller keeps bez_ppo_loss + bez_ppo_policy_backward.
    if (num_actions != PF_FUSED_LOSS_ACTIONS || !narrow || !weights_packed || !(weights_packed_flags & 2) || !loss->scratch_dev) return -3;
    if (!loss->mu_dev || !loss->logstd_dev || !loss->value_dev || !loss->actions_dev || !loss->old_logp_dev || !loss->adv_dev || !loss->old_value_dev ||
        !loss->returns_dev || !loss->old_mu_dev || !loss->old_sigma_dev) return -1;
    a.loss = bez_loss::LossArgs{loss->mu_dev, loss->logstd_dev, loss->value_dev, loss->actions_dev, loss->old_logp_dev, loss->adv_dev, loss->old_value_dev,
                                loss->returns_dev, loss->old_mu_dev, loss->old_sigma_dev, n, loss->e_clip, loss->critic_coef, loss->entropy_coef,
                                loss->bounds_coef, (int)(loss->clip_value & 9), loss->loss_scale_dev, nullptr, nullptr, nullptr, nullptr, loss->scratch_dev};
    hipLaunchKernelGGL((policy_backward_kernel<216, PF_LD, true, PF_FUSED_LOSS_ACTIONS>), dim3(nwg), dim3(PF_WAVES * 64), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
  }
  a.loss = bez_loss::LossArgs{};
  if (narrow && weights_packed) hipLaunchKernelGGL((policy_backward_kernel<216, PF_LD, true>), dim3(nwg), dim3(PF_WAVES * 64), 0, (hipStream_t)stream, a);
  else if (narrow) hipLaunchKernelGGL((policy_backward_kernel<216, PF_LD, false>), dim3(nwg), dim3(PF_WAVES * 64), 0, (hipStream_t)stream, a);
  else if (weights_packed) hipLaunchKernelGGL((policy_backward_kernel<PF_LD, PF_LD, true>), dim3(nwg), dim3(PF_WAVES * 64), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((policy_backward_kernel<PF_LD, PF_LD, false>), dim3(nwg), dim3(PF_WAVES * 64), 0, (hipStream_t)stream, a);
  if (!(weights_packed_flags & 2))  // bit 1: the per-workgroup column sums only -- bez_ppo_grad_reduce_all adds them with the step's other reductions
    hipLaunchKernelGGL(policy_bias_reduce_kernel, dim3((unsigned)((a.ptotal + num_actions + 1 + 63) / 64)), dim3(1024), 0, (hipStream_t)stream, a, (int)nwg);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int bez_ppo_policy_backward(const float* grad_mu_dev, const float* grad_value_dev, int64_t n, int32_t num_hidden, const int32_t* hidden_width,
                                       int32_t num_actions, const void* const* act_f16_dev, const void* const* wt_f16_dev, const void* heads_t_f16_dev,
                                       void* const* gz_f16_dev, void* grad_mu_f16_dev, void* grad_value_f16_dev, float* const* bias_grad_dev,
                                       float* mu_bias_grad_dev, float* value_bias_grad_dev, float* partial_dev, int32_t weights_packed_flags, void* stream) {
  return policy_backward_impl(nullptr, grad_mu_dev, grad_value_dev, n, num_hidden, hidden_width, num_actions, act_f16_dev, wt_f16_dev, heads_t_f16_dev, gz_f16_dev,
                              grad_mu_f16_dev, grad_value_f16_dev, bias_grad_dev, mu_bias_grad_dev, value_bias_grad_dev, partial_dev, weights_packed_flags, stream);
}

extern "C" int bez_ppo_policy_backward_with_loss(const BezPpoLossOperands* loss, int64_t n, int32_t num_hidden, const int32_t* hidden_width, int32_t num_actions,
                                                 const void* const* act_f16_dev, const void* const* wt_f16_dev, const void* heads_t_f16_dev,
                                                 void* const* gz_f16_dev, void* grad_mu_f16_dev, void* grad_value_f16_dev, float* const* bias_grad_dev,
                                                 float* mu_bias_grad_dev, float* value_bias_grad_dev, float* partial_dev, int32_t weights_packed_flags, void* stream) {
  if (!loss) return -1;
  return policy_backward_impl(loss, nullptr, nullptr, n, num_hidden, hidden_width, num_actions, act_f16_dev, wt_f16_dev, heads_t_f16_dev, gz_f16_dev,
                              grad_mu_f16_dev, grad_value_f16_dev, bias_grad_dev, mu_bias_grad_dev, value_bias_grad_dev, partial_dev, weights_packed_flags, stream);
}

extern "C" int bez_ppo_scatter2_f16(const void* src_f16_dev, const int32_t* map_a_dev, const int32_t* map_b_dev, int64_t n, void* dst_f16_dev, void* stream) {
  if (!src_f16_dev || !map_a_dev || !map_b_dev || !dst_f16_dev || n <= 0) return -1;
  hipLaunchKernelGGL(scatter2_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)src_f16_dev, map_a_dev, map_b_dev, n,
                     (_Float16*)dst_f16_dev);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int bez_ppo_scatter_f16(const void* src_f16_dev, const int32_t* map_dev, int64_t n, void* dst_f16_dev, void* stream) {
  if (!src_f16_dev || !map_dev || !dst_f16_dev || n <= 0) return -1;
  hipLaunchKernelGGL(scatter_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)src_f16_dev, map_dev, n,
                     (_Float16*)dst_f16_dev);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

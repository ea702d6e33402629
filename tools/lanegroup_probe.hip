// lanegroup_probe.hip -- a measurement for the NEXT redesign of the step kernel, not product code (tools/lanegroup_probe.py builds and runs it).
//
// The fused control step is bound by the instruction stream of one leg role: the articulated-body recursion of a six-joint
// chain (pass 2) is ~13 k of a substep's ~27 k cycles, one env per lane, every 6x6 operation spelled out in that lane.  SURVEY.md 7 names the
// alternative -- "lane-group mapping": an env's 6x6 rows spread over a group of lanes, cross-lane sums through DPP -- and this probe prices it on
// the recursion itself, with the same arithmetic in both mappings:
//     per joint i = 5..0:   IA += LI_i;  pA += pAl_i;  U = IA S_i;  D = S_i.U + arm;  u = tau_i - S_i.pA;
//                           IA -= U U^T / D;  pA += IA cb_i + U u / D
//   A  one lane per env, IA as a symmetric 6x6 (21 floats), as the product kernel holds it (csrc/bez_spatial.h);
//   B  eight lanes per env: lane r < 6 holds row r of IA and pA[r]; the two dot products are 3-step DPP sums (row_half_mirror + two
//      quad_perms), U is shared through six ds_swizzle broadcasts.
// Output per env: pA (6) and the sum of IA -- compared between the mappings by the driver; s_memtime per wave around the chain.
#include <hip/hip_runtime.h>
#include <cstdint>

constexpr int NJ = 6;

// ---- inputs: [joint][env][...] row-major blocks: LI (36, full symmetric matrix), pAl (6), S (6), cb (6), tau (1)
struct Inputs { const float* LI; const float* pAl; const float* S; const float* cb; const float* tau; int n; float arm; };

// ------------------------------------------------------------------------------------------------ mapping A
struct Sym6 { float a[21]; };   // upper triangle, row-major: (0,0) (0,1) .. (0,5) (1,1) ..
__device__ __forceinline__ constexpr int tri(int r, int c) { return r <= c ? r * 6 - r * (r - 1) / 2 + (c - r) : c * 6 - c * (c - 1) / 2 + (r - c); }

constexpr int NE = 36 + 6 + 6 + 6 + 1;   // floats per (joint, env): LI, pAl, S, cb, tau
__global__ __launch_bounds__(64) void chain_one_lane(Inputs in, float* __restrict__ out, unsigned long long* __restrict__ cycles, int reps) {
  __shared__ float sh[NJ][NE][64];   // the chain's inputs, lane-contiguous (the product kernel holds them in registers / LDS slots; staging is not timed)
  const int lane = threadIdx.x, e = blockIdx.x * 64 + lane, ee = e < in.n ? e : 0;
  for (int j = 0; j < NJ; ++j) {
    const size_t b = (size_t)j * in.n + ee;
    for (int k = 0; k < 36; ++k) sh[j][k][lane] = in.LI[b * 36 + k];
    for (int k = 0; k < 6; ++k) { sh[j][36 + k][lane] = in.pAl[b * 6 + k]; sh[j][42 + k][lane] = in.S[b * 6 + k]; sh[j][48 + k][lane] = in.cb[b * 6 + k]; }
    sh[j][54][lane] = in.tau[b];
  }
  __syncthreads();
  Sym6 IA;
  float pA[6];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
    for (int k = 0; k < 21; ++k) IA.a[k] = 0.f;
#pragma unroll
    for (int k = 0; k < 6; ++k) pA[k] = 0.f;
#pragma unroll
    for (int j = NJ - 1; j >= 0; --j) {
      float S[6], cb[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        S[r] = sh[j][42 + r][lane]; cb[r] = sh[j][48 + r][lane]; pA[r] += sh[j][36 + r][lane];
#pragma unroll
        for (int c = r; c < 6; ++c) IA.a[tri(r, c)] += sh[j][r * 6 + c][lane];
      }
      float U[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 6; ++c) s = fmaf(IA.a[tri(r, c)], S[c], s);
        U[r] = s;
      }
      float D = in.arm, sp = 0.f;
#pragma unroll
      for (int r = 0; r < 6; ++r) { D = fmaf(S[r], U[r], D); sp = fmaf(S[r], pA[r], sp); }
      const float Dinv = __builtin_amdgcn_rcpf(D), uD = (sh[j][54][lane] - sp) * Dinv;
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = r; c < 6; ++c) IA.a[tri(r, c)] = fmaf(-U[r] * Dinv, U[c], IA.a[tri(r, c)]);
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        float s = U[r] * uD;
#pragma unroll
        for (int c = 0; c < 6; ++c) s = fmaf(IA.a[tri(r, c)], cb[c], s);
        pA[r] += s;
      }
    }
    asm volatile("" : "+v"(pA[0]));   // (one chain per repetition: nothing is hoisted across repetitions)
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (e >= in.n) return;
  float tot = 0.f;
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < 6; ++c) tot += IA.a[tri(r, c)];
#pragma unroll
  for (int r = 0; r < 6; ++r) out[(size_t)e * 7 + r] = pA[r];
  out[(size_t)e * 7 + 6] = tot;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

// ------------------------------------------------------------------------------------------------ mapping A2
// one lane per env, the FULL 6x6 as 18 register pairs and every 6-vector as three pairs: packed fp32 math (v_pk_fma_f32 / v_pk_mul_f32 /
// v_pk_add_f32).  A lone wave issues a v_pk_fma_f32 in the time of a v_fma_f32 (tools/pk_issue_probe.py: 4.95 vs 5.72 cycles), so where the
// product kernel's leg role is bound by the instruction count of its stream, two FMAs per instruction are worth up to a factor two -- IF the
// data sits in aligned pairs without shuffles.  (Symmetry is given up: 36 elements instead of 21.)
using f32x2 = __attribute__((ext_vector_type(2))) float;
__global__ __launch_bounds__(64) void chain_one_lane_packed(Inputs in, float* __restrict__ out, unsigned long long* __restrict__ cycles, int reps) {
  __shared__ float sh[NJ][NE][64];
  const int lane = threadIdx.x, e = blockIdx.x * 64 + lane, ee = e < in.n ? e : 0;
  for (int j = 0; j < NJ; ++j) {
    const size_t b = (size_t)j * in.n + ee;
    for (int k = 0; k < 36; ++k) sh[j][k][lane] = in.LI[b * 36 + k];
    for (int k = 0; k < 6; ++k) { sh[j][36 + k][lane] = in.pAl[b * 6 + k]; sh[j][42 + k][lane] = in.S[b * 6 + k]; sh[j][48 + k][lane] = in.cb[b * 6 + k]; }
    sh[j][54][lane] = in.tau[b];
  }
  __syncthreads();
  f32x2 IA[6][3], pA[3];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int p = 0; p < 3; ++p) IA[r][p] = f32x2{0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 3; ++p) pA[p] = f32x2{0.f, 0.f};
#pragma unroll
    for (int j = NJ - 1; j >= 0; --j) {
      f32x2 S[3], cb[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        S[p] = f32x2{sh[j][42 + 2 * p][lane], sh[j][43 + 2 * p][lane]};
        cb[p] = f32x2{sh[j][48 + 2 * p][lane], sh[j][49 + 2 * p][lane]};
        pA[p] += f32x2{sh[j][36 + 2 * p][lane], sh[j][37 + 2 * p][lane]};
      }
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int p = 0; p < 3; ++p) IA[r][p] += f32x2{sh[j][r * 6 + 2 * p][lane], sh[j][r * 6 + 2 * p + 1][lane]};
      // U = IA S: per row three packed products, then the two halves
      float U[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        f32x2 t = IA[r][0] * S[0];
        t = __builtin_elementwise_fma(IA[r][1], S[1], t);
        t = __builtin_elementwise_fma(IA[r][2], S[2], t);
        U[r] = t.x + t.y;
      }
      const f32x2 U2[3] = {f32x2{U[0], U[1]}, f32x2{U[2], U[3]}, f32x2{U[4], U[5]}};
      f32x2 d2 = S[0] * U2[0], s2 = S[0] * pA[0];
      d2 = __builtin_elementwise_fma(S[1], U2[1], d2); s2 = __builtin_elementwise_fma(S[1], pA[1], s2);
      d2 = __builtin_elementwise_fma(S[2], U2[2], d2); s2 = __builtin_elementwise_fma(S[2], pA[2], s2);
      const float D = d2.x + d2.y + in.arm, sp = s2.x + s2.y;
      const float Dinv = __builtin_amdgcn_rcpf(D), uD = (sh[j][54][lane] - sp) * Dinv;
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const float k = -U[r] * Dinv;
        const f32x2 kk = {k, k};
#pragma unroll
        for (int p = 0; p < 3; ++p) IA[r][p] = __builtin_elementwise_fma(kk, U2[p], IA[r][p]);
      }
      float add[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        f32x2 t = IA[r][0] * cb[0];
        t = __builtin_elementwise_fma(IA[r][1], cb[1], t);
        t = __builtin_elementwise_fma(IA[r][2], cb[2], t);
        add[r] = t.x + t.y;
      }
      const f32x2 uu = {uD, uD};
#pragma unroll
      for (int p = 0; p < 3; ++p) pA[p] += __builtin_elementwise_fma(U2[p], uu, f32x2{add[2 * p], add[2 * p + 1]});
    }
    asm volatile("" : "+v"(pA[0]));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (e >= in.n) return;
  float tot = 0.f;
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int p = 0; p < 3; ++p) tot += IA[r][p].x + IA[r][p].y;
#pragma unroll
  for (int p = 0; p < 3; ++p) { out[(size_t)e * 7 + 2 * p] = pA[p].x; out[(size_t)e * 7 + 2 * p + 1] = pA[p].y; }
  out[(size_t)e * 7 + 6] = tot;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

// ------------------------------------------------------------------------------------------------ mapping A3
// one lane per env, the full 6x6 as COLUMN pairs P[c][rp] = (M[2 rp][c], M[2 rp + 1][c]): y = M x is 18 v_pk_fma_f32 whose x operand is one
// half of a register pair broadcast by op_sel (no moves, no horizontal adds -- the result pairs ARE (y[2 rp], y[2 rp + 1])); the rank-1 update
// is 18 more with the scalar k u[c] broadcast the same way.
#define PK_FMA_BLO(d, a, b) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(d) : "v"(a), "v"(b))                 /* d += a * b.lo */
#define PK_FMA_BHI(d, a, b) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(d) : "v"(a), "v"(b))  /* d += a * b.hi */
__global__ __launch_bounds__(64) void chain_one_lane_colpairs(Inputs in, float* __restrict__ out, unsigned long long* __restrict__ cycles, int reps) {
  __shared__ float sh[NJ][NE][64];
  const int lane = threadIdx.x, e = blockIdx.x * 64 + lane, ee = e < in.n ? e : 0;
  for (int j = 0; j < NJ; ++j) {
    const size_t b = (size_t)j * in.n + ee;
    for (int k = 0; k < 36; ++k) sh[j][k][lane] = in.LI[b * 36 + k];
    for (int k = 0; k < 6; ++k) { sh[j][36 + k][lane] = in.pAl[b * 6 + k]; sh[j][42 + k][lane] = in.S[b * 6 + k]; sh[j][48 + k][lane] = in.cb[b * 6 + k]; }
    sh[j][54][lane] = in.tau[b];
  }
  __syncthreads();
  f32x2 P[6][3], pA[3];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
    for (int c = 0; c < 6; ++c)
#pragma unroll
      for (int p = 0; p < 3; ++p) P[c][p] = f32x2{0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 3; ++p) pA[p] = f32x2{0.f, 0.f};
#pragma unroll
    for (int j = NJ - 1; j >= 0; --j) {
      f32x2 S[3], cb[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        S[p] = f32x2{sh[j][42 + 2 * p][lane], sh[j][43 + 2 * p][lane]};
        cb[p] = f32x2{sh[j][48 + 2 * p][lane], sh[j][49 + 2 * p][lane]};
        pA[p] += f32x2{sh[j][36 + 2 * p][lane], sh[j][37 + 2 * p][lane]};
      }
#pragma unroll
      for (int c = 0; c < 6; ++c)
#pragma unroll
        for (int p = 0; p < 3; ++p) P[c][p] += f32x2{sh[j][(2 * p) * 6 + c][lane], sh[j][(2 * p + 1) * 6 + c][lane]};
      // U = M S
      f32x2 U[3] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
#pragma unroll
      for (int c = 0; c < 6; ++c)
#pragma unroll
        for (int p = 0; p < 3; ++p) { if (c & 1) PK_FMA_BHI(U[p], P[c][p], S[c >> 1]); else PK_FMA_BLO(U[p], P[c][p], S[c >> 1]); }
      f32x2 d2 = S[0] * U[0], s2 = S[0] * pA[0];
      d2 = __builtin_elementwise_fma(S[1], U[1], d2); s2 = __builtin_elementwise_fma(S[1], pA[1], s2);
      d2 = __builtin_elementwise_fma(S[2], U[2], d2); s2 = __builtin_elementwise_fma(S[2], pA[2], s2);
      const float D = d2.x + d2.y + in.arm, sp = s2.x + s2.y;
      const float Dinv = __builtin_amdgcn_rcpf(D), uD = (sh[j][54][lane] - sp) * Dinv;
      // M -= U U^T / D: column c gets (-u_c / D) U
      f32x2 kU[3];
      const f32x2 nd = {-Dinv, -Dinv};
#pragma unroll
      for (int p = 0; p < 3; ++p) kU[p] = U[p] * nd;
#pragma unroll
      for (int c = 0; c < 6; ++c)
#pragma unroll
        for (int p = 0; p < 3; ++p) { if (c & 1) PK_FMA_BHI(P[c][p], U[p], kU[c >> 1]); else PK_FMA_BLO(P[c][p], U[p], kU[c >> 1]); }
      // pA += M cb + U uD
      const f32x2 uu = {uD, uD};
#pragma unroll
      for (int p = 0; p < 3; ++p) pA[p] = __builtin_elementwise_fma(U[p], uu, pA[p]);
#pragma unroll
      for (int c = 0; c < 6; ++c)
#pragma unroll
        for (int p = 0; p < 3; ++p) { if (c & 1) PK_FMA_BHI(pA[p], P[c][p], cb[c >> 1]); else PK_FMA_BLO(pA[p], P[c][p], cb[c >> 1]); }
    }
    asm volatile("" : "+v"(pA[0]));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (e >= in.n) return;
  float tot = 0.f;
#pragma unroll
  for (int c = 0; c < 6; ++c)
#pragma unroll
    for (int p = 0; p < 3; ++p) tot += P[c][p].x + P[c][p].y;
#pragma unroll
  for (int p = 0; p < 3; ++p) { out[(size_t)e * 7 + 2 * p] = pA[p].x; out[(size_t)e * 7 + 2 * p + 1] = pA[p].y; }
  out[(size_t)e * 7 + 6] = tot;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

// ------------------------------------------------------------------------------------------------ mapping B
__device__ __forceinline__ float group8_sum(float v) {   // every lane of an aligned group of 8 ends with the group's sum
  int x = __builtin_bit_cast(int, v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x141 /* row_half_mirror */, 0xF, 0xF, true));
  x = __builtin_bit_cast(int, v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true));
  x = __builtin_bit_cast(int, v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x4E /* quad_perm [2,3,0,1] */, 0xF, 0xF, true));
  return v;
}
template <int C>
__device__ __forceinline__ float group8_bcast(float v) {   // lane C of every aligned group of 8 (ds_swizzle, bit mode: and 0x18, or C)
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (C << 5) | 0x18));
}

__global__ __launch_bounds__(64) void chain_lane_group(Inputs in, float* __restrict__ out, unsigned long long* __restrict__ cycles, int reps) {
  __shared__ float sh[NJ][8][NE + 1];   // (env-major rows, padded)
  const int lane = threadIdx.x, r = lane & 7, g = lane >> 3;
  const int e = blockIdx.x * 8 + g;      // eight envs per wave
  const bool live = r < 6 && e < in.n;
  const int ee = e < in.n ? e : 0, rr = r < 6 ? r : 0;
  for (int j = 0; j < NJ; ++j) {
    const size_t b = (size_t)j * in.n + ee;
    for (int k = r; k < 36; k += 8) sh[j][g][k] = in.LI[b * 36 + k];
    if (r < 6) { sh[j][g][36 + r] = in.pAl[b * 6 + r]; sh[j][g][42 + r] = in.S[b * 6 + r]; sh[j][g][48 + r] = in.cb[b * 6 + r]; }
    if (r == 6) sh[j][g][54] = in.tau[b];
  }
  __syncthreads();
  float a[6], pA = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
    for (int c = 0; c < 6; ++c) a[c] = 0.f;
    pA = 0.f;
#pragma unroll
    for (int j = NJ - 1; j >= 0; --j) {
      const float* row = sh[j][g];
      float S[6], cb[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) { S[c] = row[42 + c]; cb[c] = row[48 + c]; a[c] += live ? row[rr * 6 + c] : 0.f; }
      pA += live ? row[36 + rr] : 0.f;
      const float Sr = live ? row[42 + rr] : 0.f;
      float U = 0.f;
#pragma unroll
      for (int c = 0; c < 6; ++c) U = fmaf(a[c], S[c], U);
      const float D = group8_sum(Sr * U) + in.arm, sp = group8_sum(Sr * pA);
      const float Dinv = __builtin_amdgcn_rcpf(D), uD = (row[54] - sp) * Dinv;
      const float Ud = -U * Dinv;
      float Uc[6];
      Uc[0] = group8_bcast<0>(U); Uc[1] = group8_bcast<1>(U); Uc[2] = group8_bcast<2>(U);
      Uc[3] = group8_bcast<3>(U); Uc[4] = group8_bcast<4>(U); Uc[5] = group8_bcast<5>(U);
      float s = U * uD;
#pragma unroll
      for (int c = 0; c < 6; ++c) { a[c] = fmaf(Ud, Uc[c], a[c]); s = fmaf(a[c], cb[c], s); }
      pA += s;
    }
    asm volatile("" : "+v"(pA));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float rowsum = 0.f;
#pragma unroll
  for (int c = 0; c < 6; ++c) rowsum += a[c];
  const float tot = group8_sum(live ? rowsum : 0.f);
  if (live) out[(size_t)e * 7 + r] = pA;
  if (r == 6 && e < in.n) out[(size_t)e * 7 + 6] = tot;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

// ------------------------------------------------------------------------------------------------ mapping D (round 6)
// Four lanes per env, no MFMA: the 6x6 as 2x2 blocks of 3x3, lane (r, c) = quad lane 2r + c holds block (r, c) in full (9 registers)
// and the row half r of pA.  U_r = sum_c M_rc S_c is one quad_perm [1,0,3,2] add per component, the two scalars S.U and S.pA one
// quad_perm [2,3,0,1] add each, the column half of U for the rank-1 update comes from the other row pair (quad_perm [2,3,0,1]).
// DPP only -- no LDS round trip (mapping B's ds_swizzle), no MFMA pipeline hop (mapping C).
__device__ __forceinline__ float qp_flip_c(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ float qp_flip_r(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)); }
constexpr int NQ = 9 + 3 + 3 + 3 + 3 + 1;   // floats per (joint, lane): its block of LI, pAl_r, S_c, S_r, cb_c, tau
__global__ __launch_bounds__(64) void chain_quad(Inputs in, float* __restrict__ out, unsigned long long* __restrict__ cycles, int reps) {
  __shared__ float sh[NJ][NQ][64];
  const int lane = threadIdx.x, sub = lane & 3, r = sub >> 1, c = sub & 1;
  const int e = blockIdx.x * 16 + (lane >> 2), ee = e < in.n ? e : 0;
  for (int j = 0; j < NJ; ++j) {
    const size_t b = (size_t)j * in.n + ee;
    for (int i = 0; i < 3; ++i) {
      for (int k = 0; k < 3; ++k) sh[j][i * 3 + k][lane] = in.LI[b * 36 + (3 * r + i) * 6 + 3 * c + k];
      sh[j][9 + i][lane] = in.pAl[b * 6 + 3 * r + i];
      sh[j][12 + i][lane] = in.S[b * 6 + 3 * c + i];
      sh[j][15 + i][lane] = in.S[b * 6 + 3 * r + i];
      sh[j][18 + i][lane] = in.cb[b * 6 + 3 * c + i];
    }
    sh[j][21][lane] = in.tau[b];
  }
  __syncthreads();
  const bool diag = r == c;
  float M[9], pA[3];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
    for (int k = 0; k < 9; ++k) M[k] = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) pA[k] = 0.f;
#pragma unroll
    for (int j = NJ - 1; j >= 0; --j) {
      float Sc[3], Sr[3], cbc[3];
#pragma unroll
      for (int k = 0; k < 9; ++k) M[k] += sh[j][k][lane];
#pragma unroll
      for (int k = 0; k < 3; ++k) { pA[k] += sh[j][9 + k][lane]; Sc[k] = sh[j][12 + k][lane]; Sr[k] = sh[j][15 + k][lane]; cbc[k] = sh[j][18 + k][lane]; }
      float u[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) { const float s = fmaf(M[3 * i], Sc[0], fmaf(M[3 * i + 1], Sc[1], M[3 * i + 2] * Sc[2])); u[i] = s + qp_flip_c(s); }
      float d = fmaf(Sr[0], u[0], fmaf(Sr[1], u[1], Sr[2] * u[2])), sp = fmaf(Sr[0], pA[0], fmaf(Sr[1], pA[1], Sr[2] * pA[2]));
      d += qp_flip_r(d); sp += qp_flip_r(sp);
      const float Dinv = __builtin_amdgcn_rcpf(d + in.arm), uD = (sh[j][21][lane] - sp) * Dinv;
      float uc[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) { const float o = qp_flip_r(u[i]); uc[i] = diag ? u[i] : o; }
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float k = -u[i] * Dinv;
#pragma unroll
        for (int q = 0; q < 3; ++q) M[3 * i + q] = fmaf(k, uc[q], M[3 * i + q]);
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float s = fmaf(M[3 * i], cbc[0], fmaf(M[3 * i + 1], cbc[1], M[3 * i + 2] * cbc[2]));
        pA[i] += fmaf(u[i], uD, s + qp_flip_c(s));
      }
    }
    asm volatile("" : "+v"(pA[0]));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float tot = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) tot += M[k];
  tot += qp_flip_c(tot); tot += qp_flip_r(tot);
  if (e >= in.n) return;
  if (c == 0) {
#pragma unroll
    for (int i = 0; i < 3; ++i) out[(size_t)e * 7 + 3 * r + i] = pA[i];
  }
  if (sub == 0) out[(size_t)e * 7 + 6] = tot;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

// ------------------------------------------------------------------------------------------------ mapping C (round 5; north_star: "MFMA only if
// the batched 6x6 spatial-inertia products prove worth it in rocprof")
// FOUR lanes per env, sixteen envs per wave, the matrix padded to 8 x 8 as 2 x 2 blocks of 4 x 4: lane q of an env's group holds COLUMNS q and 4 + q
// (= rows, the matrix is symmetric) in four float4 accumulators acc[R][C] -- exactly the C / D layout of v_mfma_f32_4x4x1_16B_f32 (16 independent
// 4 x 4 blocks per instruction: one per env).  Then
//   * U = IA S and IA cb are LOCAL dot products of a lane's two columns with the (group-uniform) vectors: no cross-lane traffic;
//   * the rank-1 update IA -= U U^T / D is FOUR MFMAs whose A / B operands -- U[4 R + q] and -U[4 C + q] / D -- already sit in lane q: the matrix
//     core does the outer product across the four lanes that mapping B needed six ds_swizzle broadcasts for;
//   * only the two scalars S.U and S.pA cross lanes (two quad_perm DPP steps each).
using f32x4 = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ float quad_sum(float v) {
  int x = __builtin_bit_cast(int, v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true));
  x = __builtin_bit_cast(int, v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x4E /* quad_perm [2,3,0,1] */, 0xF, 0xF, true));
  return v;
}
__global__ __launch_bounds__(64) void chain_mfma4(Inputs in, float* __restrict__ out, unsigned long long* __restrict__ cycles, int reps) {
  // per (joint, env): LI as 8 x 8 (zero padded), then pAl[8], S[8], cb[8], tau: rows of 16-byte aligned floats
  constexpr int ROW = 64 + 8 + 8 + 8 + 4;
  __shared__ __attribute__((aligned(16))) float sh[NJ][16][ROW + 4];
  const int lane = threadIdx.x, q = lane & 3, g = lane >> 2;
  const int e = blockIdx.x * 16 + g;     // sixteen envs per wave
  const int ee = e < in.n ? e : 0;
  for (int j = 0; j < NJ; ++j) {
    const size_t b = (size_t)j * in.n + ee;
    for (int k = q; k < 64; k += 4) { const int r = k >> 3, c = k & 7; sh[j][g][k] = (r < 6 && c < 6) ? in.LI[b * 36 + r * 6 + c] : 0.f; }
    for (int k = q; k < 8; k += 4) {
      sh[j][g][64 + k] = k < 6 ? in.pAl[b * 6 + k] : 0.f; sh[j][g][72 + k] = k < 6 ? in.S[b * 6 + k] : 0.f; sh[j][g][80 + k] = k < 6 ? in.cb[b * 6 + k] : 0.f;
    }
    if (q == 0) sh[j][g][88] = in.tau[b];
  }
  __syncthreads();
  f32x4 acc[2][2];
  float pA[2];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
    for (int R = 0; R < 2; ++R)
#pragma unroll
      for (int C = 0; C < 2; ++C) acc[R][C] = f32x4{0.f, 0.f, 0.f, 0.f};
    pA[0] = pA[1] = 0.f;
#pragma unroll
    for (int j = NJ - 1; j >= 0; --j) {
      const float* row = sh[j][g];
      float S[8], cb[8];
#pragma unroll
      for (int k = 0; k < 8; k += 4) {
        const f32x4 s4 = *reinterpret_cast<const f32x4*>(row + 72 + k), c4 = *reinterpret_cast<const f32x4*>(row + 80 + k);
        S[k] = s4.x; S[k + 1] = s4.y; S[k + 2] = s4.z; S[k + 3] = s4.w; cb[k] = c4.x; cb[k + 1] = c4.y; cb[k + 2] = c4.z; cb[k + 3] = c4.w;
      }
      // IA += LI: this lane's two columns (rows 4 R + i of column 4 C + q)
#pragma unroll
      for (int R = 0; R < 2; ++R)
#pragma unroll
        for (int C = 0; C < 2; ++C)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[R][C][i] += row[(4 * R + i) * 8 + 4 * C + q];
      pA[0] += row[64 + q]; pA[1] += row[68 + q];
      // U[4 C + q] = column (4 C + q) . S      (symmetric matrix: column = row)
      float U[2];
#pragma unroll
      for (int C = 0; C < 2; ++C) {
        float u = 0.f;
#pragma unroll
        for (int R = 0; R < 2; ++R)
#pragma unroll
          for (int i = 0; i < 4; ++i) u = fmaf(acc[R][C][i], S[4 * R + i], u);
        U[C] = u;
      }
      const float Sq0 = row[72 + q], Sq1 = row[76 + q];
      const float D = quad_sum(fmaf(Sq0, U[0], Sq1 * U[1])) + in.arm, sp = quad_sum(fmaf(Sq0, pA[0], Sq1 * pA[1]));
      const float Dinv = __builtin_amdgcn_rcpf(D), uD = (row[88] - sp) * Dinv;
      const float kU0 = -U[0] * Dinv, kU1 = -U[1] * Dinv;
      // rank-1 update on the matrix cores: block (R, C) += U[4 R + .] (x) kU[4 C + .]
      acc[0][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(U[0], kU0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(U[0], kU1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(U[1], kU0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(U[1], kU1, acc[1][1], 0, 0, 0);
      // pA[4 C + q] += column (4 C + q) . cb + U uD
#pragma unroll
      for (int C = 0; C < 2; ++C) {
        float s = U[C] * uD;
#pragma unroll
        for (int R = 0; R < 2; ++R)
#pragma unroll
          for (int i = 0; i < 4; ++i) s = fmaf(acc[R][C][i], cb[4 * R + i], s);
        pA[C] += s;
      }
    }
    asm volatile("" : "+v"(pA[0]));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float colsum = 0.f;
#pragma unroll
  for (int R = 0; R < 2; ++R)
#pragma unroll
    for (int C = 0; C < 2; ++C)
#pragma unroll
      for (int i = 0; i < 4; ++i) colsum += acc[R][C][i];
  const float tot = quad_sum(colsum);
  if (e < in.n) {
    out[(size_t)e * 7 + q] = pA[0];
    if (q < 2) out[(size_t)e * 7 + 4 + q] = pA[1];
    if (q == 3) out[(size_t)e * 7 + 6] = tot;
  }
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

// ------------------------------------------------------------------------------------------------ the OTHER per-joint work (round 5)
// Pass 2 of the product kernel spends ~370 vector instructions per joint, of which the recursion above is ~140.  The rest is per-(env, joint)
// work with no dependence BETWEEN joints: the link's spatial inertia about the common reference point from its mass / COM / rotated inertia, the
// velocity-product bias and gravity wrench, the drive / friction / limit terms with the saturation predictor's operands.  Written here with the
// product's own helpers (csrc/bez_spatial.h) on synthetic link states: mode 0 = one lane per env, the six joints one after the other (the
// product's mapping); mode 1 = one lane per (env, joint), eight lanes per env.  Outputs go to LDS as the packages the recursion consumes.
namespace bsp {
#include "../bez_isaacgym_amd/csrc/bez_spatial.h"
}
struct JointOut { float v[34]; };
__device__ __forceinline__ void joint_package(const float* __restrict__ src, float h, JointOut& o) {
  using namespace bsp;
  using bsp::Sym6;
  // src: quaternion (4), link origin r (3), joint axis code folded into a unit vector (3), V (6), q, qd, target, mass, com (3), inertia diag (3)
  const M3 E = quat_to_mat(src[0], src[1], src[2], src[3]);
  const V3 r = mk(src[4], src[5], src[6]), ax = mul(E, mk(src[7], src[8], src[9]));
  const SV V = mksv(mk(src[10], src[11], src[12]), mk(src[13], src[14], src[15]));
  const float q = src[16], qd = src[17], tgt = src[18], m = src[19];
  const V3 cw = r + mul(E, mk(src[20], src[21], src[22]));
  Sym6 I;
  I.A = rotate_inertia_diag(E, src[23], src[24], src[25]);
  // + m [c]x [c]x^T ; B = m [c]x ; C = m 1
  I.A.xx += m * (cw.y * cw.y + cw.z * cw.z); I.A.yy += m * (cw.x * cw.x + cw.z * cw.z); I.A.zz += m * (cw.x * cw.x + cw.y * cw.y);
  I.A.xy -= m * cw.x * cw.y; I.A.xz -= m * cw.x * cw.z; I.A.yz -= m * cw.y * cw.z;
  I.B = m3zero(); I.B.m01 = -m * cw.z; I.B.m02 = m * cw.y; I.B.m10 = m * cw.z; I.B.m12 = -m * cw.x; I.B.m20 = -m * cw.y; I.B.m21 = m * cw.x;
  I.C = sym3zero(); I.C.xx = I.C.yy = I.C.zz = m;
  const SV S = mksv(ax, cross(r, ax));
  const SV vj = S * qd, cbias = crm(V, vj);
  SV p = crf(V, mul(I, V)) - wrench_at(cw, mk(0.f, 0.f, -9.81f * m));
  // drive: implicit PD, regularised friction, limit spring (the product's terms), and the saturation predictor's operands
  const float kp = 100.f, kd = 7.5f;
  const float tau_pd0 = kp * (tgt - q - h * qd) - kd * qd, k_pd = h * h * kp + h * kd;
  const float cf = 0.1f * frcp(fmaxf(fabsf(qd), 0.1f)), k_f = h * cf, tau_f0 = -cf * qd;
  float k_l = 0.f, tau_l0 = 0.f;
  if (q < -0.78f) { tau_l0 = 200.f * (-0.78f - q - h * qd) - 2.f * qd; k_l = h * h * 200.f + h * 2.f; }
  else if (q > 1.57f) { tau_l0 = 200.f * (1.57f - q - h * qd) - 2.f * qd; k_l = h * h * 200.f + h * 2.f; }
  const SV U = mul(I, S);
  const float J = dot(S, U) + 1e-3f, bias = dot(S, p) + dot(U, cbias);
  const float qdd_est = (tau_pd0 + tau_f0 + tau_l0 - bias) * frcp(J + k_pd + k_f + k_l);
  const float tau_drive = tau_pd0 - k_pd * qdd_est;
  const bool sat = fabsf(tau_drive) > 2.5f;
  const float tau = sat ? copysignf(2.5f, tau_drive) + tau_f0 + tau_l0 : tau_pd0 + tau_f0 + tau_l0;
  const float kdiag = sat ? k_f + k_l : k_pd + k_f + k_l;
  float* v = o.v;
  v[0] = I.A.xx; v[1] = I.A.yy; v[2] = I.A.zz; v[3] = I.A.xy; v[4] = I.A.xz; v[5] = I.A.yz;
  v[6] = I.B.m00; v[7] = I.B.m01; v[8] = I.B.m02; v[9] = I.B.m10; v[10] = I.B.m11; v[11] = I.B.m12; v[12] = I.B.m20; v[13] = I.B.m21; v[14] = I.B.m22;
  v[15] = m; v[16] = p.a.x; v[17] = p.a.y; v[18] = p.a.z; v[19] = p.l.x; v[20] = p.l.y; v[21] = p.l.z;
  v[22] = S.a.x; v[23] = S.a.y; v[24] = S.a.z; v[25] = S.l.x; v[26] = S.l.y; v[27] = S.l.z;
  v[28] = cbias.a.x + cbias.l.x; v[29] = cbias.a.y + cbias.l.y; v[30] = cbias.a.z + cbias.l.z; v[31] = tau; v[32] = kdiag; v[33] = J;
}
template <int MODE>
__global__ __launch_bounds__(64) void joint_work(const float* __restrict__ links, int n, float* __restrict__ out, unsigned long long* __restrict__ cycles, int reps) {
  __shared__ float sh[NJ][26][64];         // mode 0: [joint][field][env lane]
  __shared__ float pk[NJ * 64][35];        // packages (padded)
  const int lane = threadIdx.x;
  const int epw = MODE == 0 ? 64 : 8;
  const int e0 = blockIdx.x * epw;
  for (int j = 0; j < NJ; ++j)
    for (int l = lane; l < epw * 26; l += 64) { const int el = l / 26, f = l - el * 26; const int e = e0 + el < n ? e0 + el : 0; sh[j][f][el] = links[((size_t)j * n + e) * 26 + f]; }
  __syncthreads();
  float keep = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int rep = 0; rep < reps; ++rep) {
    if (MODE == 0) {
#pragma unroll 1
      for (int j = 0; j < NJ; ++j) {
        float src[26];
#pragma unroll
        for (int f = 0; f < 26; ++f) src[f] = sh[j][f][lane];
        JointOut o;
        joint_package(src, 1.f / 120.f + 1e-9f * rep, o);
#pragma unroll
        for (int k = 0; k < 34; ++k) pk[j * 64 + lane][k] = o.v[k];
        keep += o.v[31];
      }
    } else {
      const int j = lane & 7, el = lane >> 3;
      if (j < NJ) {
        float src[26];
#pragma unroll
        for (int f = 0; f < 26; ++f) src[f] = sh[j][f][el];
        JointOut o;
        joint_package(src, 1.f / 120.f + 1e-9f * rep, o);
#pragma unroll
        for (int k = 0; k < 34; ++k) pk[j * 64 + el][k] = o.v[k];
        keep += o.v[31];
      }
    }
    asm volatile("" : "+v"(keep));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  // checksum per env: sum over joints and fields of the packages
  if (MODE == 0) {
    const int e = e0 + lane;
    if (e < n) { float s = 0.f; for (int j = 0; j < NJ; ++j) for (int k = 0; k < 34; ++k) s += pk[j * 64 + lane][k]; out[e] = s; }
  } else {
    const int el = lane >> 3, e = e0 + el;
    if ((lane & 7) == 0 && e < n) { float s = 0.f; for (int j = 0; j < NJ; ++j) for (int k = 0; k < 34; ++k) s += pk[j * 64 + el][k]; out[e] = s; }
  }
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  if (keep == 123.456f) out[0] = keep;
}
extern "C" int probe_joint_work(int mode, const float* links, int n, float* out, unsigned long long* cycles, int reps, void* stream) {
  if (mode == 0) hipLaunchKernelGGL(joint_work<0>, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, links, n, out, cycles, reps);
  else hipLaunchKernelGGL(joint_work<1>, dim3((n + 7) / 8), dim3(64), 0, (hipStream_t)stream, links, n, out, cycles, reps);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int probe_run(int mapping, const float* LI, const float* pAl, const float* S, const float* cb, const float* tau, int n, float arm, float* out,
                         unsigned long long* cycles, int reps, void* stream) {
  Inputs in{LI, pAl, S, cb, tau, n, arm};
  if (mapping == 5) hipLaunchKernelGGL(chain_quad, dim3((n + 15) / 16), dim3(64), 0, (hipStream_t)stream, in, out, cycles, reps);
  else if (mapping == 4) hipLaunchKernelGGL(chain_mfma4, dim3((n + 15) / 16), dim3(64), 0, (hipStream_t)stream, in, out, cycles, reps);
  else if (mapping == 3) hipLaunchKernelGGL(chain_one_lane_colpairs, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, in, out, cycles, reps);
  else if (mapping == 2) hipLaunchKernelGGL(chain_one_lane_packed, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, in, out, cycles, reps);
  else if (mapping == 0) hipLaunchKernelGGL(chain_one_lane, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, in, out, cycles, reps);
  else hipLaunchKernelGGL(chain_lane_group, dim3((n + 7) / 8), dim3(64), 0, (hipStream_t)stream, in, out, cycles, reps);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// bez_ppo_loss.h -- the PPO loss of a 64-row tile (device code shared by bez_ppo.hip's loss kernel and bez_policy.hip's backward kernel,
// which runs it in front of its own head stage: one launch and one HBM round trip of d loss / d mu, d loss / d value less per minibatch step).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace bez_loss {

constexpr float LOG_2PI = 1.8378770664093453f;
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

struct LossArgs {
  const float* mu; const float* logstd; const float* value; const float* act; const float* old_logp; const float* adv; const float* old_value;
  const float* ret; const float* old_mu; const float* old_sigma;
  int64_t B; float e_clip, critic_coef, entropy_coef, bounds_coef; int clip_value;
  const float* scale; float* grad_mu; float* grad_value; float* grad_logstd; float* stats; float* scratch;
};

// ---- the PPO loss of one minibatch and its gradient with respect to the network outputs, in one pass.
//   loss = mean(a_loss) + 0.5 * critic_coef * mean(c_loss) - entropy_coef * mean(entropy) + bounds_coef * mean(b_loss)
// stats (accumulated into, caller zeroes): [sum a_loss, sum c_loss, sum b_loss, sum kl, sum entropy]
// grad_mu / grad_value are d(loss)/d(mu), d(loss)/d(value) times *scale (GradScaler's loss scale, a device scalar; null = 1);
// grad_logstd (A) is accumulated into (caller zeroes).  With `scratch` both are fixed-order two-stage sums (per-workgroup partials +
// ppo_loss_reduce_kernel), without it float atomics.
// Layout: the (B,A) row-major operands of a workgroup's 64 consecutive rows are one contiguous block, moved with coalesced accesses
// and laid out row-major in LDS ([row][A+1]: conflict-free for the per-thread row walk).  A is a template parameter (register
// arrays, unrolled loops); the per-column / per-term sums are reduced in the workgroup first, so a launch produces A + 5 partial
// sums per 64 samples.
constexpr int LOSS_TB = 64;   // rows (samples) per workgroup: 512 workgroups for config 3's minibatch
constexpr int LOSS_CW = 4;    // waves per workgroup = column groups: wave w owns the action columns [w * CG, (w + 1) * CG)
// Round 4: 64 rows x 4 column groups per workgroup (256 threads) instead of one thread per row walking all A columns.  A lone wave
// per SIMD issues one vector instruction per ~5 cycles whatever it does, and a row's A x (exp, log, 4 divisions) was a serial chain of
// ~A x 60 instructions: 17.7 us for 32768 x 18 with half the chip's SIMDs idle.  Four waves per 64 rows give every SIMD two waves and each
// wave a quarter of the columns; the three per-row sums that cross columns (sum z^2 -> neglogp, KL, bounds loss) meet in LDS, added in
// the fixed order w = 0..3 (bit-reproducible).
// FUSED: called by a LARGER workgroup (threads >= LOSS_TB * LOSS_CW only pass the barriers) that consumes the gradients itself: d loss / d mu stays
// in tile 0 ([row][A + 1] floats at lds), d loss / d value in the 64 floats behind the partial sums; nothing is written to grad_mu / grad_value.
// lds: loss_lds_floats(A) floats.  blockIdx.x / gridDim.x: the tile index and count, as in the stand-alone launch.
constexpr int loss_lds_floats(int A) { return 4 * LOSS_TB * (A + 1) + 3 * LOSS_CW * LOSS_TB + LOSS_TB; }
template <int A, bool FUSED>
__device__ __forceinline__ void ppo_loss_tile(const LossArgs& L, float* lds, int tid) {
  const float* __restrict__ mu = L.mu; const float* __restrict__ logstd = L.logstd; const float* __restrict__ value = L.value;
  const float* __restrict__ act = L.act; const float* __restrict__ old_logp = L.old_logp; const float* __restrict__ adv = L.adv;
  const float* __restrict__ old_value = L.old_value; const float* __restrict__ ret = L.ret; const float* __restrict__ old_mu = L.old_mu;
  const float* __restrict__ old_sigma = L.old_sigma; const float* __restrict__ scale = L.scale;
  float* __restrict__ grad_mu = L.grad_mu; float* __restrict__ grad_value = L.grad_value; float* __restrict__ grad_logstd = L.grad_logstd;
  float* __restrict__ stats = L.stats; float* __restrict__ scratch = L.scratch;
  const int64_t B = L.B; const float e_clip = L.e_clip, critic_coef = L.critic_coef, entropy_coef = L.entropy_coef, bounds_coef = L.bounds_coef;
  const int clip_value = L.clip_value;
  constexpr int LD = A + 1, TB = LOSS_TB, NT = LOSS_TB * LOSS_CW, CG = (A + LOSS_CW - 1) / LOSS_CW, NL = (TB * A + NT - 1) / NT;
  float (*tile)[TB * LD] = reinterpret_cast<float (*)[TB * LD]>(lds);                                   // the four (64, A) operand blocks, row-major with a padded row
  float (*part)[LOSS_CW][TB] = reinterpret_cast<float (*)[LOSS_CW][TB]>(lds + 4 * TB * LD);              // per column group: sum z^2, KL, bounds loss of each row
  float* gvalue = lds + 4 * TB * LD + 3 * LOSS_CW * TB;                                                // FUSED: d loss / d value of the 64 rows
  const int lane = tid & 63, w = tid >> 6;
  const bool live = !FUSED || tid < NT;   // FUSED: the waves beyond the loss tile's four only pass the barriers
  const int64_t row0 = (int64_t)blockIdx.x * TB;
  const int64_t i = row0 + lane;
  const int nrow = (int)((B - row0) < (int64_t)TB ? (B - row0) : (int64_t)TB);
  const bool on = live && lane < nrow;
  const float S = scale ? scale[0] : 1.0f, invB = 1.0f / (float)B;
  const int c0 = w * CG, c1 = (c0 + CG < A) ? c0 + CG : A;   // this wave's columns
  // the workgroup's contiguous (nrow, A) blocks of the four row-major operands: coalesced loads (element k = tid + j * 256), all in
  // flight at once, then into the LDS tiles
  {
    const float* src[4] = {mu + row0 * A, act + row0 * A, old_mu + row0 * A, old_sigma + row0 * A};
    float raw[4][NL];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < NL; ++j) { const int k = tid + j * NT; raw[q][j] = (live && k < nrow * A) ? src[q][k] : 0.f; }
    if (clip_value & 8) {   // PPODataset.update_mu_sigma [ext]: the policy's current mu / sigma replace the minibatch's old ones (already in registers)
      float* om = const_cast<float*>(old_mu) + row0 * A;
      float* os = const_cast<float*>(old_sigma) + row0 * A;
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        const int k = tid + j * NT;
        if (live && k < nrow * A) { om[k] = raw[0][j]; os[k] = expf(logstd[k % A]); }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < NL; ++j) { const int k = tid + j * NT; if (live && k < TB * A) tile[q][(k / A) * LD + (k % A)] = raw[q][j]; }
  }
  __syncthreads();
  // ---- phase A: this thread's columns of its row
  float m[CG], z[CG], sg[CG];
  float acc = 0.f, klp = 0.f, blp = 0.f;
#pragma unroll
  for (int jj = 0; jj < CG; ++jj) {
    const int j = c0 + jj;
    m[jj] = 0.f; z[jj] = 0.f; sg[jj] = 1.f;
    if (j < c1) {
      const float l = logstd[j], s = expf(l);
      const float mm = tile[0][lane * LD + j], x = tile[1][lane * LD + j], om = tile[2][lane * LD + j], s1 = tile[3][lane * LD + j];
      const float zz = (x - mm) / s;
      m[jj] = mm; z[jj] = zz; sg[jj] = s;
      acc = fmaf(zz, zz, acc);
      const float dm = om - mm;
      klp += logf(s1 / s + 1e-5f) + (s * s + dm * dm) / (2.0f * (s1 * s1 + 1e-5f)) - 0.5f;   // policy_kl(current, old)
      const float hi = fmaxf(mm - 1.1f, 0.f), lo = fminf(mm + 1.1f, 0.f);
      blp += hi * hi + lo * lo;
    }
  }
  if (live) { part[0][w][lane] = acc; part[1][w][lane] = klp; part[2][w][lane] = blp; }
  __syncthreads();
  float accs = 0.f, kl = 0.f, b_l = 0.f;
#pragma unroll
  for (int q = 0; q < LOSS_CW; ++q) { accs += part[0][q][lane]; kl += part[1][q][lane]; b_l += part[2][q][lane]; }
  // ---- the row's scalars (every column group needs g_nlp; the statistics and the value gradient are wave 0's)
  float a_l = 0.f, c_l = 0.f, ent = 0.f, g_nlp = 0.f, g_val = 0.f;
  if (on) {
    float ls = 0.f;
#pragma unroll
    for (int j = 0; j < A; ++j) { const float l = logstd[j]; ls += l; ent += 0.5f + 0.5f * LOG_2PI + l; }
    const float neglogp = 0.5f * accs + 0.5f * LOG_2PI * (float)A + ls;
    const float ratio = expf(old_logp[i] - neglogp), ad = adv[i];
    const float rc = fminf(fmaxf(ratio, 1.0f - e_clip), 1.0f + e_clip);
    const float l1 = -ad * ratio, l2 = -ad * rc;
    a_l = fmaxf(l1, l2);
    // d a_loss / d ratio (torch.max splits ties evenly: with an unclipped ratio both branches carry -adv)
    const bool unclipped = (ratio >= 1.0f - e_clip) && (ratio <= 1.0f + e_clip);
    float g_ratio;
    if (l1 > l2) g_ratio = -ad;
    else if (l1 < l2) g_ratio = unclipped ? -ad : 0.f;
    else g_ratio = 0.5f * (-ad) + 0.5f * (unclipped ? -ad : 0.f);
    g_nlp = -ratio * g_ratio;  // d a_loss / d neglogp
    if (w == 0) {   // value loss
      const float v = value[i], ov = old_value[i], rt = ret[i];
      float g_v;
      if (clip_value & 1) {
        const float dv = v - ov, dvc = fminf(fmaxf(dv, -e_clip), e_clip), vc = ov + dvc;
        const float q1 = (v - rt) * (v - rt), q2 = (vc - rt) * (vc - rt);
        c_l = fmaxf(q1, q2);
        const bool vin = (dv >= -e_clip) && (dv <= e_clip);
        const float g1 = 2.0f * (v - rt), g2 = vin ? 2.0f * (vc - rt) : 0.f;
        g_v = q1 > q2 ? g1 : (q1 < q2 ? g2 : 0.5f * (g1 + g2));
      } else {
        c_l = (rt - v) * (rt - v);
        g_v = 2.0f * (v - rt);
      }
      g_val = 0.5f * critic_coef * g_v * invB * S;
      if (!FUSED) grad_value[i] = g_val;
    }
    if (!(bounds_coef > 0.f)) b_l = 0.f;
  } else { kl = 0.f; b_l = 0.f; }
  // ---- phase B: gradients of this thread's columns; d loss / d mu goes back through tile 0 (coalesced store of the contiguous block),
  // d loss / d log-sigma is summed over the 64 rows in the wave
  float mine = 0.f;
#pragma unroll
  for (int jj = 0; jj < CG; ++jj) {
    const int j = c0 + jj;
    float gm = 0.f, gl = 0.f;
    if (on && j < c1) {
      const float hi = fmaxf(m[jj] - 1.1f, 0.f), lo = fminf(m[jj] + 1.1f, 0.f);
      const float g_b = bounds_coef > 0.f ? bounds_coef * 2.0f * (hi + lo) : 0.f;
      // d neglogp / d mu_j = -z_j / sigma_j ; d neglogp / d logstd_j = 1 - z_j^2
      gm = (g_nlp * (-z[jj] / sg[jj]) + g_b) * invB * S;
      gl = (g_nlp * (1.0f - z[jj] * z[jj]) - entropy_coef) * invB * S;
    }
    if (j < c1) tile[0][lane * LD + j] = gm;   // (each (row, column) slot has one owner; its own read of it is behind it)
    const float g = wave_sum(gl);
    if (lane == jj) mine = g;                  // lane jj of wave w holds column c0 + jj
  }
  if (FUSED) {
    if (w == 0) gvalue[lane] = g_val;   // (0 beyond the last row) -- with tile 0 the caller's head-gradient input, no round trip through HBM
  } else {
    __syncthreads();
    float* blk = grad_mu + row0 * A;
#pragma unroll
    for (int j = 0; j < NL; ++j) { const int k = tid + j * NT; if (k < nrow * A) blk[k] = tile[0][(k / A) * LD + (k % A)]; }
  }
  // the five statistics: wave 0
  float st = 0.f;
  if (w == 0) {
    a_l = wave_sum(a_l); c_l = wave_sum(c_l); b_l = wave_sum(b_l); kl = wave_sum(kl); ent = wave_sum(ent);
    st = lane == CG ? a_l : lane == CG + 1 ? c_l : lane == CG + 2 ? b_l : lane == CG + 3 ? kl : ent;
  }
  // outputs of this workgroup: column c0 + jj from lane jj of wave w (jj < c1 - c0), statistic q from lane CG + q of wave 0
  const bool has_col = lane < c1 - c0, has_st = w == 0 && lane >= CG && lane < CG + 5;
  const int out = has_col ? c0 + lane : A + (lane - CG);
  const float val = has_col ? mine : st;
  if (has_col || has_st) {
  if (!scratch) {   // one float atomic each (A + 5 per 64 samples): the sums then differ in their last bits from run to run
    atomicAdd(out < A ? &grad_logstd[out] : &stats[out - A], val);
  } else {
  // bit-reproducible: every workgroup stores its A + 5 partials (column-major: a column's partials are contiguous); a second stage
  // (ppo_loss_reduce_kernel, or bez_ppo_grad_reduce_all with the step's other reductions) adds them in a fixed order and is the only
  // writer of grad_logstd / stats.  (A last-workgroup-reduces scheme inside this kernel was measured first: its agent-scope fences -- an L2
  // write-back + invalidate per workgroup on the 8-XCD part -- cost 65 us per call.)
  scratch[2 + (size_t)out * gridDim.x + blockIdx.x] = val;
  }
  }
  if (FUSED) __syncthreads();   // tile 0 (d loss / d mu) and gvalue are complete
}

}  // namespace bez_loss

// bez_ppo.hip -- the elementwise / reduction glue of the PPO consumer loop as single gfx950 kernels (C ABI: include/bez_sim.h,
// "bez_ppo_*").  The MLP forward / backward stays in PyTorch-ROCm (hipBLASLt GEMMs); what is fused here is everything around it
// that torch would run as ~250 four-microsecond kernels per minibatch step: the observation normaliser (moments, running
// update, normalise + clamp), the whole PPO loss with its analytic gradient, action sampling and the rollout bookkeeping.
// Semantics restate rl_games' a2c_continuous (bez_isaacgym_amd/ppo/a2c_continuous.py is the readable reference, and the
// tests compare these kernels with it term by term).
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bez_sim.h"
#include "bez_ppo_loss.h"

namespace {

constexpr int PPO_TB = 256;
constexpr float LOG_2PI = 1.8378770664093453f;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- RunningMeanStd (rl_games): per-column sum / sum of squares of x (B,D) in fp64 -> out[0:D], out[D:2D]; out[2D] = B
__global__ __launch_bounds__(PPO_TB) void rms_moments_kernel(const float* __restrict__ x, int64_t B, int D, double* __restrict__ out,
                                                             double* __restrict__ scratch) {
  // 256 threads = 4 row lanes x 64 column lanes (D <= 64)
  const int col = threadIdx.x & 63, rl = threadIdx.x >> 6;
  __shared__ double sh[2][4][64];
  double s1 = 0.0, s2 = 0.0;
  if (col < D) {
#pragma unroll 8
    for (int64_t r = (int64_t)blockIdx.x * 4 + rl; r < B; r += (int64_t)gridDim.x * 4) {  // (unrolled: eight rows in flight per thread)
      const double v = (double)x[r * D + col];
      s1 += v; s2 += v * v;
    }
  }
  sh[0][rl][col] = s1; sh[1][rl][col] = s2;
  __syncthreads();
  if (rl == 0 && col < D) {
    s1 = sh[0][0][col] + sh[0][1][col] + sh[0][2][col] + sh[0][3][col];
    s2 = sh[1][0][col] + sh[1][1][col] + sh[1][2][col] + sh[1][3][col];
    if (!scratch) {   // fp64 atomics: the sums differ in their last bits from run to run
      atomicAdd(&out[col], s1);
      atomicAdd(&out[D + col], s2);
    } else {
      scratch[1 + (size_t)blockIdx.x * 2 * D + col] = s1;
      scratch[1 + (size_t)blockIdx.x * 2 * D + D + col] = s2;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) out[2 * D] = (double)B;
}
// second stage of the fixed-order column sums: a block per 64 of the 2D columns, 16 row lanes x 64 columns -- lane r adds workgroups
// r, r + 16, ... (independent coalesced loads), the lanes meet in LDS in order
__global__ __launch_bounds__(1024) void rms_reduce_kernel(const double* __restrict__ scratch, int D, unsigned int nblocks, double* __restrict__ out) {
  __shared__ double lds[16][64];
  const int W = 2 * D, l = threadIdx.x & 63, r = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + l;
  double acc = 0.0;
  if (c < W) {
#pragma unroll 8
    for (unsigned int b = r; b < nblocks; b += 16) acc += scratch[1 + (size_t)b * W + c];
  }
  lds[r][l] = acc;
  __syncthreads();
  if (r == 0 && c < W) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += lds[q][l];
    out[c] = t;
  }
}

// parallel-variance update of (mean, var, count) from the (possibly all-reduced) moments: a2c_continuous.py RunningMeanStd.update
__global__ void rms_apply_kernel(const double* __restrict__ mom, int D, double* __restrict__ mean, double* __restrict__ var, double* __restrict__ count) {
  const int c = threadIdx.x;
  const double n = mom[2 * D], cnt = count[0], tot = cnt + n;
  if (c < D) {
    const double b_mean = mom[c] / n;
    double b_var = mom[D + c] / n - b_mean * b_mean;
    if (b_var < 0.0) b_var = 0.0;
    b_var *= n / (n - 1.0 > 1.0 ? n - 1.0 : 1.0);  // unbiased, as torch.var
    const double delta = b_mean - mean[c];
    const double m2 = var[c] * cnt + b_var * n + delta * delta * cnt * n / tot;
    mean[c] += delta * n / tot;
    var[c] = m2 / tot;
  }
  __syncthreads();
  if (c == 0) count[0] = tot;
}

// y = clamp((x - mean) / sqrt(var + eps), -5, 5), written as fp32 or fp16 (the autocast input of the first Linear)
template <typename OUT>
__global__ __launch_bounds__(PPO_TB) void rms_normalize_kernel(const float* __restrict__ x, int64_t total, int D, const double* __restrict__ mean,
                                                               const double* __restrict__ var, float eps, OUT* __restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % D);
  const float m = (float)mean[c], v = (float)var[c];
  float t = (x[i] - m) / sqrtf(v + eps);
  t = fminf(fmaxf(t, -5.0f), 5.0f);
  y[i] = (OUT)t;
}

// ---- action sampling (rollout): a = mu + exp(logstd) * noise; neglogp(a); env action = clamp(a, -1, 1)
__global__ __launch_bounds__(PPO_TB) void ppo_sample_kernel(const float* __restrict__ mu, const float* __restrict__ logstd, const float* __restrict__ noise,
                                                            int64_t N, int A, float* __restrict__ act, float* __restrict__ act_env,
                                                            float* __restrict__ neglogp, float* __restrict__ sigma_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float acc = 0.f, ls = 0.f;
  for (int j = 0; j < A; ++j) {
    const float l = logstd[j], s = expf(l), z = noise[i * A + j];
    const float a = fmaf(s, z, mu[i * A + j]);
    act[i * A + j] = a;
    act_env[i * A + j] = fminf(fmaxf(a, -1.0f), 1.0f);
    sigma_out[i * A + j] = s;
    const float zz = (a - mu[i * A + j]) / s;  // as the reference computes it from the stored action
    acc = fmaf(zz, zz, acc); ls += l;
  }
  neglogp[i] = 0.5f * acc + 0.5f * LOG_2PI * (float)A + ls;
}

// ---- everything of one rollout step between the policy forward and the env step, in one launch: the fp32 copies of the
// network outputs, the de-normalised value (RunningMeanStd(unnorm=True): clamp to +-5, x sqrt(var + eps) + mean), the rollout
// buffer rows (obs, dones, mu, value), and the action sampling of ppo_sample_kernel.  IN = __half (explicit fp16 path) or float.
template <typename IN>
__global__ __launch_bounds__(PPO_TB) void ppo_rollout_pre_kernel(const IN* __restrict__ mu_in, const IN* __restrict__ value_in, const float* __restrict__ logstd,
                                                                 const float* __restrict__ noise, const float* __restrict__ obs, const float* __restrict__ dones,
                                                                 const double* __restrict__ vmean, const double* __restrict__ vvar, float veps, int64_t N, int A,
                                                                 int D, float* __restrict__ mb_obs, float* __restrict__ mb_dones, float* __restrict__ mb_mu,
                                                                 float* __restrict__ mb_val, float* __restrict__ act, float* __restrict__ act_env,
                                                                 float* __restrict__ neglogp, float* __restrict__ sigma_out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = t; k < N * D; k += nt) mb_obs[k] = obs[k];  // flat coalesced copy of the observation block
  if (t >= N) return;
  const int64_t i = t;
  mb_dones[i] = dones[i];
  float v = (float)value_in[i];
  if (vmean) v = sqrtf((float)vvar[0] + veps) * fminf(fmaxf(v, -5.0f), 5.0f) + (float)vmean[0];
  mb_val[i] = v;
  float acc = 0.f, ls = 0.f;
  for (int j = 0; j < A; ++j) {
    const float m = (float)mu_in[i * A + j];
    mb_mu[i * A + j] = m;
    const float l = logstd[j], sg = expf(l), z = noise[i * A + j];
    const float a = fmaf(sg, z, m);
    act[i * A + j] = a;
    act_env[i * A + j] = fminf(fmaxf(a, -1.0f), 1.0f);
    sigma_out[i * A + j] = sg;
    const float zz = (a - m) / sg;  // as the reference computes it from the stored action
    acc = fmaf(zz, zz, acc); ls += l;
  }
  neglogp[i] = 0.5f * acc + 0.5f * LOG_2PI * (float)A + ls;
}

// ---- rollout bookkeeping of one env step (a2c_continuous.py _rollout_impl): shaped reward with the time-out bootstrap, done
// flags as floats, running episode return / length and the finished-episode statistics
__global__ __launch_bounds__(PPO_TB) void ppo_rollout_post_kernel(const float* __restrict__ rew, const int64_t* __restrict__ dones, const int64_t* __restrict__ timeouts,
                                                                  const float* __restrict__ values, int64_t N, float reward_scale, float gamma, int bootstrap,
                                                                  float* __restrict__ shaped, float* __restrict__ dones_f, float* __restrict__ cur_rew,
                                                                  float* __restrict__ cur_len, double* __restrict__ ep_stats, double* __restrict__ ep_parts,
                                                                  int nslots) {
  double c = 0.0, r = 0.0, l = 0.0;
  if (ep_parts && blockIdx.x == gridDim.x - 1) {
    // one extra workgroup (bez_ppo_rollout_post_fold): the per-workgroup slots the policy launches of this rollout filled (BezPpoRolloutPost.ep_parts)
    // are added to ep_stats and cleared
    for (int s = (int)threadIdx.x; s < nslots; s += (int)blockDim.x) {
      c += ep_parts[4 * s]; r += ep_parts[4 * s + 1]; l += ep_parts[4 * s + 2];
      ep_parts[4 * s] = 0.0; ep_parts[4 * s + 1] = 0.0; ep_parts[4 * s + 2] = 0.0;
    }
    c = wave_sum(c); r = wave_sum(r); l = wave_sum(l);
    if ((threadIdx.x & 63) == 0 && c != 0.0) { atomicAdd(&ep_stats[0], c); atomicAdd(&ep_stats[1], r); atomicAdd(&ep_stats[2], l); }
    return;
  }
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) {
    const float rw = rew[i];
    float s = rw * reward_scale;
    if (bootstrap) s += gamma * values[i] * (float)timeouts[i];
    shaped[i] = s;
    const float d = (float)dones[i];
    dones_f[i] = d;
    const float cr = cur_rew[i] + rw, cl = cur_len[i] + 1.0f;
    c = d; r = cr * d; l = cl * d;
    cur_rew[i] = cr * (1.0f - d); cur_len[i] = cl * (1.0f - d);
  }
  c = wave_sum(c); r = wave_sum(r); l = wave_sum(l);
  if ((threadIdx.x & 63) == 0 && c != 0.0) { atomicAdd(&ep_stats[0], c); atomicAdd(&ep_stats[1], r); atomicAdd(&ep_stats[2], l); }
}

// (the loss of a 64-row tile lives in bez_ppo_loss.h: the backward kernel of csrc/bez_policy.hip runs the same code in front of its head stage)
using bez_loss::LOSS_TB;
using bez_loss::LOSS_CW;
template <int A>
__global__ __launch_bounds__(LOSS_TB * LOSS_CW) void ppo_loss_kernel(bez_loss::LossArgs L) {
  __shared__ float lds[bez_loss::loss_lds_floats(A)];
  bez_loss::ppo_loss_tile<A, false>(L, lds, (int)threadIdx.x);
}
// second stage of the fixed-order sums: one wave per column (blockIdx.x = column); lane l adds workgroups l, l + 64, ... (independent,
// coalesced loads), then the butterfly
__global__ __launch_bounds__(64) void ppo_loss_reduce_kernel(const float* __restrict__ scratch, int A, unsigned int nblocks,
                                                             float* __restrict__ grad_logstd, float* __restrict__ stats) {
  const int tid = threadIdx.x, c = blockIdx.x;
  const float* col = scratch + 2 + (size_t)c * nblocks;
  float acc = 0.f;
#pragma unroll 8
  for (unsigned int b = tid; b < nblocks; b += 64) acc += col[b];
  acc = wave_sum(acc);
  if (tid == 0) { if (c < A) grad_logstd[c] += acc; else stats[c - A] += acc; }
}

// ---- gradient reductions of the explicit-fp16 linear layers (a2c_continuous.py _HalfLinearFn), written straight into the fp32
// master gradient: (1) the sum over the S split-K partial products of dW = dY^T X (fp16, [S][n]), (2) the bias gradient = column
// sums of dY (fp16, [B][D]).  Replaces torch's .float() copy + sum(0) + AccumulateGrad add per tensor.
// 256 threads = 64 element pairs x 4 split lanes: each thread sums every 4th partial of its half2 (independent loads, unrolled),
// the four split lanes meet in LDS (odd n: scalar loads).
__global__ __launch_bounds__(PPO_TB) void wgrad_sum_kernel(const __half* __restrict__ part, int S, int64_t n, float* __restrict__ out, int accumulate) {
  __shared__ float2 sh[4][64];
  const int l = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int64_t i = ((int64_t)blockIdx.x * 64 + l) * 2;
  float2 a = make_float2(0.f, 0.f);
  if ((n & 1) == 0) {  // every partial starts 4-byte aligned: half2 loads
    if (i < n) {
#pragma unroll 8
      for (int s = sl; s < S; s += 4) {
        const float2 v = __half22float2(*reinterpret_cast<const __half2*>(part + (int64_t)s * n + i));
        a.x += v.x; a.y += v.y;
      }
    }
  } else {
    for (int s = sl; s < S; s += 4) {
      if (i < n) a.x += __half2float(part[(int64_t)s * n + i]);
      if (i + 1 < n) a.y += __half2float(part[(int64_t)s * n + i + 1]);
    }
  }
  sh[sl][l] = a;
  __syncthreads();
  if (sl == 0 && i < n) {
    float2 t = make_float2((sh[0][l].x + sh[1][l].x) + (sh[2][l].x + sh[3][l].x), (sh[0][l].y + sh[1][l].y) + (sh[2][l].y + sh[3][l].y));
    if (accumulate) { t.x += out[i]; if (i + 1 < n) t.y += out[i + 1]; }
    out[i] = t.x;
    if (i + 1 < n) out[i + 1] = t.y;
  }
}
// grid (column tiles of 128, row slabs); 256 threads = 4 row lanes x 64 column pairs (half2 loads, 8 rows in flight per thread);
// partial column sums meet in `out` by atomics (one per column per slab).  D must be even for the half2 path (odd D: scalar).
__global__ __launch_bounds__(PPO_TB) void colsum_half_kernel(const __half* __restrict__ y, int64_t B, int D, int64_t rows_per_block, float* __restrict__ out) {
  __shared__ float2 sh[4][64];
  const int l = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + l) * 2;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = (r0 + rows_per_block < B) ? r0 + rows_per_block : B;
  float2 a = make_float2(0.f, 0.f);
  if ((D & 1) == 0) {
    if (c < D) {
#pragma unroll 8
      for (int64_t r = r0 + rl; r < r1; r += 4) {
        const float2 v = __half22float2(*reinterpret_cast<const __half2*>(y + r * D + c));
        a.x += v.x; a.y += v.y;
      }
    }
  } else {
    for (int64_t r = r0 + rl; r < r1; r += 4) {
      if (c < D) a.x += __half2float(y[r * D + c]);
      if (c + 1 < D) a.y += __half2float(y[r * D + c + 1]);
    }
  }
  sh[rl][l] = a;
  __syncthreads();
  if (rl == 0 && c < D) {
    atomicAdd(&out[c], (sh[0][l].x + sh[1][l].x) + (sh[2][l].x + sh[3][l].x));
    if (c + 1 < D) atomicAdd(&out[c + 1], (sh[0][l].y + sh[1][l].y) + (sh[2][l].y + sh[3][l].y));
  }
}

// ELU backward + bias gradient in one pass over dY: gz = gy * (y > 0 ? 1 : y + 1)  (alpha = 1; y = the layer's ELU output), written
// as fp16 for the following GEMMs, and its column sums accumulated into the fp32 bias gradient.  Same tiling as colsum_half_kernel.
constexpr int ELU_RL = 8;  // row lanes of the ELU-backward kernel: 512 threads = 8 x 64 column pairs, eight rows in flight per thread
__global__ __launch_bounds__(ELU_RL * 64) void elu_bwd_colsum_kernel(const __half* __restrict__ gy, const __half* __restrict__ y, __half* __restrict__ gz, int64_t B,
                                                                     int D, int64_t rows_per_block, float* __restrict__ out) {
  __shared__ float2 sh[ELU_RL][64];
  const int l = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + l) * 2;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = (r0 + rows_per_block < B) ? r0 + rows_per_block : B;
  float2 a = make_float2(0.f, 0.f);
  if ((D & 1) == 0) {
    if (c < D) {
#pragma unroll 8
      for (int64_t r = r0 + rl; r < r1; r += ELU_RL) {
        const float2 g = __half22float2(*reinterpret_cast<const __half2*>(gy + r * D + c));
        const float2 v = __half22float2(*reinterpret_cast<const __half2*>(y + r * D + c));
        const __half2 z = __floats2half2_rn(g.x * (v.x > 0.f ? 1.f : v.x + 1.f), g.y * (v.y > 0.f ? 1.f : v.y + 1.f));
        *reinterpret_cast<__half2*>(gz + r * D + c) = z;
        const float2 zf = __half22float2(z);  // the bias gradient sums what the GEMMs will see
        a.x += zf.x; a.y += zf.y;
      }
    }
  } else {
    for (int64_t r = r0 + rl; r < r1; r += ELU_RL) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        if (c + k < D) {
          const float g = __half2float(gy[r * D + c + k]), v = __half2float(y[r * D + c + k]);
          const __half z = __float2half(g * (v > 0.f ? 1.f : v + 1.f));
          gz[r * D + c + k] = z;
          (k ? a.y : a.x) += __half2float(z);
        }
      }
    }
  }
  sh[rl][l] = a;
  __syncthreads();
  if (rl == 0 && c < D) {
    float sx = 0.f, sy = 0.f;
#pragma unroll
    for (int q = 0; q < ELU_RL; ++q) { sx += sh[q][l].x; sy += sh[q][l].y; }
    atomicAdd(&out[c], sx);
    if (c + 1 < D) atomicAdd(&out[c + 1], sy);
  }
}

// ---- the two heads' backward inputs in one pass: fp16 copies of d loss / d mu (B,A) and d loss / d value (B,1) (what autocast's cast
// nodes hand the head GEMMs) and the column sums of those fp16 values = the head bias gradients (accumulated)
__global__ __launch_bounds__(PPO_TB) void head_grads_kernel(const float* __restrict__ gmu, const float* __restrict__ gval, int64_t B, int A,
                                                            __half* __restrict__ gmu16, __half* __restrict__ gv16, float* __restrict__ bmu_grad,
                                                            float* __restrict__ bv_grad, int64_t rows_per_block) {
  __shared__ float sh[PPO_TB];
  const int tid = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = (r0 + rows_per_block < B) ? r0 + rows_per_block : B;
  // thread tid owns column tid % A of rows tid / A, tid / A + TB / A, ...: consecutive threads on consecutive elements of the slab
  const int rows_par = PPO_TB / A;  // rows a pass of the workgroup covers
  const int col = tid % A, rsub = tid / A;
  float acc = 0.f;
  if (rsub < rows_par) {
#pragma unroll 4
    for (int64_t r = r0 + rsub; r < r1; r += rows_par) {
      const __half h = __float2half(gmu[r * A + col]);
      gmu16[r * A + col] = h;
      acc += __half2float(h);
    }
  }
  sh[tid] = acc;
  float vacc = 0.f;
  for (int64_t r = r0 + tid; r < r1; r += PPO_TB) {
    const __half h = __float2half(gval[r]);
    gv16[r] = h;
    vacc += __half2float(h);
  }
  __syncthreads();
  if (tid < A) {
    float s = 0.f;
    for (int q = 0; q < rows_par; ++q) s += sh[q * A + tid];
    atomicAdd(&bmu_grad[tid], s);
  }
  vacc = wave_sum(vacc);
  if ((tid & 63) == 0) atomicAdd(bv_grad, vacc);
}

// ---- rl_games AdaptiveScheduler.update on device scalars: lr /= 1.5 (floor min_lr) when kl > 2 thr, lr *= 1.5 (cap max_lr) when
// kl < thr / 2, in the reference's order (both tests see the lr the first one left)
__global__ void adaptive_lr_kernel(float* __restrict__ lr, const float* __restrict__ kl, float thr, float min_lr, float max_lr) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float v = lr[0];
  const float k = kl[0];
  if (k > 2.0f * thr) v = fmaxf(v / 1.5f, min_lr);
  if (k < 0.5f * thr) v = fminf(v * 1.5f, max_lr);
  lr[0] = v;
}

// ---- GAE backward scan over the horizon (rl_games a2c_common.discount_values): one thread per env walks t = H-1 .. 0 with the
// same fp32 operations, in the same order, as the 8 elementwise launches per step of the torch formulation
__global__ __launch_bounds__(PPO_TB) void gae_kernel(const float* __restrict__ rew, const float* __restrict__ val, const float* __restrict__ mb_dones,
                                                     const float* __restrict__ dones, const float* __restrict__ last_values, int H, int64_t N,
                                                     float gamma, float tau, float* __restrict__ advs, float* __restrict__ returns,
                                                     const double* __restrict__ vmean, const double* __restrict__ vvar, float veps) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float last = 0.f;
  float nnt = 1.0f - dones[i], nv = last_values[i];
  if (vmean) {   // the bootstrap values come straight from the network: RunningMeanStd(unnorm=True) here instead of five elementwise launches
    const float sd = sqrtf((float)vvar[0] + veps), y = fminf(fmaxf(nv, -5.0f), 5.0f);
    nv = __fadd_rn(__fmul_rn(sd, y), (float)vmean[0]);   // (product, then sum: torch's two roundings)
  }
  // eight steps of loads in flight per thread, then their scan: the chain itself is 32 dependent fused multiply-adds, the loads are not part of it
  constexpr int CH = 8;
  for (int t1 = H; t1 > 0; t1 -= CH) {
    float v[CH], r[CH], d[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int t = t1 - 1 - u;
      if (t >= 0) { v[u] = val[(int64_t)t * N + i]; r[u] = rew[(int64_t)t * N + i]; d[u] = mb_dones[(int64_t)t * N + i]; }
    }
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int t = t1 - 1 - u;
      if (t >= 0) {
        const float delta = r[u] + gamma * nv * nnt - v[u];
        last = delta + gamma * tau * nnt * last;
        advs[(int64_t)t * N + i] = last;
        if (returns) returns[(int64_t)t * N + i] = last + v[u];
        nnt = 1.0f - d[u]; nv = v[u];  // for step t-1: "next" = step t
      }
    }
  }
}

// ---- the epoch's dataset preparation between GAE and the first minibatch (rl_games a2c_continuous.prepare_dataset + the per-minibatch
// input-normaliser moments) in FOUR launches instead of ~30 small ones (6 x moments + 6 x reduce, 2 x apply, 2 x normalise, two
// transposes, difference, mean, Welford, three elementwise, three copies: ~150 us of launch latencies per epoch):
//   1  column sums / sums of squares of every minibatch's observation rows AND of the values / returns (seen as 64 virtual columns),
//      per-workgroup partials in fp64 (grid.y = task);
//   2  their fixed-order second stage -> the (2 D + 1) moments of each task;
//   3  values / returns through the value normaliser as RunningMeanStd.forward does in train mode -- update with the values' moments,
//      normalise the values, update with the returns' moments, normalise the returns (every workgroup forms the two updates itself from
//      the moments and the OLD statistics; nobody writes them here) --, transposed from the rollout's (H, N) into the dataset's env-major
//      rows, advantage = return - value, per-workgroup partial sums of the advantage (fp64);
//   4  every workgroup adds those partials in the same fixed order, (adv - mean) / (std + 1e-8) with torch's unbiased std in place; workgroup 0
//      commits the value normaliser's new statistics.
struct PrepTask { const float* x; long long rows; int cols; double* out; };
constexpr int PREP_MAXT = 10;
struct PrepTasks { PrepTask t[PREP_MAXT]; int n; };
__global__ __launch_bounds__(PPO_TB) void prep_moments_kernel(PrepTasks T, double* __restrict__ scratch) {
  const PrepTask& t = T.t[blockIdx.y];
  const int D = t.cols, col = threadIdx.x & 63, rl = threadIdx.x >> 6;
  __shared__ double sh[2][4][64];
  double s1 = 0.0, s2 = 0.0;
  if (col < D) {
#pragma unroll 8
    for (long long r = (long long)blockIdx.x * 4 + rl; r < t.rows; r += (long long)gridDim.x * 4) {
      const double v = (double)t.x[r * D + col];
      s1 += v; s2 += v * v;
    }
  }
  sh[0][rl][col] = s1; sh[1][rl][col] = s2;
  __syncthreads();
  if (rl == 0) {
    double* dst = scratch + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 128;
    dst[col] = col < D ? sh[0][0][col] + sh[0][1][col] + sh[0][2][col] + sh[0][3][col] : 0.0;
    dst[64 + col] = col < D ? sh[1][0][col] + sh[1][1][col] + sh[1][2][col] + sh[1][3][col] : 0.0;
  }
}
// one workgroup per task: 16 row lanes x 64 columns add the partials of workgroups r, r + 16, ... (fixed order); a task with `fold` set is a
// scalar seen as 64 virtual columns, whose column sums are then added in lane order
__global__ __launch_bounds__(1024) void prep_reduce_kernel(PrepTasks T, const double* __restrict__ scratch, int nblocks, int fold_from) {
  __shared__ double lds[2][16][64];
  const PrepTask& t = T.t[blockIdx.x];
  const int l = threadIdx.x & 63, r = threadIdx.x >> 6;
  double a1 = 0.0, a2 = 0.0;
  for (int b = r; b < nblocks; b += 16) {
    const double* src = scratch + ((size_t)blockIdx.x * nblocks + b) * 128;
    a1 += src[l]; a2 += src[64 + l];
  }
  lds[0][r][l] = a1; lds[1][r][l] = a2;
  __syncthreads();
  if (r == 0) {
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) { t1 += lds[0][q][l]; t2 += lds[1][q][l]; }
    lds[0][0][l] = t1; lds[1][0][l] = t2;
  }
  __syncthreads();
  if ((int)blockIdx.x >= fold_from) {   // scalar task
    if (threadIdx.x == 0) {
      double t1 = 0.0, t2 = 0.0;
      for (int q = 0; q < 64; ++q) { t1 += lds[0][0][q]; t2 += lds[1][0][q]; }
      t.out[0] = t1; t.out[1] = t2; t.out[2] = (double)(t.rows * t.cols);
    }
  } else if (r == 0) {
    if (l < t.cols) { t.out[l] = lds[0][0][l]; t.out[t.cols + l] = lds[1][0][l]; }
    if (l == 0) t.out[2 * t.cols] = (double)t.rows;
  }
}
// RunningMeanStd.update_from_moments on scalars (rms_apply_kernel's arithmetic)
struct Rms1 { double mean, var, count; };
__device__ __forceinline__ Rms1 rms1_apply(Rms1 s, const double* mom) {
  const double n = mom[2], tot = s.count + n;
  const double b_mean = mom[0] / n;
  double b_var = mom[1] / n - b_mean * b_mean;
  if (b_var < 0.0) b_var = 0.0;
  b_var *= n / (n - 1.0 > 1.0 ? n - 1.0 : 1.0);
  const double delta = b_mean - s.mean;
  const double m2 = s.var * s.count + b_var * n + delta * delta * s.count * n / tot;
  Rms1 o;
  o.mean = s.mean + delta * n / tot; o.var = m2 / tot; o.count = tot;
  return o;
}
__global__ __launch_bounds__(PPO_TB) void prep_values_kernel(const float* __restrict__ values, const float* __restrict__ returns, int H, long long N,
                                                             const double* __restrict__ vmean, const double* __restrict__ vvar, const double* __restrict__ vcount,
                                                             float eps, const double* __restrict__ val_mom, const double* __restrict__ ret_mom,
                                                             float* __restrict__ old_values, float* __restrict__ ds_returns, float* __restrict__ adv,
                                                             double* __restrict__ partial) {
  float m1 = 0.f, d1 = 1.f, m2 = 0.f, d2 = 1.f;
  const bool norm = vmean != nullptr;
  if (norm) {
    Rms1 s{vmean[0], vvar[0], vcount[0]};
    s = rms1_apply(s, val_mom); m1 = (float)s.mean; d1 = sqrtf((float)s.var + eps);
    s = rms1_apply(s, ret_mom); m2 = (float)s.mean; d2 = sqrtf((float)s.var + eps);
  }
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x, total = (long long)H * N;
  double a1 = 0.0, a2 = 0.0;
  if (j < total) {
    const long long e = j / H;
    const int t = (int)(j - e * H);
    float v = values[(long long)t * N + e], r = returns[(long long)t * N + e];
    if (norm) {
      v = fminf(fmaxf((v - m1) / d1, -5.0f), 5.0f);
      r = fminf(fmaxf((r - m2) / d2, -5.0f), 5.0f);
    }
    old_values[j] = v; ds_returns[j] = r;
    const float a = r - v;
    adv[j] = a;
    a1 = (double)a; a2 = (double)a * (double)a;
  }
  __shared__ double sh[2][PPO_TB / 64];
  a1 = wave_sum(a1); a2 = wave_sum(a2);
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = a1; sh[1][threadIdx.x >> 6] = a2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t1 = 0.0, t2 = 0.0;
    for (int q = 0; q < PPO_TB / 64; ++q) { t1 += sh[0][q]; t2 += sh[1][q]; }
    partial[2 * blockIdx.x] = t1; partial[2 * blockIdx.x + 1] = t2;
  }
}
// data parallel: the rank's (sum adv, sum adv^2, count) from the per-workgroup partials, in the order prep_advantage_kernel adds them
__global__ __launch_bounds__(PPO_TB) void prep_adv_sums_kernel(const double* __restrict__ partial, int nparts, long long total, double* __restrict__ out) {
  __shared__ double sh[2][PPO_TB];
  double a1 = 0.0, a2 = 0.0;
  for (int q = threadIdx.x; q < nparts; q += PPO_TB) { a1 += partial[2 * q]; a2 += partial[2 * q + 1]; }
  sh[0][threadIdx.x] = a1; sh[1][threadIdx.x] = a2;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t1 = 0.0, t2 = 0.0;
    for (int q = 0; q < PPO_TB; ++q) { t1 += sh[0][q]; t2 += sh[1][q]; }
    out[0] = t1; out[1] = t2; out[2] = (double)total;
  }
}
// `global` (data parallel): (sum adv, sum adv^2, count) of the WHOLE job's batch, all-reduced; NULL: this launch adds the partials itself
__global__ __launch_bounds__(PPO_TB) void prep_advantage_kernel(float* __restrict__ adv, long long total, const double* __restrict__ partial, int nparts, int normalize,
                                                                double* __restrict__ vmean, double* __restrict__ vvar, double* __restrict__ vcount,
                                                                const double* __restrict__ val_mom, const double* __restrict__ ret_mom,
                                                                const double* __restrict__ global) {
  __shared__ double sh[2][PPO_TB];
  __shared__ float ms[2];
  if (normalize) {
    double a1 = 0.0, a2 = 0.0;
    if (!global) for (int q = threadIdx.x; q < nparts; q += PPO_TB) { a1 += partial[2 * q]; a2 += partial[2 * q + 1]; }
    sh[0][threadIdx.x] = a1; sh[1][threadIdx.x] = a2;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t1 = 0.0, t2 = 0.0;
      if (global) { t1 = global[0]; t2 = global[1]; }
      else for (int q = 0; q < PPO_TB; ++q) { t1 += sh[0][q]; t2 += sh[1][q]; }
      const double n = global ? global[2] : (double)total, mean = t1 / n;
      double var = (t2 - t1 * mean) / (n - 1.0 > 1.0 ? n - 1.0 : 1.0);   // unbiased, as torch.std
      if (var < 0.0) var = 0.0;
      ms[0] = (float)mean; ms[1] = (float)sqrt(var);
    }
    __syncthreads();
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < total) adv[j] = (adv[j] - ms[0]) / (ms[1] + 1e-8f);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && vmean) {   // the value normaliser's statistics after both updates
    Rms1 s{vmean[0], vvar[0], vcount[0]};
    s = rms1_apply(s, val_mom);
    s = rms1_apply(s, ret_mom);
    vmean[0] = s.mean; vvar[0] = s.var; vcount[0] = s.count;
  }
}

// ---- the optimiser tail of one minibatch step on the flat fp32 buffers: GradScaler.unscale_ + clip_grad_norm_ + Adam (torch
// semantics, no amsgrad) + GradScaler.update + the step's bookkeeping, ONE launch (round 3: three; torch: a dozen).
//   phase 1  every workgroup forms the squared norm of the WHOLE unscaled gradient and its non-finite count by itself, in the same
//            fixed order (thread t owns the float4s t, t + 1024, ...; wave butterfly; 16 wave sums added in order): no float atomics,
//            the clip coefficient -- and with it every weight -- is bit-identical in every workgroup, run to run and rank to rank
//            (tests: test_training_is_bit_reproducible, test_two_ranks_on_the_gpu_stay_bit_identical).  0.5 MB from L2 per
//            workgroup, 31 independent 16-byte loads per thread: cheaper than the launch it replaces.
//   phase 2  every workgroup reads what the commit will overwrite (loss scale, learning rate, step count), waits for those loads
//            and draws a ticket (ONE agent-scope atomic add per workgroup on work[0]).
//   phase 3  Adam on the workgroup's own slice; the fp16 working copy and the fragment-major forward / backward copies the MFMA
//            policy kernels read (maps of bez_ppo_scatter2_f16) are written in the same pass.
//   phase 4  the workgroup that drew the LAST ticket commits: by then every other workgroup has read the old scale / lr / steps, so the
//            new ones cannot be seen too early -- step counters, loss-scale schedule, the tail sums (epoch KL / loss accumulators),
//            rl_games' per-step learning-rate rule; it also resets the ticket counter.  No fence: nothing a workgroup WRITES is read
//            by another workgroup of this launch.
//   extra    workgroup 0 applies the NEXT minibatch's observation moments to the input normaliser (rms_apply), which nothing in this
//            launch reads: one launch less in front of the next forward pass.
constexpr int ADAM_TB = 1024;
struct AdamTail { float* dst[4]; const float* src[4]; float scale[4]; int n; float* lr; const float* kl; float kl_thr, min_lr, max_lr; };
// data parallel: per-workgroup (sum g^2, non-finite count) of the all-reduced, still scaled gradient -- the shares bez_ppo_grad_reduce_all leaves
// on one rank.  Workgroup b owns the float4s b * 1024 + t (+ a tail workgroup for n % 4); wave butterfly, 16 wave sums added in order.
__global__ __launch_bounds__(1024) void grad_norm_parts_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ parts) {
  __shared__ float red[2][16];
  const int tid = threadIdx.x;
  const int64_t n4 = n >> 2, i = (int64_t)blockIdx.x * 1024 + tid;
  float s2 = 0.f, bad = 0.f;
  if (i < n4) {
    const float4 x = reinterpret_cast<const float4*>(g)[i];
    bad = (fabsf(x.x) <= 3.4028234e38f ? 0.f : 1.f) + (fabsf(x.y) <= 3.4028234e38f ? 0.f : 1.f) + (fabsf(x.z) <= 3.4028234e38f ? 0.f : 1.f) +
          (fabsf(x.w) <= 3.4028234e38f ? 0.f : 1.f);
    s2 = fmaf(x.x, x.x, s2); s2 = fmaf(x.y, x.y, s2); s2 = fmaf(x.z, x.z, s2); s2 = fmaf(x.w, x.w, s2);
  } else if (i - n4 < (n & 3)) {      // the last n % 4 elements, one thread each
    const float a = g[(n4 << 2) + (i - n4)];
    bad = fabsf(a) <= 3.4028234e38f ? 0.f : 1.f;
    s2 = a * a;
  }
  s2 = wave_sum(s2); bad = wave_sum(bad);
  if ((tid & 63) == 0) { red[0][tid >> 6] = s2; red[1][tid >> 6] = bad; }
  __syncthreads();
  if (tid == 0) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) { a += red[0][w]; b += red[1][w]; }
    parts[2 * blockIdx.x] = a; parts[2 * blockIdx.x + 1] = b;
  }
}

struct AdamExtra {
  float grad_div;                                                       // the buffer holds the sum over this many ranks (0 / 1: it is the gradient)
  float* gridnorm;                                                      // data parallel: 2 * 256 share slots + the arrival counter of the in-launch norm (NULL: off)
  const float* normpart; int nnormpart;                                 // per-block (sum g^2, non-finite count) of the scaled gradient (NULL: phase 1 reads the gradient)
  const int32_t* map_a; const int32_t* map_b; __half* packed;          // fragment-major weight copies (NULL: none)
  const double* rms_mom; int rms_d; double* rms_mean; double* rms_var; double* rms_count;  // next normaliser update (NULL: none)
};
__global__ __launch_bounds__(ADAM_TB) void adam_fused_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                            int64_t n, float* __restrict__ steps, int nsteps, float* __restrict__ lr, float beta1, float beta2,
                                                            float eps, float weight_decay, float max_norm, float* __restrict__ scale,
                                                            int32_t* __restrict__ growth_tracker, float growth_factor, float backoff_factor,
                                                            int32_t growth_interval, unsigned int* __restrict__ ticket, __half* __restrict__ p16,
                                                            AdamTail tail, AdamExtra ex) {
  __shared__ float red[2][ADAM_TB / 64];
  __shared__ unsigned int my_ticket;
  const int tid = threadIdx.x;
  // ---- phase 2 first in program order (the loads are in flight while phase 1 runs)
  const float scale_v = scale ? scale[0] : 1.0f;
  const float lr_v = lr[0];
  const float step_v = steps[0];
  // ... and this thread's slice of the update (at most ADAM_PER elements: gradient, parameter, moments, packed-copy slots): nothing of
  // it depends on the norm, so the loads are issued before the reduction and overlap it
  constexpr int ADAM_PER = 2;
  float gq[ADAM_PER], pq[ADAM_PER], mq[ADAM_PER], vq[ADAM_PER];
  int32_t ia[ADAM_PER], ib[ADAM_PER];
  const int64_t stride = (int64_t)gridDim.x * ADAM_TB, first = (int64_t)blockIdx.x * ADAM_TB + tid;
#pragma unroll
  for (int k = 0; k < ADAM_PER; ++k) {
    const int64_t i = first + k * stride;
    const bool in = i < n;
    gq[k] = in ? g[i] : 0.f; pq[k] = in ? p[i] : 0.f; mq[k] = in ? m[i] : 0.f; vq[k] = in ? v[i] : 0.f;
    ia[k] = (in && ex.packed) ? ex.map_a[i] : -1; ib[k] = (in && ex.packed) ? ex.map_b[i] : -1;
  }
  // ---- phase 1
  const float inv = 1.0f / (ex.grad_div > 1.0f ? scale_v * ex.grad_div : scale_v);
  float s2 = 0.f, bad = 0.f;
  if (ex.normpart) {   // the producer of the gradient (bez_ppo_grad_reduce_all) left per-block shares of sum g^2 (scaled) and of the non-finite count
    for (int i = tid; i < ex.nnormpart; i += ADAM_TB) { const float2 q = reinterpret_cast<const float2*>(ex.normpart)[i]; s2 += q.x; bad += q.y; }
    s2 *= inv * inv;
    if (!(fabsf(s2) <= 3.4028234e38f)) bad += 1.f;   // (a share that overflowed or holds a NaN)
  } else if (ex.gridnorm) {
    // data parallel, no shares from the gradient's producer (an all-reduce has replaced the gradient): every workgroup sums ITS OWN slice --
    // the elements it is about to update, already in registers --, publishes the pair, and all workgroups meet at a counter before each adds
    // the pairs in the same fixed order.  The grid is <= 256 workgroups of 1024 threads: co-resident on 256 CUs, so the wait cannot
    // starve a workgroup that has not started.  Slots and counter are exchanged with agent-scope atomics only (no cache-wide fence).
    float q2 = 0.f, qb = 0.f;
#pragma unroll
    for (int k = 0; k < ADAM_PER; ++k) {
      const int64_t i = first + k * stride;
      if (i < n) { const float a = gq[k] * inv; qb += fabsf(a) <= 3.4028234e38f ? 0.f : 1.f; q2 = fmaf(a, a, q2); }
    }
    for (int64_t i = first + ADAM_PER * stride; i < n; i += stride) { const float a = g[i] * inv; qb += fabsf(a) <= 3.4028234e38f ? 0.f : 1.f; q2 = fmaf(a, a, q2); }
    q2 = wave_sum(q2); qb = wave_sum(qb);
    if ((tid & 63) == 0) { red[0][tid >> 6] = q2; red[1][tid >> 6] = qb; }
    __syncthreads();
    unsigned int* arrive = reinterpret_cast<unsigned int*>(ex.gridnorm + 512);
    if (tid == 0) {
      float a2 = 0.f, ab = 0.f;
#pragma unroll
      for (int w = 0; w < ADAM_TB / 64; ++w) { a2 += red[0][w]; ab += red[1][w]; }
      __hip_atomic_store(ex.gridnorm + 2 * blockIdx.x, a2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ex.gridnorm + 2 * blockIdx.x + 1, ab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(arrive, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();   // (also: red[] is free again)
    for (int i = tid; i < (int)gridDim.x; i += ADAM_TB) {
      s2 += __hip_atomic_load(ex.gridnorm + 2 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      bad += __hip_atomic_load(ex.gridnorm + 2 * i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!(fabsf(s2) <= 3.4028234e38f)) bad += 1.f;
  } else {
    const int64_t n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
#pragma unroll 8
    for (int64_t i = tid; i < n4; i += ADAM_TB) {
      const float4 x = g4[i];
      const float a = x.x * inv, b = x.y * inv, c = x.z * inv, d = x.w * inv;
      bad += (fabsf(a) <= 3.4028234e38f ? 0.f : 1.f) + (fabsf(b) <= 3.4028234e38f ? 0.f : 1.f) + (fabsf(c) <= 3.4028234e38f ? 0.f : 1.f) +
             (fabsf(d) <= 3.4028234e38f ? 0.f : 1.f);
      s2 = fmaf(a, a, s2); s2 = fmaf(b, b, s2); s2 = fmaf(c, c, s2); s2 = fmaf(d, d, s2);
    }
    if (tid < (int)(n & 3)) {  // the last n % 4 elements
      const float a = g[(n4 << 2) + tid] * inv;
      bad += fabsf(a) <= 3.4028234e38f ? 0.f : 1.f;
      s2 = fmaf(a, a, s2);
    }
  }
  s2 = wave_sum(s2); bad = wave_sum(bad);
  if ((tid & 63) == 0) { red[0][tid >> 6] = s2; red[1][tid >> 6] = bad; }
  // every wave holds scale / lr / step count in registers BEFORE this barrier (the asm makes them operands here), so when wave 0 draws
  // the workgroup's ticket below, no wave of this workgroup still has one of those loads outstanding
  asm volatile("" :: "v"(scale_v), "v"(lr_v), "v"(step_v));
  __syncthreads();
  float norm2 = 0.f, nbad = 0.f;
#pragma unroll
  for (int w = 0; w < ADAM_TB / 64; ++w) { norm2 += red[0][w]; nbad += red[1][w]; }
  const bool skip = scale && nbad > 0.f;   // GradScaler.step: a non-finite gradient skips the step
  // ---- phase 3
  if (!skip) {
    const float coef = max_norm > 0.f ? fminf(max_norm / (sqrtf(norm2) + 1e-6f), 1.0f) : 1.0f;
    const float t = step_v + 1.0f;
    const float bc1 = 1.0f - powf(beta1, t), bc2 = 1.0f - powf(beta2, t);
    const float rs2 = 1.0f / sqrtf(bc2), step_size = lr_v / bc1;
#pragma unroll
    for (int k = 0; k < ADAM_PER; ++k) {
      const int64_t i = first + k * stride;
      if (i < n) {
        float x = gq[k] * inv * coef;
        float w = pq[k];
        if (weight_decay != 0.f) x = fmaf(weight_decay, w, x);
        const float mi = fmaf(beta1, mq[k], (1.0f - beta1) * x);
        const float vi = fmaf(beta2, vq[k], (1.0f - beta2) * x * x);
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) * rs2 + eps;
        w -= step_size * (mi / denom);
        p[i] = w;
        if (p16) {
          const __half h = __float2half(w);
          p16[i] = h;
          if (ia[k] >= 0) ex.packed[ia[k]] = h;
          if (ib[k] >= 0) ex.packed[ib[k]] = h;
        }
      }
    }
    for (int64_t i = first + ADAM_PER * stride; i < n; i += stride) {   // (only when the grid was capped: n > 256 x 1024 x ADAM_PER)
      float x = g[i] * inv * coef;
      float w = p[i];
      if (weight_decay != 0.f) x = fmaf(weight_decay, w, x);
      const float mi = fmaf(beta1, m[i], (1.0f - beta1) * x);
      const float vi = fmaf(beta2, v[i], (1.0f - beta2) * x * x);
      m[i] = mi; v[i] = vi;
      const float denom = sqrtf(vi) * rs2 + eps;
      w -= step_size * (mi / denom);
      p[i] = w;
      if (p16) {
        const __half h = __float2half(w);
        p16[i] = h;
        if (ex.packed) { const int32_t a = ex.map_a[i], b = ex.map_b[i]; if (a >= 0) ex.packed[a] = h; if (b >= 0) ex.packed[b] = h; }
      }
    }
  }
  // the ticket: drawn at the END of the workgroup's work -- all that matters is that its reads of scale / lr / steps have returned
  // (they were consumed above); the add's round trip then delays nobody but the committing workgroup
  if (tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    my_ticket = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // ---- extra: the next minibatch's moments into the input normaliser (parallel-variance update, as rms_apply_kernel)
  if (blockIdx.x == 0 && ex.rms_mom) {
    const int D = ex.rms_d;
    const double nn = ex.rms_mom[2 * D], cnt = ex.rms_count[0], tot = cnt + nn;
    if (tid < D) {
      const double b_mean = ex.rms_mom[tid] / nn;
      double b_var = ex.rms_mom[D + tid] / nn - b_mean * b_mean;
      if (b_var < 0.0) b_var = 0.0;
      b_var *= nn / (nn - 1.0 > 1.0 ? nn - 1.0 : 1.0);
      const double delta = b_mean - ex.rms_mean[tid];
      const double m2 = ex.rms_var[tid] * cnt + b_var * nn + delta * delta * cnt * nn / tot;
      ex.rms_mean[tid] += delta * nn / tot;
      ex.rms_var[tid] = m2 / tot;
    }
    __syncthreads();   // (uniform: the whole workgroup is in this branch) count is read above by every thread before it moves
    if (tid == 0) ex.rms_count[0] = tot;
  }
  // ---- phase 4
  __syncthreads();
  if (my_ticket != gridDim.x - 1) return;
  if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // (every workgroup drew its ticket behind the norm's meeting point: nobody is still waiting on the arrival counter)
  if (tid == 0 && ex.gridnorm) __hip_atomic_store(reinterpret_cast<unsigned int*>(ex.gridnorm + 512), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid < tail.n) *tail.dst[tid] += *tail.src[tid] * tail.scale[tid];
  if (tid == 0 && tail.kl) {   // rl_games AdaptiveScheduler.update ('legacy' schedule: after every minibatch step, for the NEXT one)
    float nv = lr_v;
    const float k = tail.kl[0];
    if (k > 2.0f * tail.kl_thr) nv = fmaxf(nv / 1.5f, tail.min_lr);
    if (k < 0.5f * tail.kl_thr) nv = fminf(nv * 1.5f, tail.max_lr);
    lr[0] = nv;
  }
  if (tid < nsteps && !skip) steps[tid] = step_v + 1.0f;
  if (tid == 0 && scale) {
    if (skip) { scale[0] = scale_v * backoff_factor; growth_tracker[0] = 0; }
    else {
      const int32_t t = growth_tracker[0] + 1;
      if (t == growth_interval) { scale[0] = scale_v * growth_factor; growth_tracker[0] = 0; }
      else growth_tracker[0] = t;
    }
  }
}

int launch_ok() { return hipGetLastError() == hipSuccess ? 0 : -2; }
unsigned nblk(int64_t n) { return (unsigned)((n + PPO_TB - 1) / PPO_TB); }

}  // namespace

extern "C" {

int32_t bez_ppo_abi_version(void) { return BEZ_PPO_ABI_VERSION; }

int bez_ppo_rms_moments(const float* x_dev, int64_t rows, int32_t cols, double* moments_dev, double* scratch_dev, void* stream) {
  if (!x_dev || !moments_dev || rows <= 0 || cols <= 0 || cols > 64) return -1;
  if (!scratch_dev) (void)hipMemsetAsync(moments_dev, 0, (size_t)(2 * cols + 1) * sizeof(double), (hipStream_t)stream);  // (the fixed-order path overwrites)
  unsigned g = (unsigned)((rows + 127) / 128);
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(rms_moments_kernel, dim3(g), dim3(PPO_TB), 0, (hipStream_t)stream, x_dev, rows, (int)cols, moments_dev, scratch_dev);
  if (scratch_dev) hipLaunchKernelGGL(rms_reduce_kernel, dim3((unsigned)((2 * cols + 63) / 64)), dim3(1024), 0, (hipStream_t)stream, (const double*)scratch_dev, (int)cols, g, moments_dev);
  return launch_ok();
}
int bez_ppo_rms_apply(const double* moments_dev, int32_t cols, double* mean_dev, double* var_dev, double* count_dev, void* stream) {
  if (!moments_dev || !mean_dev || !var_dev || !count_dev || cols <= 0 || cols > 64) return -1;
  hipLaunchKernelGGL(rms_apply_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, moments_dev, (int)cols, mean_dev, var_dev, count_dev);
  return launch_ok();
}
int bez_ppo_rms_normalize(const float* x_dev, int64_t rows, int32_t cols, const double* mean_dev, const double* var_dev, float eps, void* y_dev,
                          int32_t out_f16, void* stream) {
  if (!x_dev || !mean_dev || !var_dev || !y_dev || rows <= 0 || cols <= 0) return -1;
  const int64_t total = rows * cols;
  if (out_f16) hipLaunchKernelGGL(rms_normalize_kernel<_Float16>, dim3(nblk(total)), dim3(PPO_TB), 0, (hipStream_t)stream, x_dev, total, (int)cols, mean_dev, var_dev, eps, (_Float16*)y_dev);
  else hipLaunchKernelGGL(rms_normalize_kernel<float>, dim3(nblk(total)), dim3(PPO_TB), 0, (hipStream_t)stream, x_dev, total, (int)cols, mean_dev, var_dev, eps, (float*)y_dev);
  return launch_ok();
}
int bez_ppo_sample(const float* mu_dev, const float* logstd_dev, const float* noise_dev, int64_t n, int32_t num_actions, float* actions_dev,
                   float* env_actions_dev, float* neglogp_dev, float* sigma_dev, void* stream) {
  if (!mu_dev || !logstd_dev || !noise_dev || !actions_dev || !env_actions_dev || !neglogp_dev || !sigma_dev || n <= 0 || num_actions <= 0) return -1;
  hipLaunchKernelGGL(ppo_sample_kernel, dim3(nblk(n)), dim3(PPO_TB), 0, (hipStream_t)stream, mu_dev, logstd_dev, noise_dev, n, (int)num_actions, actions_dev,
                     env_actions_dev, neglogp_dev, sigma_dev);
  return launch_ok();
}
int bez_ppo_rollout_pre(const void* mu_dev, const void* value_dev, int32_t inputs_f16, const float* logstd_dev, const float* noise_dev, const float* obs_dev,
                        const float* dones_dev, const double* value_mean_dev, const double* value_var_dev, float value_eps, int64_t n, int32_t num_actions,
                        int32_t num_obs, float* mb_obs_dev, float* mb_dones_dev, float* mb_mu_dev, float* mb_val_dev, float* actions_dev,
                        float* env_actions_dev, float* neglogp_dev, float* sigma_dev, void* stream) {
  if (!mu_dev || !value_dev || !logstd_dev || !noise_dev || !obs_dev || !dones_dev || !mb_obs_dev || !mb_dones_dev || !mb_mu_dev || !mb_val_dev ||
      !actions_dev || !env_actions_dev || !neglogp_dev || !sigma_dev || n <= 0 || num_actions <= 0 || num_obs <= 0 || (value_mean_dev && !value_var_dev)) return -1;
  int64_t work = n * num_obs;
  unsigned g = nblk(work > n ? work : n);
  if (g > 1024) g = 1024;
  if (g < nblk(n)) g = nblk(n);
  if (inputs_f16)
    hipLaunchKernelGGL(ppo_rollout_pre_kernel<__half>, dim3(g), dim3(PPO_TB), 0, (hipStream_t)stream, (const __half*)mu_dev, (const __half*)value_dev, logstd_dev,
                       noise_dev, obs_dev, dones_dev, value_mean_dev, value_var_dev, value_eps, n, (int)num_actions, (int)num_obs, mb_obs_dev, mb_dones_dev,
                       mb_mu_dev, mb_val_dev, actions_dev, env_actions_dev, neglogp_dev, sigma_dev);
  else
    hipLaunchKernelGGL(ppo_rollout_pre_kernel<float>, dim3(g), dim3(PPO_TB), 0, (hipStream_t)stream, (const float*)mu_dev, (const float*)value_dev, logstd_dev,
                       noise_dev, obs_dev, dones_dev, value_mean_dev, value_var_dev, value_eps, n, (int)num_actions, (int)num_obs, mb_obs_dev, mb_dones_dev,
                       mb_mu_dev, mb_val_dev, actions_dev, env_actions_dev, neglogp_dev, sigma_dev);
  return launch_ok();
}
int bez_ppo_rollout_post(const float* rew_dev, const int64_t* dones_dev, const int64_t* timeouts_dev, const float* values_dev, int64_t n, float reward_scale,
                         float gamma, int32_t value_bootstrap, float* shaped_dev, float* dones_f_dev, float* cur_rew_dev, float* cur_len_dev,
                         double* ep_stats_dev, void* stream) {
  if (!rew_dev || !dones_dev || !timeouts_dev || !values_dev || !shaped_dev || !dones_f_dev || !cur_rew_dev || !cur_len_dev || !ep_stats_dev || n <= 0) return -1;
  hipLaunchKernelGGL(ppo_rollout_post_kernel, dim3(nblk(n)), dim3(PPO_TB), 0, (hipStream_t)stream, rew_dev, dones_dev, timeouts_dev, values_dev, n, reward_scale,
                     gamma, (int)value_bootstrap, shaped_dev, dones_f_dev, cur_rew_dev, cur_len_dev, ep_stats_dev, (double*)nullptr, 0);
  return launch_ok();
}
int bez_ppo_rollout_post_fold(const float* rew_dev, const int64_t* dones_dev, const int64_t* timeouts_dev, const float* values_dev, int64_t n, float reward_scale,
                              float gamma, int32_t value_bootstrap, float* shaped_dev, float* dones_f_dev, float* cur_rew_dev, float* cur_len_dev,
                              double* ep_stats_dev, double* ep_parts_dev, int32_t nslots, void* stream) {
  if (!rew_dev || !dones_dev || !timeouts_dev || !values_dev || !shaped_dev || !dones_f_dev || !cur_rew_dev || !cur_len_dev || !ep_stats_dev || n <= 0 ||
      !ep_parts_dev || nslots <= 0) return -1;
  hipLaunchKernelGGL(ppo_rollout_post_kernel, dim3(nblk(n) + 1), dim3(PPO_TB), 0, (hipStream_t)stream, rew_dev, dones_dev, timeouts_dev, values_dev, n, reward_scale,
                     gamma, (int)value_bootstrap, shaped_dev, dones_f_dev, cur_rew_dev, cur_len_dev, ep_stats_dev, ep_parts_dev, (int)nslots);
  return launch_ok();
}
int bez_ppo_loss(const float* mu_dev, const float* logstd_dev, const float* value_dev, const float* actions_dev, const float* old_logp_dev,
                 const float* adv_dev, const float* old_value_dev, const float* returns_dev, const float* old_mu_dev, const float* old_sigma_dev,
                 int64_t batch, int32_t num_actions, float e_clip, float critic_coef, float entropy_coef, float bounds_coef, int32_t clip_value,
                 const float* loss_scale_dev, float* grad_mu_dev, float* grad_value_dev, float* grad_logstd_dev, float* stats_dev,
                 float* scratch_dev, void* stream) {
  if (!mu_dev || !logstd_dev || !value_dev || !actions_dev || !old_logp_dev || !adv_dev || !old_value_dev || !returns_dev || !old_mu_dev ||
      !old_sigma_dev || !grad_mu_dev || !grad_value_dev || !grad_logstd_dev || !stats_dev || batch <= 0 || num_actions <= 0 || num_actions > 32) return -1;
  const bool defer = (clip_value & 16) != 0;  // bit 4: per-workgroup partials only (needs scratch); bez_ppo_grad_reduce_all writes grad_logstd / stats
  if (defer && !scratch_dev) return -1;
  if (!defer && !(clip_value & 2)) (void)hipMemsetAsync(grad_logstd_dev, 0, (size_t)num_actions * sizeof(float), (hipStream_t)stream);  // bit 1: accumulate
  if (!defer && !(clip_value & 4)) (void)hipMemsetAsync(stats_dev, 0, 5 * sizeof(float), (hipStream_t)stream);                    // bit 2: the caller zeroed stats
  bez_loss::LossArgs LA{mu_dev, logstd_dev, value_dev, actions_dev, old_logp_dev, adv_dev, old_value_dev, returns_dev, old_mu_dev, old_sigma_dev, batch, e_clip,
                           critic_coef, entropy_coef, bounds_coef, (int)(clip_value & 9), loss_scale_dev, grad_mu_dev, grad_value_dev, grad_logstd_dev, stats_dev, scratch_dev};
#define BEZ_PPO_LOSS(AA) hipLaunchKernelGGL(ppo_loss_kernel<AA>, dim3((unsigned)((batch + LOSS_TB - 1) / LOSS_TB)), dim3(LOSS_TB * LOSS_CW), 0, (hipStream_t)stream, LA)
  switch (num_actions) {  // the action width is a compile-time constant of the kernel (register arrays, unrolled loops): bez has 18
    case 1: BEZ_PPO_LOSS(1); break;
    case 2: BEZ_PPO_LOSS(2); break;
    case 3: BEZ_PPO_LOSS(3); break;
    case 4: BEZ_PPO_LOSS(4); break;
    case 5: BEZ_PPO_LOSS(5); break;
    case 6: BEZ_PPO_LOSS(6); break;
    case 7: BEZ_PPO_LOSS(7); break;
    case 8: BEZ_PPO_LOSS(8); break;
    case 9: BEZ_PPO_LOSS(9); break;
    case 10: BEZ_PPO_LOSS(10); break;
    case 11: BEZ_PPO_LOSS(11); break;
    case 12: BEZ_PPO_LOSS(12); break;
    case 13: BEZ_PPO_LOSS(13); break;
    case 14: BEZ_PPO_LOSS(14); break;
    case 15: BEZ_PPO_LOSS(15); break;
    case 16: BEZ_PPO_LOSS(16); break;
    case 17: BEZ_PPO_LOSS(17); break;
    case 18: BEZ_PPO_LOSS(18); break;
    case 19: BEZ_PPO_LOSS(19); break;
    case 20: BEZ_PPO_LOSS(20); break;
    case 21: BEZ_PPO_LOSS(21); break;
    case 22: BEZ_PPO_LOSS(22); break;
    case 23: BEZ_PPO_LOSS(23); break;
    case 24: BEZ_PPO_LOSS(24); break;
    case 25: BEZ_PPO_LOSS(25); break;
    case 26: BEZ_PPO_LOSS(26); break;
    case 27: BEZ_PPO_LOSS(27); break;
    case 28: BEZ_PPO_LOSS(28); break;
    case 29: BEZ_PPO_LOSS(29); break;
    case 30: BEZ_PPO_LOSS(30); break;
    case 31: BEZ_PPO_LOSS(31); break;
    case 32: BEZ_PPO_LOSS(32); break;
    default: return -1;
  }
#undef BEZ_PPO_LOSS
  if (scratch_dev && !defer)
    hipLaunchKernelGGL(ppo_loss_reduce_kernel, dim3((unsigned)num_actions + 5), dim3(64), 0, (hipStream_t)stream, (const float*)scratch_dev, (int)num_actions,
                       (unsigned int)((batch + LOSS_TB - 1) / LOSS_TB), grad_logstd_dev, stats_dev);
  return launch_ok();
}

int bez_ppo_wgrad_sum(const void* partials_f16_dev, int32_t splits, int64_t n, float* out_dev, int32_t accumulate, void* stream) {
  if (!partials_f16_dev || !out_dev || splits <= 0 || n <= 0) return -1;
  hipLaunchKernelGGL(wgrad_sum_kernel, dim3((unsigned)((n + 127) / 128)), dim3(PPO_TB), 0, (hipStream_t)stream, (const __half*)partials_f16_dev,
                     (int)splits, n, out_dev, (int)accumulate);
  return launch_ok();
}

int bez_ppo_colsum_f16(const void* y_f16_dev, int64_t rows, int32_t cols, float* out_dev, int32_t accumulate, void* stream) {
  if (!y_f16_dev || !out_dev || rows <= 0 || cols <= 0) return -1;
  if (!accumulate) (void)hipMemsetAsync(out_dev, 0, (size_t)cols * sizeof(float), (hipStream_t)stream);
  const int64_t rpb = 256;
  hipLaunchKernelGGL(colsum_half_kernel, dim3((unsigned)((cols + 127) / 128), (unsigned)((rows + rpb - 1) / rpb)), dim3(PPO_TB), 0, (hipStream_t)stream,
                     (const __half*)y_f16_dev, rows, (int)cols, rpb, out_dev);
  return launch_ok();
}

int bez_ppo_elu_bwd_colsum_f16(const void* gy_f16_dev, const void* y_f16_dev, void* gz_f16_dev, int64_t rows, int32_t cols, float* bias_grad_dev,
                               int32_t accumulate, void* stream) {
  if (!gy_f16_dev || !y_f16_dev || !gz_f16_dev || !bias_grad_dev || rows <= 0 || cols <= 0) return -1;
  if (!accumulate) (void)hipMemsetAsync(bias_grad_dev, 0, (size_t)cols * sizeof(float), (hipStream_t)stream);
  const int64_t rpb = 256;
  hipLaunchKernelGGL(elu_bwd_colsum_kernel, dim3((unsigned)((cols + 127) / 128), (unsigned)((rows + rpb - 1) / rpb)), dim3(ELU_RL * 64), 0, (hipStream_t)stream,
                     (const __half*)gy_f16_dev, (const __half*)y_f16_dev, (__half*)gz_f16_dev, rows, (int)cols, rpb, bias_grad_dev);
  return launch_ok();
}

int bez_ppo_adaptive_lr(float* lr_dev, const float* kl_dev, float kl_threshold, float min_lr, float max_lr, void* stream) {
  if (!lr_dev || !kl_dev) return -1;
  hipLaunchKernelGGL(adaptive_lr_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, lr_dev, kl_dev, kl_threshold, min_lr, max_lr);
  return launch_ok();
}

int bez_ppo_gae(const float* rewards_dev, const float* values_dev, const float* mb_dones_dev, const float* dones_dev, const float* last_values_dev,
                int32_t horizon, int64_t num_envs, float gamma, float tau, float* advantages_dev, float* returns_dev, const double* value_mean_dev,
                const double* value_var_dev, float value_eps, void* stream) {
  if (!rewards_dev || !values_dev || !mb_dones_dev || !dones_dev || !last_values_dev || !advantages_dev || horizon <= 0 || num_envs <= 0 ||
      (value_mean_dev && !value_var_dev)) return -1;
  hipLaunchKernelGGL(gae_kernel, dim3(nblk(num_envs)), dim3(PPO_TB), 0, (hipStream_t)stream, rewards_dev, values_dev, mb_dones_dev, dones_dev, last_values_dev,
                     (int)horizon, num_envs, gamma, tau, advantages_dev, returns_dev, value_mean_dev, value_var_dev, value_eps);
  return launch_ok();
}

int bez_ppo_dataset_prep_staged(int32_t stages, const float* obs_dev, int64_t minibatch_rows, int32_t num_minibatches, int32_t num_obs,
                                double* obs_moments_dev, const float* values_dev, const float* returns_dev, int32_t horizon, int64_t num_envs,
                                double* value_mean_dev, double* value_var_dev, double* value_count_dev, float value_eps, double* value_moments_dev,
                                double* return_moments_dev, float* old_values_dev, float* ds_returns_dev, float* advantages_dev,
                                int32_t normalize_advantage, double* adv_sums_dev, double* scratch_dev, int64_t scratch_doubles, void* stream) {
  const int64_t total = (int64_t)horizon * num_envs;
  if (!values_dev || !returns_dev || horizon <= 0 || num_envs <= 0 || !value_moments_dev || !return_moments_dev || !old_values_dev || !ds_returns_dev ||
      !advantages_dev || !scratch_dev || (value_mean_dev && (!value_var_dev || !value_count_dev)) || num_minibatches < 0 || num_minibatches > PREP_MAXT - 2 ||
      (num_minibatches > 0 && (!obs_dev || !obs_moments_dev || minibatch_rows <= 0 || num_obs <= 0 || num_obs > 64))) return -1;
  if (stages <= 0 || stages > 7 || ((((stages & 2) != 0) != ((stages & 4) != 0)) && !adv_sums_dev)) return -1;   // stages 2 and 4 apart: via the buffer the second collective reduces
  if (total % 64 != 0) return -3;   // the scalar tasks are read as (total / 64, 64): the caller keeps its separate launches
  hipStream_t st = (hipStream_t)stream;
  PrepTasks T;
  T.n = num_minibatches + 2;
  for (int i = 0; i < num_minibatches; ++i)
    T.t[i] = PrepTask{obs_dev + (size_t)i * minibatch_rows * num_obs, (long long)minibatch_rows, (int)num_obs, obs_moments_dev + (size_t)i * (2 * num_obs + 1)};
  T.t[num_minibatches] = PrepTask{values_dev, (long long)(total / 64), 64, value_moments_dev};
  T.t[num_minibatches + 1] = PrepTask{returns_dev, (long long)(total / 64), 64, return_moments_dev};
  for (int i = T.n; i < PREP_MAXT; ++i) T.t[i] = PrepTask{nullptr, 0, 0, nullptr};
  int64_t maxrows = total / 64;
  if (num_minibatches > 0 && minibatch_rows > maxrows) maxrows = minibatch_rows;
  unsigned g = (unsigned)((maxrows + 127) / 128);
  if (g > 256) g = 256;
  const unsigned nvb = nblk(total);
  if ((int64_t)T.n * g * 128 + 2 * (int64_t)nvb > scratch_doubles) return -1;
  double* partial = scratch_dev + (size_t)T.n * g * 128;
  if (stages & 1) {
    hipLaunchKernelGGL(prep_moments_kernel, dim3(g, (unsigned)T.n), dim3(PPO_TB), 0, st, T, scratch_dev);
    hipLaunchKernelGGL(prep_reduce_kernel, dim3((unsigned)T.n), dim3(1024), 0, st, T, (const double*)scratch_dev, (int)g, (int)num_minibatches);
  }
  if (stages & 2) {
    hipLaunchKernelGGL(prep_values_kernel, dim3(nvb), dim3(PPO_TB), 0, st, values_dev, returns_dev, (int)horizon, (long long)num_envs, (const double*)value_mean_dev,
                       (const double*)value_var_dev, (const double*)value_count_dev, value_eps, (const double*)value_moments_dev, (const double*)return_moments_dev,
                       old_values_dev, ds_returns_dev, advantages_dev, partial);
    if (adv_sums_dev) hipLaunchKernelGGL(prep_adv_sums_kernel, dim3(1), dim3(PPO_TB), 0, st, (const double*)partial, (int)nvb, (long long)total, adv_sums_dev);
  }
  if (stages & 4)
    hipLaunchKernelGGL(prep_advantage_kernel, dim3(nvb), dim3(PPO_TB), 0, st, advantages_dev, (long long)total, (const double*)partial, (int)nvb,
                       (int)(normalize_advantage != 0), value_mean_dev, value_var_dev, value_count_dev, (const double*)value_moments_dev, (const double*)return_moments_dev,
                       (const double*)adv_sums_dev);
  return launch_ok();
}

int bez_ppo_dataset_prep(const float* obs_dev, int64_t minibatch_rows, int32_t num_minibatches, int32_t num_obs, double* obs_moments_dev,
                         const float* values_dev, const float* returns_dev, int32_t horizon, int64_t num_envs, double* value_mean_dev, double* value_var_dev,
                         double* value_count_dev, float value_eps, double* value_moments_dev, double* return_moments_dev, float* old_values_dev,
                         float* ds_returns_dev, float* advantages_dev, int32_t normalize_advantage, double* scratch_dev, int64_t scratch_doubles, void* stream) {
  return bez_ppo_dataset_prep_staged(7, obs_dev, minibatch_rows, num_minibatches, num_obs, obs_moments_dev, values_dev, returns_dev, horizon, num_envs, value_mean_dev,
                                     value_var_dev, value_count_dev, value_eps, value_moments_dev, return_moments_dev, old_values_dev, ds_returns_dev, advantages_dev,
                                     normalize_advantage, nullptr, scratch_dev, scratch_doubles, stream);
}

int bez_ppo_head_grads_f16(const float* grad_mu_dev, const float* grad_value_dev, int64_t rows, int32_t num_actions, void* grad_mu_f16_dev,
                           void* grad_value_f16_dev, float* mu_bias_grad_dev, float* value_bias_grad_dev, void* stream) {
  if (!grad_mu_dev || !grad_value_dev || !grad_mu_f16_dev || !grad_value_f16_dev || !mu_bias_grad_dev || !value_bias_grad_dev || rows <= 0 ||
      num_actions <= 0 || num_actions > PPO_TB) return -1;
  const int64_t rpb = 256;
  hipLaunchKernelGGL(head_grads_kernel, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(PPO_TB), 0, (hipStream_t)stream, grad_mu_dev, grad_value_dev, rows,
                     (int)num_actions, (__half*)grad_mu_f16_dev, (__half*)grad_value_f16_dev, mu_bias_grad_dev, value_bias_grad_dev, rpb);
  return launch_ok();
}

int bez_ppo_grad_norm_parts(const float* grads_dev, int64_t n, float* parts_dev, int32_t parts, void* stream) {
  if (!grads_dev || !parts_dev || n <= 0 || (reinterpret_cast<uintptr_t>(grads_dev) & 15) != 0) return -1;
  const int64_t units = (n >> 2) + (n & 3);
  const int64_t g = (units + 1023) / 1024;
  if (g > parts) return -1;
  hipLaunchKernelGGL(grad_norm_parts_kernel, dim3((unsigned)g), dim3(1024), 0, (hipStream_t)stream, grads_dev, n, parts_dev);
  const int rc = launch_ok();
  return rc ? rc : (int)g;
}

static unsigned adam_grid(int64_t n) {
  unsigned g = (unsigned)((n + 2 * ADAM_TB - 1) / (2 * ADAM_TB));
  if (g < 8) g = 8;
  if (g > 256) g = 256;
  return g;
}
// workgroups of adam_fused_kernel that fit on the current device at the same time (queried once per device)
static int adam_resident_capacity(int* out) {
  static int cached[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
  if (cached[dev] == 0) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, adam_fused_kernel, ADAM_TB, 0) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
    cached[dev] = per_cu * cus > 0 ? per_cu * cus : -1;
  }
  *out = cached[dev] > 0 ? cached[dev] : 0;
  return 0;
}
int bez_ppo_adam_grid_capacity(int64_t n, int32_t* resident_workgroups, int32_t* launch_workgroups) {
  if (n <= 0 || !resident_workgroups || !launch_workgroups) return -1;
  int cap = 0;
  if (adam_resident_capacity(&cap) != 0) return -2;
  *resident_workgroups = cap; *launch_workgroups = (int32_t)adam_grid(n);
  return 0;
}
int bez_ppo_adam_step(float* params_dev, const float* grads_dev, float* exp_avg_dev, float* exp_avg_sq_dev, int64_t n, float* steps_dev,
                      int32_t nsteps, float* lr_dev, float beta1, float beta2, float eps, float weight_decay, float max_norm, float* scale_dev,
                      int32_t* growth_tracker_dev, float growth_factor, float backoff_factor, int32_t growth_interval, float* work_dev,
                      void* params_f16_dev, int32_t ntail, float* const* tail_dst_dev, const float* const* tail_src_dev, const float* tail_scale,
                      const float* adapt_kl_dev, float adapt_kl_threshold, float min_lr, float max_lr, const BezPpoAdamExtra* extra, void* stream) {
  if (!params_dev || !grads_dev || !exp_avg_dev || !exp_avg_sq_dev || !steps_dev || !lr_dev || !work_dev || n <= 0 || nsteps <= 0 || nsteps > 64 ||
      (scale_dev && !growth_tracker_dev) || ntail < 0 || ntail > 4 || (ntail > 0 && (!tail_dst_dev || !tail_src_dev || !tail_scale))) return -1;
  if ((reinterpret_cast<uintptr_t>(grads_dev) & 15) != 0) return -1;  // the norm pass reads the gradient as float4
  hipStream_t st = (hipStream_t)stream;
  AdamTail tail;
  tail.n = ntail;
  tail.lr = lr_dev; tail.kl = adapt_kl_dev; tail.kl_thr = adapt_kl_threshold; tail.min_lr = min_lr; tail.max_lr = max_lr;
  for (int i = 0; i < 4; ++i) {
    tail.dst[i] = i < ntail ? tail_dst_dev[i] : nullptr; tail.src[i] = i < ntail ? tail_src_dev[i] : nullptr; tail.scale[i] = i < ntail ? tail_scale[i] : 0.f;
    if (i < ntail && (!tail.dst[i] || !tail.src[i])) return -1;
  }
  AdamExtra ex{};
  if (extra) {
    if (extra->packed_f16_dev && (!extra->map_a_dev || !extra->map_b_dev || !params_f16_dev)) return -1;
    if (extra->rms_moments_dev && (!extra->rms_mean_dev || !extra->rms_var_dev || !extra->rms_count_dev || extra->rms_cols <= 0 || extra->rms_cols > ADAM_TB)) return -1;
    if (extra->norm_parts_dev && (extra->norm_parts <= 0 || (reinterpret_cast<uintptr_t>(extra->norm_parts_dev) & 7) != 0)) return -1;
    ex.normpart = extra->norm_parts_dev; ex.nnormpart = extra->norm_parts; ex.grad_div = extra->grad_div;
    if (extra->grid_norm_dev && (extra->norm_parts_dev || (reinterpret_cast<uintptr_t>(extra->grid_norm_dev) & 3) != 0)) return -1;
    ex.gridnorm = extra->grid_norm_dev;
    ex.map_a = extra->map_a_dev; ex.map_b = extra->map_b_dev; ex.packed = (__half*)extra->packed_f16_dev;
    ex.rms_mom = extra->rms_moments_dev; ex.rms_d = extra->rms_cols; ex.rms_mean = extra->rms_mean_dev; ex.rms_var = extra->rms_var_dev; ex.rms_count = extra->rms_count_dev;
  }
  // one slice of <= 2 elements per thread keeps the update short; at least 8 workgroups so that every XCD has one
  const unsigned g = adam_grid(n);
  if (ex.gridnorm) {   // the in-launch norm is a grid barrier: every workgroup of the launch has to be resident at once (round-5 advisor finding)
    int cap = 0;
    if (adam_resident_capacity(&cap) != 0 || (int)g > cap) return -6;
  }
  hipLaunchKernelGGL(adam_fused_kernel, dim3(g), dim3(ADAM_TB), 0, st, params_dev, grads_dev, exp_avg_dev, exp_avg_sq_dev, n, steps_dev, (int)nsteps, lr_dev,
                     beta1, beta2, eps, weight_decay, max_norm, scale_dev, growth_tracker_dev, growth_factor, backoff_factor, growth_interval,
                     reinterpret_cast<unsigned int*>(work_dev), (__half*)params_f16_dev, tail, ex);
  return launch_ok();
}

}  // extern "C"

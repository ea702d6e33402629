// bez_kernels.h -- gfx950 device code of the bez_kick hot path: one environment per lane.
//
// Replaces, for all envs in ONE launch (include/bez_sim.h: bez_sim_step):
//   vec_task.py:317 action clamp, kick_env.py:410-419 PD targets, vec_task.py:322-324 gym.simulate
//   (this build's own articulated-body step), vec_task.py:331-332 timeout, kick_env.py:426-438
//   post_physics_step incl. reset_idx (kick_env.py:779-850), compute_observations (kick_env.py:749-777)
//   and compute_bez_reward (kick_env.py:1198-1395).
//
// Layout: the simulator state is SoA, state[field * N + env] (coalesced across the 64 lanes of a wave).
// Per substep each lane runs Featherstone's ABA over the 19-link tree in world-aligned coordinates about
// the torso origin (no inter-link spatial transforms), chain by chain so that only one chain's link data
// is live in VGPRs; the per-joint quantities pass 3 needs (U/D, u/D, S, c) and the contact-point records
// are staged in LDS as lds[slot * BLOCK + lane] (bank = lane: conflict-free).
#pragma once
#include <stdint.h>

#include <utility>

#include "../../include/bez_sim.h"
#include "bez_model_gen.h"
#include "bez_shapes_gen.h"
#include "bez_spatial.h"
#include "bez_dr_noise.h"

namespace bez {

constexpr int BLOCK = 64;  // one wave per workgroup
constexpr float SELF_IMPLICIT = 2.0f;  // BEZ_SELF_IMPLICIT of the oracle (self_contact_scale)

// ---- SoA state fields (floats per env)
enum : int {
  F_ROOT_POS = 0, F_ROOT_QUAT = 3, F_ROOT_LIN = 7, F_ROOT_ANG = 10, F_Q = 13, F_QD = 31,
  F_BALL_POS = 49, F_BALL_QUAT = 52, F_BALL_LIN = 56, F_BALL_ANG = 59, F_TARGET = 62, F_PREV = 80,
  F_CF = 83 /* up to BEZ_NBE_MAX = 30 bodies x 3 (22 used without cleats) */, F_FEET = F_CF + BEZ_NBE_MAX * 3, F_GOAL = F_FEET + 8 /* bez_walk: per-env goal xy */,
  F_COUNT = F_GOAL + 2
};

// ---- LDS slots per lane
constexpr int P3_STRIDE = 20;                       // UD(6) uD(1) S(6) cb(6) per joint + the leg<->leg contacts' share of uD (1)
constexpr int LDS_P3 = 0;                           // 18 joints
constexpr int LDS_HIT = LDS_P3 + BEZ_ND * P3_STRIDE;  // ground-point records, 8 floats each
constexpr int HIT_STRIDE = 8;                       // x(3) fn0 kn ct ftx0 fty0
constexpr int LDS_STOP = LDS_HIT + BEZ_NPT * HIT_STRIDE;   // BEZ_FLAG_ANKLE_STOP: per leg 4 corner records (Ja, Jf, lam0) + the foot's normal (3)
constexpr int STOP_STRIDE = 15;
constexpr int LDS_SLOTS = LDS_STOP + 2 * STOP_STRIDE;
static_assert(LDS_SLOTS * BLOCK * 4 <= 160 * 1024, "LDS budget of the one-env-per-lane kernel");
constexpr float STOP_KN = 2.0e5f, STOP_CN = 1.0e3f;   // calf <-> foot-plate contact (oracle: m_stop_kn / m_stop_cn defaults)

// (DrState, the device-resident state of the domain randomisation: bez_dr_noise.h)
// (DrSnap, the action-noise snapshot the step kernels keep: bez_dr_noise.h)

struct Params {
  int n, substeps, max_len, use_prev, obs_only;
  float dt, h, inv_h;
  float ang_init;  // atan2 of the unit vector ball_init -> goal (kick_env.py:1238-1243): constant of the config
  float g[3];
  float kp, kd, armature, effort, vel_limit, jfric, mu, clip;
  float bez_init[7], ball_init[7], goal[2];
  float kn, cn, ct, veps, lim_k, lim_d, jf_veps, ball_damp;
  int lean;        // fused step only: skip the stores of the contact-force rows, the feet flags and prev_lin_vel (BEZ_FLAG_LEAN_STEP)
  float bkn, bcn;  // ball <-> ground / ball <-> robot spring and damper (BezSimConfig.ball_kn / ball_cn, defaulting to kn / cn)
  float self_kn, self_cn;
  float cf_w;  // weight of one substep in the net-contact-force mean (1/substeps, or 1 with BEZ_FLAG_CF_LAST_SUBSTEP)
  int task, nobs;       // BEZ_TASK_*; observation width (54 kick, 52 walk / orient)
  float goal_angle;     // bez_orient
  float goal_draw[2];   // bez_walk / bez_orient: the ONE goal every env reset by this launch receives (walk_env.py:570-575)
  const float* goal_dev; // != null: that goal lives in device memory, written by goal_draw_kernel just before this launch from a
                         // DEVICE-resident call counter -- a HIP-graph replay then draws a fresh goal per step (by-value kernel
                         // arguments are frozen at capture)
  uint32_t flags;
  uint64_t seed;
  int64_t env_off;
  float* state;
  float* obs;
  float* rew;
  int64_t* reset;
  int64_t* progress;
  int64_t* timeout;
  uint32_t* episode;
  const float* actions;
  const DrState* dr_state;   // != null with obs_noise: frame counter + noise parameters of the device-side domain randomisation
  int obs_noise;             // the POST part adds the observation noise itself (BEZ_FLAG_OBS_NOISE_IN_STEP)
  DrSnap* dr_snap;           // != null: POST leaves the action-noise snapshot for the next step's consumer
  const float* dr_friction;  // (N)      or null
  const float* dr_kp;        // (N,18)   or null
  const float* dr_kd;        // (N,18)   or null
  const float* dr_mass;      // (N,19)   or null
  const float* dr_gravity;   // (N,3)    or null
  int dr_gravity_uniform;    // every row of dr_gravity is the same (the randomisation's one draw per sim, vec_task.py:620-632): read row 0 as a scalar
  const float4* dr_pack;     // (N,18) {kp scale, kd scale, lower, upper} = dr_kp / dr_kd / dr_lower / dr_upper in one 16-byte load per joint, or null
  const float* dr_lower;     // (N,18)   or null: physical joint limits (targets keep the model's, kick_env.py:393-400)
  const float* dr_upper;     // (N,18)   or null
  float* xhit;               // != null: BEZ_FLAG_ALL_GROUND_SHAPES -- scratch for the BEZ_NXPT extra ground-point records, [(point * 8 + k) * n + env]
  unsigned long long* stamps; // diagnostic builds only (-DBEZ_WS_STAMPS): s_memtime per role / phase of workgroup 0
};
// the goal an env reset by this launch receives (bez_walk / bez_orient)
BEZ_DEV float reset_goal(const Params& P, int k) { return P.goal_dev ? P.goal_dev[k] : P.goal_draw[k]; }



// ---- model variant: CL = the cleats asset (BEZ_FLAG_CLEATS): same tree, heavier feet, 8 cleat bodies, per-cleat ground points
template <bool CL> BEZ_DEV constexpr int nb_of() { return CL ? BEZ_NB_CL : BEZ_NB; }   // robot bodies; the ball's row follows
template <bool CL> BEZ_DEV constexpr int link_body(int l) { return CL ? BEZ_LINK_BODY_CL[l] : BEZ_LINK_BODY[l]; }
template <bool CL> BEZ_DEV constexpr double link_mass(int l) { return CL ? BEZ_LINK_MASS_CL[l] : BEZ_LINK_MASS[l]; }
template <bool CL> BEZ_DEV constexpr double link_com(int l, int k) { return CL ? BEZ_LINK_COM_CL[l][k] : BEZ_LINK_COM[l][k]; }
template <bool CL> BEZ_DEV constexpr double link_inertia_c(int l, int k) { return CL ? BEZ_LINK_INERTIA_CL[l][k] : BEZ_LINK_INERTIA[l][k]; }
template <bool CL> BEZ_DEV constexpr double pt_pos(int i, int k) { return CL ? BEZ_PT_POS_CL[i][k] : BEZ_PT_POS[i][k]; }
// the same coordinate for the asset the sim was created with: the box asset (BEZ_FLAG_BOX_ASSET) moves the upper-body guard
// points only; where the two tables agree (feet) this folds to the constant, elsewhere it is one scalar select of two literals
template <bool CL> BEZ_DEV float pt_pos_of(const Params& P, int i, int k) {
  // foot / cleat points (the first 8) are the stl assets'; the upper-body points are the same with and without cleats
  const float a = CL ? (float)BEZ_PT_POS_CL[i][k] : (float)BEZ_PT_POS[i][k], b = i < 8 ? a : (float)BEZ_PT_POS_BOX[i][k];
  return a == b ? a : ((P.flags & BEZ_FLAG_BOX_ASSET) ? b : a);
}
// soccerbot_box_sensor.urdf (box asset WITH cleats) also moves one joint origin: z of link BEZ_BOXCL_LINK's joint in its parent.
// link_kinematics takes it as an argument whose default is the common constant: only CL code pays the scalar select.
constexpr float BEZ_QUIRK_RZ = (float)BEZ_LINK_XYZ[BEZ_BOXCL_LINK][2];
static_assert(BEZ_LINK_XYZ[BEZ_BOXCL_LINK][0] == 0.0 && BEZ_LINK_XYZ[BEZ_BOXCL_LINK][1] == 0.0, "the quirk joint origin is a pure z offset");
template <bool CL> BEZ_DEV float quirk_rz(uint32_t flags) {
  if constexpr (!CL) return BEZ_QUIRK_RZ;
  else return (flags & BEZ_FLAG_BOX_ASSET) ? (float)BEZ_BOXCL_LINK_Z : BEZ_QUIRK_RZ;
}
template <bool CL> BEZ_DEV constexpr int pt_body(int i) { return CL ? BEZ_PT_BODY_CL[i] : BEZ_PT_BODY[i]; }
template <bool CL> BEZ_DEV constexpr int lfoot_body() { return CL ? BEZ_LFOOT_BODY_CL : BEZ_LFOOT_BODY; }
template <bool CL> BEZ_DEV constexpr int rfoot_body() { return CL ? BEZ_RFOOT_BODY_CL : BEZ_RFOOT_BODY; }

// ---- compile-time model access
BEZ_DEV constexpr int axis_index(int l) { return (BEZ_LINK_AXIS[l] < 0 ? -BEZ_LINK_AXIS[l] : BEZ_LINK_AXIS[l]) - 1; }
BEZ_DEV constexpr float axis_sign(int l) { return BEZ_LINK_AXIS[l] < 0 ? -1.f : 1.f; }
BEZ_DEV constexpr bool link_has_box(int l) {
  for (int b = 0; b < BEZ_NBOX; ++b) if (BEZ_BOX_LINK[b] == l) return true;
  return false;
}
BEZ_DEV constexpr int link_box(int l) {
  for (int b = 0; b < BEZ_NBOX; ++b) if (BEZ_BOX_LINK[b] == l) return b;
  return -1;
}

// ---- Philox4x32-10 (reset-noise stream) and the domain-randomisation noise quad: bez_dr_noise.h (included above)
// observation element i (flat index into this shard's obs_buf) with the domain-randomisation noise of this frame
BEZ_DEV float obs_with_noise(const Params& P, long long i, float v) {
  float z[4];
  dr_noise_quad(P.seed, P.env_off, P.dr_state->frame, 0, i >> 2, z);
  return v + fmaf(z[i & 3], P.dr_state->noise[1], P.dr_state->noise[0]);
}

// Per-lane working copy of the generalized state
struct EnvState {
  V3 root_pos; float rq[4]; V3 root_lin, root_ang;
  float q[BEZ_ND], qd[BEZ_ND];
  V3 ball_pos; float bq[4]; V3 ball_lin, ball_ang;
};

BEZ_DEV void load_state(const float* __restrict__ s, int n, int e, EnvState& S) {
  auto ld = [&](int f) { return s[(size_t)f * n + e]; };
  S.root_pos = mk(ld(F_ROOT_POS), ld(F_ROOT_POS + 1), ld(F_ROOT_POS + 2));
#pragma unroll
  for (int i = 0; i < 4; ++i) { S.rq[i] = ld(F_ROOT_QUAT + i); S.bq[i] = ld(F_BALL_QUAT + i); }
  S.root_lin = mk(ld(F_ROOT_LIN), ld(F_ROOT_LIN + 1), ld(F_ROOT_LIN + 2));
  S.root_ang = mk(ld(F_ROOT_ANG), ld(F_ROOT_ANG + 1), ld(F_ROOT_ANG + 2));
#pragma unroll
  for (int j = 0; j < BEZ_ND; ++j) { S.q[j] = ld(F_Q + j); S.qd[j] = ld(F_QD + j); }
  S.ball_pos = mk(ld(F_BALL_POS), ld(F_BALL_POS + 1), ld(F_BALL_POS + 2));
  S.ball_lin = mk(ld(F_BALL_LIN), ld(F_BALL_LIN + 1), ld(F_BALL_LIN + 2));
  S.ball_ang = mk(ld(F_BALL_ANG), ld(F_BALL_ANG + 1), ld(F_BALL_ANG + 2));
}
BEZ_DEV void store_state(float* __restrict__ s, int n, int e, const EnvState& S) {
  auto st = [&](int f, float v) { s[(size_t)f * n + e] = v; };
  st(F_ROOT_POS, S.root_pos.x); st(F_ROOT_POS + 1, S.root_pos.y); st(F_ROOT_POS + 2, S.root_pos.z);
#pragma unroll
  for (int i = 0; i < 4; ++i) { st(F_ROOT_QUAT + i, S.rq[i]); st(F_BALL_QUAT + i, S.bq[i]); }
  st(F_ROOT_LIN, S.root_lin.x); st(F_ROOT_LIN + 1, S.root_lin.y); st(F_ROOT_LIN + 2, S.root_lin.z);
  st(F_ROOT_ANG, S.root_ang.x); st(F_ROOT_ANG + 1, S.root_ang.y); st(F_ROOT_ANG + 2, S.root_ang.z);
#pragma unroll
  for (int j = 0; j < BEZ_ND; ++j) { st(F_Q + j, S.q[j]); st(F_QD + j, S.qd[j]); }
  st(F_BALL_POS, S.ball_pos.x); st(F_BALL_POS + 1, S.ball_pos.y); st(F_BALL_POS + 2, S.ball_pos.z);
  st(F_BALL_LIN, S.ball_lin.x); st(F_BALL_LIN + 1, S.ball_lin.y); st(F_BALL_LIN + 2, S.ball_lin.z);
  st(F_BALL_ANG, S.ball_ang.x); st(F_BALL_ANG + 1, S.ball_ang.y); st(F_BALL_ANG + 2, S.ball_ang.z);
}

// Per-env model parameters that domain randomisation may change
struct EnvDyn {
  float mu;
  V3 g;
  float kp_scale[BEZ_ND], kd_scale[BEZ_ND], mass_scale[BEZ_NL], lo[BEZ_ND], hi[BEZ_ND];
};

// ---- contact: implicit spring-damper at a point against the ground plane z = 0.
// Folds the point's implicit stiffness into (IA, pA) of its body and returns the hit record.
struct Hit { V3 x; float fn0, kn, ct, ftx0, fty0; };
BEZ_DEV Hit hit_none() { Hit h; h.x = mk(0, 0, 0); h.fn0 = h.kn = h.ct = h.ftx0 = h.fty0 = 0.f; return h; }

BEZ_DEV Hit ground_contact(const Params& P, float ckn, float ccn, float mu, V3 x, float z, SV V, Sym6& IA, SV& pA) {
  Hit hit = hit_none();
  float d = -z;
  if (d > 0.f) {
    V3 vp = point_of(V, x);
    float kd = fmaf(P.h, ckn, ccn);
    float fn0 = fmaf(ckn, d, -kd * vp.z);
    if (fn0 > 0.f) {
      float kn = P.h * kd;
      float vt = fsqrt(fmaf(vp.x, vp.x, vp.y * vp.y));
      float ct = fminf(mu * fn0 * frcp(fmaxf(vt, P.veps)), P.ct);
      float kt = P.h * ct;
      add_point_stiffness_diag(IA, x, kt, kn);
      hit.x = x; hit.fn0 = fn0; hit.kn = kn; hit.ct = ct; hit.ftx0 = -ct * vp.x; hit.fty0 = -ct * vp.y;
      pA = pA - wrench_at(x, mk(hit.ftx0, hit.fty0, fn0));
    }
  }
  return hit;
}
BEZ_DEV V3 hit_force(const Params& P, const Hit& h, SV acc) {
  V3 ap = point_of(acc, h.x);
  return mk(fmaf(-P.h * h.ct, ap.x, h.ftx0), fmaf(-P.h * h.ct, ap.y, h.fty0), fmaf(-h.kn, ap.z, h.fn0));
}
BEZ_DEV void lds_store_hit(float* lds, int lane, int idx, const Hit& h) {
  float* p = lds + (size_t)(LDS_HIT + idx * HIT_STRIDE) * BLOCK + lane;
  p[0 * BLOCK] = h.x.x; p[1 * BLOCK] = h.x.y; p[2 * BLOCK] = h.x.z; p[3 * BLOCK] = h.fn0;
  p[4 * BLOCK] = h.kn; p[5 * BLOCK] = h.ct; p[6 * BLOCK] = h.ftx0; p[7 * BLOCK] = h.fty0;
}
BEZ_DEV Hit lds_load_hit(const float* lds, int lane, int idx) {
  const float* p = lds + (size_t)(LDS_HIT + idx * HIT_STRIDE) * BLOCK + lane;
  Hit h;
  h.x = mk(p[0 * BLOCK], p[1 * BLOCK], p[2 * BLOCK]); h.fn0 = p[3 * BLOCK];
  h.kn = p[4 * BLOCK]; h.ct = p[5 * BLOCK]; h.ftx0 = p[6 * BLOCK]; h.fty0 = p[7 * BLOCK];
  return h;
}

// ---- net contact force output (Isaac NET_CONTACT_FORCE rows).  Only the two foot rows are consumed by the
// observation (kick_env.py:193-196), so only those stay in registers; every other row goes straight to HBM.
struct CfOut { float* base; int n; V3 lf, rf; };  // base = &state[F_CF * n + env]
BEZ_DEV void cf_store(const CfOut& c, int body, V3 f) {
  c.base[(size_t)(body * 3 + 0) * c.n] = f.x; c.base[(size_t)(body * 3 + 1) * c.n] = f.y; c.base[(size_t)(body * 3 + 2) * c.n] = f.z;
}
// The tensor holds the MEAN force over the substeps of the control step (physx.contact_collection 2 = CC_ALL_SUBSTEPS,
// bez_kick.yaml:147): the first contributing substep stores w * f, later ones add.
BEZ_DEV void cf_accum(const CfOut& c, int body, V3 f, float w, bool first) {
  float* p = c.base + (size_t)(body * 3) * c.n;
  if (first) { p[0] = f.x * w; p[(size_t)c.n] = f.y * w; p[(size_t)2 * c.n] = f.z * w; }
  else { p[0] = fmaf(f.x, w, p[0]); p[(size_t)c.n] = fmaf(f.y, w, p[(size_t)c.n]); p[(size_t)2 * c.n] = fmaf(f.z, w, p[(size_t)2 * c.n]); }
}
// Isaac Gym's net contact force sums the NORMAL contact impulses only [ext] (checkpoint obs statistics);
// BEZ_FLAG_CF_WITH_FRICTION keeps the friction part.  Ground normal = +z.
BEZ_DEV V3 cf_ground(const Params& P, V3 f) { return (P.flags & BEZ_FLAG_CF_WITH_FRICTION) ? f : mk(0.f, 0.f, f.z); }
BEZ_DEV V3 cf_along(const Params& P, V3 f, V3 n) { return (P.flags & BEZ_FLAG_CF_WITH_FRICTION) ? f : n * dot(f, n); }

// ---- ball <-> leg-box contact bookkeeping (deepest penetration only)
struct BallSel {
  int link;      // -1 none
  float depth;
  V3 n, P;       // world normal (box -> ball), contact point rel. O
  // filled when the selected link is visited in pass 1:
  Sym3 A;        // effective point stiffness seen by the link  (K^-1 + G)^-1
  V3 f0p;        // explicit force on the link
  V3 x, xb;      // contact point rel. O / rel. ball centre
};

// sphere (centre bc rel. O) against box `b` of a link with frame (E, r)
BEZ_DEV void test_box_at(int l, V3 cl, V3 he, const M3& E, V3 r, V3 bc, BallSel& sel) {
  const float R = (float)BEZ_BALL_RADIUS;
  V3 ql = mulT(E, bc - r) - cl;
  V3 cp = mk(fminf(fmaxf(ql.x, -he.x), he.x), fminf(fmaxf(ql.y, -he.y), he.y), fminf(fmaxf(ql.z, -he.z), he.z));
  bool inside = (cp.x == ql.x) && (cp.y == ql.y) && (cp.z == ql.z);
  V3 nl; float depth;
  if (!inside) {
    V3 dlt = ql - cp;
    float d2 = dot(dlt, dlt);
    float dist = fsqrt(d2);
    depth = R - dist;
    if (!(depth > 0.f)) return;
    nl = dlt * frsq(d2);
  } else {
    float dx = he.x - fabsf(ql.x), dy = he.y - fabsf(ql.y), dz = he.z - fabsf(ql.z);
    int ax = 0; float md = dx;
    if (dy < md) { md = dy; ax = 1; }
    if (dz < md) { md = dz; ax = 2; }
    nl = mk(0, 0, 0);
    if (ax == 0) { nl.x = ql.x >= 0.f ? 1.f : -1.f; cp.x = nl.x * he.x; }
    else if (ax == 1) { nl.y = ql.y >= 0.f ? 1.f : -1.f; cp.y = nl.y * he.y; }
    else { nl.z = ql.z >= 0.f ? 1.f : -1.f; cp.z = nl.z * he.z; }
    depth = R + md;
  }
  if (depth > sel.depth) {
    sel.depth = depth; sel.link = l;
    sel.n = mul(E, nl);
    sel.P = r + mul(E, cp + cl);
  }
}
template <int B>
BEZ_DEV void test_box(const M3& E, V3 r, V3 bc, BallSel& sel) {
  test_box_at(BEZ_BOX_LINK[B], mk((float)BEZ_BOX_CENTER[B][0], (float)BEZ_BOX_CENTER[B][1], (float)BEZ_BOX_CENTER[B][2]),
              mk((float)BEZ_BOX_HALF[B][0], (float)BEZ_BOX_HALF[B][1], (float)BEZ_BOX_HALF[B][2]), E, r, bc, sel);
}
// ball <-> torso: the stl asset's bounding box of the torso mesh, or the box asset's own torso box (BEZ_FLAG_BOX_ASSET)
BEZ_DEV void test_torso_box(const Params& P, const M3& E, V3 r, V3 bc, BallSel& sel) {
  constexpr int B = BEZ_TORSO_BOX;
  static_assert(BEZ_BOX_LINK[B] == 0, "the torso box is the last entry of the box table");
  const bool bx = (P.flags & BEZ_FLAG_BOX_ASSET) != 0;
  auto pick = [&](double a, double b) { return (float)a == (float)b ? (float)a : (bx ? (float)b : (float)a); };
  test_box_at(0, mk(pick(BEZ_BOX_CENTER[B][0], BEZ_TORSO_BOX_CENTER_BOX[0]), pick(BEZ_BOX_CENTER[B][1], BEZ_TORSO_BOX_CENTER_BOX[1]),
                    pick(BEZ_BOX_CENTER[B][2], BEZ_TORSO_BOX_CENTER_BOX[2])),
              mk(pick(BEZ_BOX_HALF[B][0], BEZ_TORSO_BOX_HALF_BOX[0]), pick(BEZ_BOX_HALF[B][1], BEZ_TORSO_BOX_HALF_BOX[1]),
                 pick(BEZ_BOX_HALF[B][2], BEZ_TORSO_BOX_HALF_BOX[2])), E, r, bc, sel);
}

// The free ball about its own centre with its (implicit) ground contact folded in: the 6x6 "mass" is
// block-diagonal in the pairs (wy,vx), (wx,vy) and the scalars wz, vz -> closed-form inverse.
struct BallBody {
  float i_wz, i_vz;             // 1/Ib, 1/(m + kn)
  float a11, a12, a22;          // inverse of [[Ib + kt R^2, -kt R], [-kt R, m + kt]]  (wy, vx)
  float b11, b12, b22;          // inverse of [[Ib + kt R^2, +kt R], [+kt R, m + kt]]  (wx, vy)
  SV pb;                        // bias: M a + pb = external
  Hit ghit; bool ground;
};
BEZ_DEV SV ball_minv(const BallBody& B, SV f) {
  SV a;
  a.a.z = f.a.z * B.i_wz; a.l.z = f.l.z * B.i_vz;
  a.a.y = fmaf(B.a11, f.a.y, B.a12 * f.l.x); a.l.x = fmaf(B.a12, f.a.y, B.a22 * f.l.x);
  a.a.x = fmaf(B.b11, f.a.x, B.b12 * f.l.y); a.l.y = fmaf(B.b12, f.a.x, B.b22 * f.l.y);
  return a;
}
BEZ_DEV BallBody ball_setup(const Params& P, float mu, V3 g, float ball_z, V3 ball_ang, V3 ball_lin) {
  const float R = (float)BEZ_BALL_RADIUS, mb = (float)BEZ_BALL_MASS, Ib = (float)BEZ_BALL_INERTIA;
  BallBody B;
  B.pb = mksv(mk(0, 0, 0), g * (-mb));
  Sym6 dummy = sym6zero();
  SV Vb = mksv(ball_ang, ball_lin);
  B.ghit = ground_contact(P, P.bkn, P.bcn, mu, mk(0, 0, -R), ball_z - R, Vb, dummy, B.pb);
  B.ground = B.ghit.kn > 0.f;
  float kt = P.h * B.ghit.ct, kn = B.ghit.kn;
  // det = (Ib + kt R^2)(m + kt) - kt^2 R^2 = Ib m + Ib kt + m kt R^2  (expanded: no cancellation)
  float p = fmaf(kt, R * R, Ib), s = mb + kt, o = kt * R;
  float idet = frcp(fmaf(Ib, mb, fmaf(Ib, kt, mb * kt * R * R)));
  B.a11 = s * idet; B.a22 = p * idet; B.a12 = o * idet;    // inverse of [[p,-o],[-o,s]] = 1/det [[s,o],[o,p]]
  B.b11 = s * idet; B.b22 = p * idet; B.b12 = -o * idet;   // inverse of [[p, o],[ o,s]] = 1/det [[s,-o],[-o,p]]
  B.i_wz = 1.0f / Ib; B.i_vz = frcp(mb + kn);
  return B;
}

// Evaluate the ball<->link contact when the selected link is reached in pass 1 (needs the link velocity).
BEZ_DEV void ball_link_contact(const Params& P, float mu, V3 ball_ang, V3 ball_lin, const BallBody& B, V3 bc, SV Vl, BallSel& sel) {
  V3 x = sel.P, xb = sel.P - bc, n = sel.n;
  SV Vb = mksv(ball_ang, ball_lin);
  V3 u = point_of(Vl, x) - point_of(Vb, xb);
  float un = dot(u, n);
  float kd = fmaf(P.h, P.bkn, P.bcn);
  float fmag = fmaf(P.bkn, sel.depth, kd * un);
  if (!(fmag > 0.f)) { sel.link = -1; return; }
  V3 ut = u - n * un;
  float vt = fsqrt(dot(ut, ut));
  float ct = fminf(mu * fmag * frcp(fmaxf(vt, P.veps)), P.ct);
  float kn = P.h * kd, kt = P.h * ct;
  V3 f0 = -(n * fmag + ut * ct);
  Sym3 K;  // kn nn^T + kt (1 - nn^T)
  float dk = kn - kt;
  K.xx = fmaf(dk * n.x, n.x, kt); K.yy = fmaf(dk * n.y, n.y, kt); K.zz = fmaf(dk * n.z, n.z, kt);
  K.xy = dk * n.x * n.y; K.xz = dk * n.x * n.z; K.yz = dk * n.y * n.z;
  // G = Jb Mb^-1 Jb^T, gb = Jb Mb^-1 pb with Jb^T e_j = wrench_at(xb, e_j)
  SV w0 = wrench_at(xb, mk(1, 0, 0)), w1 = wrench_at(xb, mk(0, 1, 0)), w2 = wrench_at(xb, mk(0, 0, 1));
  SV y0 = ball_minv(B, w0), y1 = ball_minv(B, w1), y2 = ball_minv(B, w2), yp = ball_minv(B, B.pb);
  M3 G;
  G.m00 = dot(w0, y0); G.m01 = dot(w0, y1); G.m02 = dot(w0, y2);
  G.m10 = dot(w1, y0); G.m11 = dot(w1, y1); G.m12 = dot(w1, y2);
  G.m20 = dot(w2, y0); G.m21 = dot(w2, y1); G.m22 = dot(w2, y2);
  V3 gb = mk(dot(w0, yp), dot(w1, yp), dot(w2, yp));
  M3 Km = to_m3(K);
  M3 KG = matmul(Km, G);
  KG.m00 += 1.f; KG.m11 += 1.f; KG.m22 += 1.f;
  M3 inv = inverse(KG);
  M3 A = matmul(inv, Km);
  sel.A.xx = A.m00; sel.A.yy = A.m11; sel.A.zz = A.m22;
  sel.A.xy = 0.5f * (A.m01 + A.m10); sel.A.xz = 0.5f * (A.m02 + A.m20); sel.A.yz = 0.5f * (A.m12 + A.m21);
  sel.f0p = mul(inv, f0 - mul(Km, gb));
  sel.x = x; sel.xb = xb;
}

// ---- leg <-> leg self-collision (kick_env.py:365-366: collision_filter 0).  Each leg box is a capsule (BEZ_CAP_*);
// a penetrating left x right pair is ONE spring-damper + regularised Coulomb point contact with equal and opposite forces
// on the two links (a contact inside the tree closes a loop the ABA recursion cannot fold in): lambda0 = k (depth - h u_n)
// - c u_n from the state at the start of the substep, then every pair of the env scaled by ONE factor that stands for the
// implicit part (self_scale below; the oracle's self_contact_scale).
BEZ_DEV float clamp01(float s) { return fminf(fmaxf(s, 0.f), 1.f); }
// closest points of two (non-degenerate) segments, branch-free: the unconstrained s, the t it implies, and -- when that t
// had to be clamped to its segment -- the s that is closest to the clamped end point
BEZ_DEV void segment_closest(V3 p1, V3 q1, V3 p2, V3 q2, V3& c1, V3& c2) {
  V3 d1 = q1 - p1, d2 = q2 - p2, r = p1 - p2;
  float a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, r), c = dot(d1, r), b = dot(d1, d2);
  float den = fmaf(a, e, -b * b);
  float s = den > 1e-12f ? clamp01(fmaf(b, f, -c * e) * frcp(den)) : 0.f;
  float t = fmaf(b, s, f) * frcp(e);
  float tc = clamp01(t);
  float s2 = clamp01(fmaf(b, tc, -c) * frcp(a));
  s = (t != tc) ? s2 : s;
  c1 = fma3(d1, s, p1);
  c2 = fma3(d2, tc, p2);
}
// capsule pair (world endpoints rel. O, link velocities about O) -> force f on link a at x; fn = its normal part
BEZ_DEV bool self_pair(const Params& P, float mu, float ra, float rb, V3 a0, V3 a1, V3 b0, V3 b1, SV Va, SV Vb, V3& x, V3& f, V3& fn) {
  V3 ca, cb;
  segment_closest(a0, a1, b0, b1, ca, cb);
  V3 dl = ca - cb;
  float d2 = dot(dl, dl), rs = ra + rb;
  if (!(d2 < rs * rs) || !(d2 > 1e-12f)) return false;
  float idist = frsq(d2), depth = rs - d2 * idist;
  V3 n = dl * idist;
  x = fma3(n, rb - 0.5f * depth, cb);
  V3 u = point_of(Va, x) - point_of(Vb, x);
  float un = dot(u, n);
  float fmag = fmaf(P.self_kn, depth, -fmaf(P.h, P.self_kn, P.self_cn) * un);
  if (!(fmag > 0.f)) return false;
  V3 ut = u - n * un;
  float vt = fsqrt(dot(ut, ut));
  float ct = fminf(mu * fmag * frcp(fmaxf(vt, P.veps)), P.self_cn);
  fn = n * fmag;
  f = fn - ut * ct;
  return true;
}
// world endpoints of the capsules carried by link L
template <int L>
BEZ_DEV void link_capsules(const M3& E, V3 r, SV V, V3* c0, V3* c1, SV* cV) {
#pragma unroll
  for (int c = 0; c < BEZ_NCAP; ++c) {
    if (BEZ_CAP_LINK[c] == L) {
      c0[c] = r + mul(E, mk((float)BEZ_CAP_P0[c][0], (float)BEZ_CAP_P0[c][1], (float)BEZ_CAP_P0[c][2]));
      c1[c] = r + mul(E, mk((float)BEZ_CAP_P1[c][0], (float)BEZ_CAP_P1[c][1], (float)BEZ_CAP_P1[c][2]));
      cV[c] = V;
    }
  }
}

// ---- one kinematic step down the tree: child frame / joint axis / velocity from the parent's
template <int L>
BEZ_DEV void link_kinematics(float q, float qd, M3& E, V3& r, SV& V, SV& S, SV& cb, float quirk_z = BEZ_QUIRK_RZ) {
  constexpr int ax = axis_index(L);
  constexpr float sg = axis_sign(L);
  V3 a = col(E, ax) * sg;  // joint axis in world (parent frame column; unchanged by the joint rotation)
  constexpr float tx = (float)BEZ_LINK_XYZ[L][0], ty = (float)BEZ_LINK_XYZ[L][1], tz = (float)BEZ_LINK_XYZ[L][2];
  if (tx != 0.f) r = fma3(col(E, 0), tx, r);
  if (ty != 0.f) r = fma3(col(E, 1), ty, r);
  if constexpr (L == BEZ_BOXCL_LINK) r = fma3(col(E, 2), quirk_z, r);  // see quirk_rz
  else if (tz != 0.f) r = fma3(col(E, 2), tz, r);
  float s, c;
  fsincos(sg * q, &s, &c);
  E = rotate_about(E, ax, s, c);
  S = mksv(a, cross(r, a));
  SV vj = S * qd;
  V = V + vj;
  cb = crm(V, vj);
}

// rigid-body inertia of link L about O in compact form + its bias force (velocity product - gravity)
struct LinkInertia { float m; V3 h; Sym3 Ibar; };
template <int L, bool CL = false>
BEZ_DEV void link_inertia(float ms, V3 g, const M3& E, V3 r, SV V, LinkInertia& I, SV& pA) {
  const float m = (float)link_mass<CL>(L) * ms;
  const V3 cl = mk((float)link_com<CL>(L, 0), (float)link_com<CL>(L, 1), (float)link_com<CL>(L, 2));
  Sym3 Il;
  Il.xx = (float)link_inertia_c<CL>(L, 0) * ms; Il.yy = (float)link_inertia_c<CL>(L, 1) * ms; Il.zz = (float)link_inertia_c<CL>(L, 2) * ms;
  Il.xy = (float)link_inertia_c<CL>(L, 3) * ms; Il.xz = (float)link_inertia_c<CL>(L, 4) * ms; Il.yz = (float)link_inertia_c<CL>(L, 5) * ms;
  V3 c = r + mul(E, cl);
  // the URDF inertias are diagonal in the link frame for most links: drop the products with structural zeros
  constexpr bool DIAG = link_inertia_c<CL>(L, 3) == 0. && link_inertia_c<CL>(L, 4) == 0. && link_inertia_c<CL>(L, 5) == 0.;
  Sym3 Iw = DIAG ? rotate_inertia_diag(E, Il.xx, Il.yy, Il.zz) : rotate_inertia(E, Il);
  float cc = dot(c, c);
  I.m = m; I.h = c * m;
  I.Ibar.xx = fmaf(m, cc - c.x * c.x, Iw.xx); I.Ibar.yy = fmaf(m, cc - c.y * c.y, Iw.yy); I.Ibar.zz = fmaf(m, cc - c.z * c.z, Iw.zz);
  I.Ibar.xy = fmaf(-m * c.x, c.y, Iw.xy); I.Ibar.xz = fmaf(-m * c.x, c.z, Iw.xz); I.Ibar.yz = fmaf(-m * c.y, c.z, Iw.yz);
  // momentum  I V = [Ibar w + h x v ; m v - h x w]
  V3 ha = mul(I.Ibar, V.a) + cross(I.h, V.l);
  V3 hl = V.l * m - cross(I.h, V.a);
  pA = crf(V, mksv(ha, hl)) - mksv(cross(I.h, g), g * m);
}
BEZ_DEV void add_link_inertia(Sym6& IA, const LinkInertia& I) {
  add_to(IA.A, I.Ibar);
  // B += skew(h)
  IA.B.m01 -= I.h.z; IA.B.m02 += I.h.y; IA.B.m10 += I.h.z; IA.B.m12 -= I.h.x; IA.B.m20 -= I.h.y; IA.B.m21 += I.h.x;
  IA.C.xx += I.m; IA.C.yy += I.m; IA.C.zz += I.m;
}

// ground points of link L (compile-time filtered), using the link's frame and velocity
template <int L, bool CL = false>
BEZ_DEV void link_ground_points(const Params& P, float mu, float root_z, const M3& E, V3 r, SV V, Sym6& IA, SV& pA,
                                float* lds, int lane, bool keep) {
#pragma unroll
  for (int i = 0; i < BEZ_NPT; ++i) {
    if (BEZ_PT_LINK[i] == L) {
      V3 pl = mk(pt_pos_of<CL>(P, i, 0), pt_pos_of<CL>(P, i, 1), pt_pos_of<CL>(P, i, 2));
      V3 x = r + mul(E, pl);
      Hit hit = ground_contact(P, P.kn, P.cn, mu, x, root_z + x.z, V, IA, pA);
      if (keep) lds_store_hit(lds, lane, i, hit);
    }
  }
}
template <int L>
BEZ_DEV V3 link_ground_forces(const Params& P, SV acc, const float* lds, int lane) {
  V3 f = mk(0, 0, 0);
#pragma unroll
  for (int i = 0; i < BEZ_NPT; ++i) {
    if (BEZ_PT_LINK[i] == L) {
      Hit hit = lds_load_hit(lds, lane, i);
      if (hit.kn > 0.f) f = f + hit_force(P, hit, acc);
    }
  }
  return f;
}
// cleats: every ground point of the foot link reports into its own cleat body's row
template <int L>
BEZ_DEV void link_ground_forces_cleats(const Params& P, SV acc, const float* lds, int lane, const CfOut& co, bool first) {
#pragma unroll
  for (int i = 0; i < BEZ_NPT; ++i) {
    if (BEZ_PT_LINK[i] == L) {
      Hit hit = lds_load_hit(lds, lane, i);
      V3 f = hit.kn > 0.f ? cf_ground(P, hit_force(P, hit, acc)) : mk(0, 0, 0);
      cf_accum(co, BEZ_PT_BODY_CL[i], f, P.cf_w, first);
    }
  }
}

// ---- BEZ_FLAG_ALL_GROUND_SHAPES (one-env-per-lane kernel only: the scenario harness): ground contact at the corners of every collision
// shape of soccerbot_stl.urdf (bez_shapes_gen.h).  118 records do not fit LDS beside the pass-3 operands: they live in a global scratch
// buffer (coalesced: [(point * 8 + k) * n + env]).  Pass 1 has the link frames: evaluate and store; pass 2 folds a link's records into its
// articulated inertia / bias; pass 3 resolves the forces into the body rows.
BEZ_DEV void xhit_store(const Params& P, int e, int idx, const Hit& h) {
  float* p = P.xhit + (size_t)(idx * 8) * P.n + e;
  const size_t n = (size_t)P.n;
  p[0] = h.x.x; p[n] = h.x.y; p[2 * n] = h.x.z; p[3 * n] = h.fn0; p[4 * n] = h.kn; p[5 * n] = h.ct; p[6 * n] = h.ftx0; p[7 * n] = h.fty0;
}
BEZ_DEV Hit xhit_load(const Params& P, int e, int idx) {
  const float* p = P.xhit + (size_t)(idx * 8) * P.n + e;
  const size_t n = (size_t)P.n;
  Hit h; h.x = mk(p[0], p[n], p[2 * n]); h.fn0 = p[3 * n]; h.kn = p[4 * n]; h.ct = p[5 * n]; h.ftx0 = p[6 * n]; h.fty0 = p[7 * n];
  return h;
}
template <int L>
BEZ_DEV void link_xpoints_eval(const Params& P, int e, float mu, float root_z, const M3& E, V3 r, SV V) {
#pragma unroll
  for (int i = 0; i < BEZ_NXPT; ++i) {
    if (BEZ_XPT_LINK[i] == L) {
      V3 x = r + mul(E, mk((float)BEZ_XPT_POS[i][0], (float)BEZ_XPT_POS[i][1], (float)BEZ_XPT_POS[i][2]));
      Sym6 dI = sym6zero(); SV dp = svzero();
      xhit_store(P, e, i, ground_contact(P, P.kn, P.cn, mu, x, root_z + x.z, V, dI, dp));
    }
  }
}
template <int L>
BEZ_DEV void link_xpoints_fold(const Params& P, int e, Sym6& IA, SV& pA) {
#pragma unroll
  for (int i = 0; i < BEZ_NXPT; ++i) {
    if (BEZ_XPT_LINK[i] == L) {
      const Hit h = xhit_load(P, e, i);
      if (h.kn > 0.f) {
        add_point_stiffness_diag(IA, h.x, P.h * h.ct, h.kn);
        pA = pA - wrench_at(h.x, mk(h.ftx0, h.fty0, h.fn0));
      }
    }
  }
}
template <int L>
BEZ_DEV V3 link_xpoints_force(const Params& P, int e, SV acc) {   // the sum over the link's extra points: they all report into the link's own body row
  V3 f = mk(0, 0, 0);
#pragma unroll
  for (int i = 0; i < BEZ_NXPT; ++i) {
    if (BEZ_XPT_LINK[i] == L) {
      const Hit h = xhit_load(P, e, i);
      if (h.kn > 0.f) f = f + cf_ground(P, hit_force(P, h, acc));
    }
  }
  return f;
}
BEZ_DEV constexpr bool link_has_xpoints(int l) {
  for (int i = 0; i < BEZ_NXPT; ++i) if (BEZ_XPT_LINK[i] == l) return true;
  return false;
}

// joint drive / friction / limit terms and the ABA joint-space quantities for DOF d = L-1:  g = 1/D, w = (tau - S.pA)/D (pass 3
// forms qdd = w - g U.(a_parent + c)) and the held-parent acceleration qdd_hp = w - g U.c the predictors work with.  A joint whose
// held-parent rate would end the substep beyond the speed limit (kick_env.py:327) is a prescribed-rate joint: g = 0, w = the
// acceleration that puts it ON the limit -- same recursion, and the reaction reaches the parent through pA (oracle: dynamics_x).
template <int L>
BEZ_DEV void joint_terms(const Params& P, float kp_scale, float kd_scale, float lo, float hi, float q, float qd, float target, const Sym6& IA, SV pA,
                         SV S, SV cb, SV& U, float& g, float& w, float& qdd_hp, float stop_tau = 0.f, float stop_k = 0.f) {
  U = mul(IA, S);
  float J = dot(S, U) + P.armature;
  float kp = P.kp * kp_scale, kdm = P.kd * kd_scale;
  float tau_pd0 = fmaf(kp, target - q - P.h * qd, -kdm * qd);
  float k_pd = fmaf(P.h * P.h, kp, P.h * kdm);
  float cf = P.jfric * frcp(fmaxf(fabsf(qd), P.jf_veps));
  float k_f = P.h * cf, tau_f0 = -cf * qd;
  float k_l = 0.f, tau_l0 = 0.f;
  if (q < lo) { tau_l0 = fmaf(P.lim_k, lo - q - P.h * qd, -P.lim_d * qd); k_l = fmaf(P.h * P.h, P.lim_k, P.h * P.lim_d); }
  else if (q > hi) { tau_l0 = fmaf(P.lim_k, hi - q - P.h * qd, -P.lim_d * qd); k_l = fmaf(P.h * P.h, P.lim_k, P.h * P.lim_d); }
  tau_l0 += stop_tau; k_l += stop_k;   // BEZ_FLAG_ANKLE_STOP: the calf <-> foot-plate contact as a coupled limit of the two ankle joints (0 otherwise)
  float sp = dot(S, pA);
  float ucb = dot(U, cb);
  float bias = sp + ucb;
  float qdd_est = (tau_pd0 + tau_f0 + tau_l0 - bias) * frcp(J + k_pd + k_f + k_l);
  float tau_drive = fmaf(-k_pd, qdd_est, tau_pd0);
  float tau, Dj;
  if (tau_drive > P.effort) { tau = P.effort + tau_f0 + tau_l0; Dj = J + k_f + k_l; }
  else if (tau_drive < -P.effort) { tau = -P.effort + tau_f0 + tau_l0; Dj = J + k_f + k_l; }
  else { tau = tau_pd0 + tau_f0 + tau_l0; Dj = J + k_pd + k_f + k_l; }
  g = frcp(Dj);
  w = (tau - sp) * g;
  qdd_hp = fmaf(-ucb, g, w);
  // v_pred = qd + h qdd_hp beyond +-vel_limit  <=>  qdd_hp outside [alo, ahi], the accelerations that reach the limits in one substep
  const float ahi = (P.vel_limit - qd) * P.inv_h, alo = (-P.vel_limit - qd) * P.inv_h;
  const float fix = fminf(fmaxf(qdd_hp, alo), ahi);
  const bool lock = fix != qdd_hp;
  w = lock ? fix : w; g = lock ? 0.f : g; qdd_hp = fix;
}

// The scalar part of joint_terms for the lane-group kernel (bez_step_ws8q.hip), which forms Jraw = S.(IA S), sp = S.pA and
// ucb = (IA S).cb across a quad of lanes; the same arithmetic in the same order as above.
template <int L>
BEZ_DEV void joint_scalar(const Params& P, float kp_scale, float kd_scale, float lo, float hi, float q, float qd, float target, float Jraw, float sp, float ucb,
                          float& g, float& w, float& qdd_hp) {
  float J = Jraw + P.armature;
  float kp = P.kp * kp_scale, kdm = P.kd * kd_scale;
  float tau_pd0 = fmaf(kp, target - q - P.h * qd, -kdm * qd);
  float k_pd = fmaf(P.h * P.h, kp, P.h * kdm);
  float cf = P.jfric * frcp(fmaxf(fabsf(qd), P.jf_veps));
  float k_f = P.h * cf, tau_f0 = -cf * qd;
  float k_l = 0.f, tau_l0 = 0.f;
  if (q < lo) { tau_l0 = fmaf(P.lim_k, lo - q - P.h * qd, -P.lim_d * qd); k_l = fmaf(P.h * P.h, P.lim_k, P.h * P.lim_d); }
  else if (q > hi) { tau_l0 = fmaf(P.lim_k, hi - q - P.h * qd, -P.lim_d * qd); k_l = fmaf(P.h * P.h, P.lim_k, P.h * P.lim_d); }
  float bias = sp + ucb;
  float qdd_est = (tau_pd0 + tau_f0 + tau_l0 - bias) * frcp(J + k_pd + k_f + k_l);
  float tau_drive = fmaf(-k_pd, qdd_est, tau_pd0);
  float tau, Dj;
  if (tau_drive > P.effort) { tau = P.effort + tau_f0 + tau_l0; Dj = J + k_f + k_l; }
  else if (tau_drive < -P.effort) { tau = -P.effort + tau_f0 + tau_l0; Dj = J + k_f + k_l; }
  else { tau = tau_pd0 + tau_f0 + tau_l0; Dj = J + k_pd + k_f + k_l; }
  g = frcp(Dj);
  w = (tau - sp) * g;
  qdd_hp = fmaf(-ucb, g, w);
  const float ahi = (P.vel_limit - qd) * P.inv_h, alo = (-P.vel_limit - qd) * P.inv_h;
  const float fix = fminf(fmaxf(qdd_hp, alo), ahi);
  const bool lock = fix != qdd_hp;
  w = lock ? fix : w; g = lock ? 0.f : g; qdd_hp = fix;
}

// sums of the leg<->leg contact scale (oracle: self_contact_scale), accumulated joint by joint while the legs run pass 2
struct SelfSums { float am, as, f2; };
BEZ_DEV float self_scale(const Params& P, const SelfSums& Z) {
  if (!(Z.f2 > 0.f)) return 0.f;
  const float K = fmaf(P.h * P.h, P.self_kn, P.h * P.self_cn);
  const float sc = fmaf(-K, Z.am, Z.f2) * frcp(fmaf(SELF_IMPLICIT * K, Z.as, Z.f2));
  return fminf(fmaxf(sc, 0.f), 8.f);   // (NaN-safe the oracle's way: !(sc > 0) -> 0)
}

// compile-time loop: f(std::integral_constant<int, i>) for i in [0, N)
template <int... Is, class F>
BEZ_DEV void static_for_impl(std::integer_sequence<int, Is...>, F&& f) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
BEZ_DEV void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

// ---- passes 1+2 of one serial chain FIRST..FIRST+LEN-1 hanging off the torso.  Accumulates the chain's
// articulated inertia / bias into the torso's (IA0, pA0) and stages pass-3 data in LDS.
template <int FIRST, int LEN, bool CL>
BEZ_DEV void chain_up(const Params& P, const EnvDyn& D, const EnvState& S, const float* target, const M3& E0, SV V0,
                      const BallBody& ball, V3 bc, BallSel& sel, const SV* selfw, Sym6& IA0, SV& pA0, SV& pS0, SelfSums& Z, float* lds, int lane, bool keep, int e) {
  LinkInertia LI[LEN];
  constexpr bool LEG = (FIRST == 5 || FIRST == 13);
  M3 Ecalf = E0; V3 rcalf = mk(0, 0, 0);   // BEZ_FLAG_ANKLE_STOP: the calf's frame, kept from pass 1 (legs only)
  SV pAl[LEN], Sl[LEN], cbl[LEN];
  M3 E = E0;
  V3 r = mk(0, 0, 0);
  SV V = V0;
  // pass 1: root -> tip
  static_for<LEN>([&](auto I) {
    constexpr int i = decltype(I)::value;
    constexpr int L = FIRST + i;
    link_kinematics<L>(S.q[L - 1], S.qd[L - 1], E, r, V, Sl[i], cbl[i], quirk_rz<CL>(P.flags));
    link_inertia<L, CL>(D.mass_scale[L], D.g, E, r, V, LI[i], pAl[i]);
    if constexpr (link_has_box(L)) {
      if (sel.link == L) ball_link_contact(P, D.mu, S.ball_ang, S.ball_lin, ball, bc, V, sel);
    }
    if constexpr (!CL && link_has_xpoints(L)) { if (P.xhit) link_xpoints_eval<L>(P, e, D.mu, S.root_pos.z, E, r, V); }
    if constexpr (LEG && i == 3) { Ecalf = E; rcalf = r; }
  });
  // BEZ_FLAG_ANKLE_STOP (oracle: ankle_stop): the bottom corners of the calf box against the top face of the foot plate of the same leg.
  // The gap depends on the ankle-pitch and foot-roll angles alone, so the contact is the joint-space force tau_d = J_d lambda on those two
  // joints, lambda = -k (g + h g') - c g' - (h^2 k + h c) sum J_e qdd_e: the own-joint part goes into each joint's D, the cross term is dropped.
  float stop_tau[2] = {0.f, 0.f}, stop_k[2] = {0.f, 0.f};   // [0] ankle pitch (link FIRST+4), [1] foot roll (link FIRST+5)
  if constexpr (LEG && !CL) {
    if (P.flags & BEZ_FLAG_ANKLE_STOP) {
      constexpr int side = FIRST == 5 ? 0 : 1, bcf = 2 + 5 * side, bff = 4 + 5 * side;
      static_assert(BEZ_BOX_LINK[bcf] == FIRST + 3 && BEZ_BOX_LINK[bff] == FIRST + 5, "calf / foot boxes of this leg");
      const float ztop = (float)(BEZ_BOX_CENTER[bff][2] + BEZ_BOX_HALF[bff][2]);
      const V3 nf = col(E, 2);   // (E, r are the foot's after pass 1)
      const float kimp = fmaf(P.h * P.h, STOP_KN, P.h * STOP_CN);
      float* rec = lds + (size_t)(LDS_STOP + side * STOP_STRIDE) * BLOCK + lane;
      int k = 0;
#pragma unroll
      for (int cx = -1; cx <= 1; cx += 2) {
#pragma unroll
        for (int cy = -1; cy <= 1; cy += 2) {
          const V3 pl = mk((float)(BEZ_BOX_CENTER[bcf][0] + cx * BEZ_BOX_HALF[bcf][0]), (float)(BEZ_BOX_CENTER[bcf][1] + cy * BEZ_BOX_HALF[bcf][1]),
                           (float)(BEZ_BOX_CENTER[bcf][2] - BEZ_BOX_HALF[bcf][2]));
          const V3 x = rcalf + mul(Ecalf, pl);
          const float g = dot(nf, x - r) - ztop;
          const float Ja = -dot(nf, point_of(Sl[4], x)), Jf = -dot(nf, point_of(Sl[5], x));
          const float gd = fmaf(Ja, S.qd[FIRST + 3], Jf * S.qd[FIRST + 4]);   // dof indices: link - 1
          float lam0 = -STOP_KN * fmaf(P.h, gd, g) - STOP_CN * gd;
          const bool on = (g < 0.f) && (lam0 > 0.f);
          lam0 = on ? lam0 : 0.f;
          stop_tau[0] = fmaf(Ja, lam0, stop_tau[0]); stop_tau[1] = fmaf(Jf, lam0, stop_tau[1]);
          if (on) { stop_k[0] = fmaf(kimp * Ja, Ja, stop_k[0]); stop_k[1] = fmaf(kimp * Jf, Jf, stop_k[1]); }
          rec[(3 * k + 0) * BLOCK] = Ja; rec[(3 * k + 1) * BLOCK] = Jf; rec[(3 * k + 2) * BLOCK] = lam0;
          ++k;
        }
      }
      rec[12 * BLOCK] = nf.x; rec[13 * BLOCK] = nf.y; rec[14 * BLOCK] = nf.z;
    }
  }
  // tip: ground points of the chain-end link (E, r, V are still the tip's)
  Sym6 IA = sym6zero();
  SV pA = svzero();
  SV pS = svzero();  // leg<->leg contact wrenches (as bias forces), propagated next to pA (the predictors do not see them); their common scale comes later
  SV wsub = svzero();  // the same, summed over the subtree WITHOUT the projections: S . wsub = the contacts' generalised force on the joint
  link_ground_points<FIRST + LEN - 1, CL>(P, D.mu, S.root_pos.z, E, r, V, IA, pA, lds, lane, keep);
  // pass 2: tip -> root
  static_for<LEN>([&](auto I) {
    constexpr int i = LEN - 1 - decltype(I)::value;
    constexpr int L = FIRST + i;
    add_link_inertia(IA, LI[i]);
    pA = pA + pAl[i];
    if constexpr (link_has_box(L)) { pS = pS + selfw[L]; wsub = wsub + selfw[L]; }
    if constexpr (link_has_box(L)) {
      if (sel.link == L) {
        add_point_stiffness(IA, sel.x, sel.A);
        pA = pA - wrench_at(sel.x, sel.f0p);
      }
    }
    if constexpr (!CL && link_has_xpoints(L)) { if (P.xhit) link_xpoints_fold<L>(P, e, IA, pA); }
    SV U; float Dinv, uD, qhp;
    joint_terms<L>(P, D.kp_scale[L - 1], D.kd_scale[L - 1], D.lo[L - 1], D.hi[L - 1], S.q[L - 1], S.qd[L - 1], target[L - 1], IA, pA, Sl[i], cbl[i], U,
                   Dinv, uD, qhp, (LEG && i >= 4) ? stop_tau[i >= 4 ? i - 4 : 0] : 0.f, (LEG && i >= 4) ? stop_k[i >= 4 ? i - 4 : 0] : 0.f);
    SV UD = U * Dinv;
    const float duD = -dot(Sl[i], pS) * Dinv;
    if constexpr (FIRST == 5 || FIRST == 13) {   // only the legs carry leg<->leg contacts
      const float tq = dot(Sl[i], wsub);         // (sign: wsub holds bias forces = minus the wrenches; tq enters squared and times am's own sign convention below)
      Z.as = fmaf(tq * tq, Dinv, Z.as); Z.am = fmaf(-tq, qhp, Z.am);
    }
    float* p3 = lds + (size_t)(LDS_P3 + (L - 1) * P3_STRIDE) * BLOCK + lane;
    p3[0 * BLOCK] = UD.a.x; p3[1 * BLOCK] = UD.a.y; p3[2 * BLOCK] = UD.a.z; p3[3 * BLOCK] = UD.l.x; p3[4 * BLOCK] = UD.l.y; p3[5 * BLOCK] = UD.l.z;
    p3[6 * BLOCK] = uD;
    p3[7 * BLOCK] = Sl[i].a.x; p3[8 * BLOCK] = Sl[i].a.y; p3[9 * BLOCK] = Sl[i].a.z; p3[10 * BLOCK] = Sl[i].l.x; p3[11 * BLOCK] = Sl[i].l.y; p3[12 * BLOCK] = Sl[i].l.z;
    p3[13 * BLOCK] = cbl[i].a.x; p3[14 * BLOCK] = cbl[i].a.y; p3[15 * BLOCK] = cbl[i].a.z; p3[16 * BLOCK] = cbl[i].l.x; p3[17 * BLOCK] = cbl[i].l.y; p3[18 * BLOCK] = cbl[i].l.z;
    p3[19 * BLOCK] = duD;
    // Ia = IA - U U^T / D ;  pa = pA + Ia c + U u / D
    add_outer(IA, U, -Dinv);
    pA = pA + mul(IA, cbl[i]) + U * uD;
    pS = pS + U * duD;
  });
  add_to(IA0, IA);
  pA0 = pA0 + pA;
  pS0 = pS0 + pS;
}

// ---- pass 3 of one chain: joint accelerations from the torso acceleration; integrates the joints in
// place (semi-implicit Euler) and resolves contact forces on the way.  sc = the leg<->leg contacts' common scale.
template <int FIRST, int LEN, bool CL>
BEZ_DEV void chain_down(const Params& P, EnvState& S, SV a0, float sc, BallSel& sel, const V3* selfcf, V3& ball_link_force, CfOut& co, const float* lds,
                        int lane, bool keep, bool first, int e) {
  SV a = a0;
  constexpr int Lend = FIRST + LEN - 1;
  constexpr bool LEG = (FIRST == 5 || FIRST == 13);
  float qdd_ankle = 0.f;   // BEZ_FLAG_ANKLE_STOP: the ankle-pitch acceleration, needed with the foot roll's for the contact force
  V3 fstop = mk(0, 0, 0);  // ... and the force on the calf (the foot receives the opposite)
  V3 fend = mk(0, 0, 0);  // ball force on the chain-end link (a foot), if it is the selected box
  static_for<LEN>([&](auto I) {
    constexpr int i = decltype(I)::value;
    constexpr int L = FIRST + i;
    const float* p3 = lds + (size_t)(LDS_P3 + (L - 1) * P3_STRIDE) * BLOCK + lane;
    SV UD = mksv(mk(p3[0 * BLOCK], p3[1 * BLOCK], p3[2 * BLOCK]), mk(p3[3 * BLOCK], p3[4 * BLOCK], p3[5 * BLOCK]));
    float uD = fmaf(sc, p3[19 * BLOCK], p3[6 * BLOCK]);
    SV Sj = mksv(mk(p3[7 * BLOCK], p3[8 * BLOCK], p3[9 * BLOCK]), mk(p3[10 * BLOCK], p3[11 * BLOCK], p3[12 * BLOCK]));
    SV cb = mksv(mk(p3[13 * BLOCK], p3[14 * BLOCK], p3[15 * BLOCK]), mk(p3[16 * BLOCK], p3[17 * BLOCK], p3[18 * BLOCK]));
    SV ap = a + cb;
    float qdd = uD - dot(UD, ap);
    a = ap + Sj * qdd;
    float v = fmaf(P.h, qdd, S.qd[L - 1]);   // the speed limit is inside the dynamics (joint_terms): no rate is edited here
    S.qd[L - 1] = v;
    S.q[L - 1] = fmaf(P.h, v, S.q[L - 1]);
    V3 fx = mk(0, 0, 0);   // BEZ_FLAG_ALL_GROUND_SHAPES: this link's extra ground points
    if constexpr (!CL && link_has_xpoints(L)) { if (P.xhit) fx = link_xpoints_force<L>(P, e, a); }
    if constexpr (link_has_box(L)) {
      V3 f = selfcf[L] * sc + fx;
      if (sel.link == L) {
        ball_link_force = sel.f0p - mul(sel.A, point_of(a, sel.x));
        f = f + cf_along(P, ball_link_force, sel.n);
      }
      if (keep) {
        if constexpr (L == Lend) fend = f;
        else cf_accum(co, link_body<CL>(L), f, P.cf_w, first);
      }
    } else if constexpr (!CL && link_has_xpoints(L)) {
      if (keep && P.xhit) {
        if constexpr (L == Lend) fend = fx;                                  // head / forearms: joins the chain-end guard points below
        else cf_accum(co, link_body<CL>(L), fx, P.cf_w, first);              // neck: a row nothing else writes
      }
    }
    if constexpr (LEG && !CL) {   // BEZ_FLAG_ANKLE_STOP: lambda = lam0 - (h^2 k + h c) (Ja qdd_ankle + Jf qdd_foot) per corner, +lambda n on the calf, -lambda n on the foot
      if constexpr (i == 4) qdd_ankle = qdd;
      if constexpr (i == 5) {
        if (P.flags & BEZ_FLAG_ANKLE_STOP) {
          const float* rec = lds + (size_t)(LDS_STOP + (FIRST == 5 ? 0 : 1) * STOP_STRIDE) * BLOCK + lane;
          const float kimp = fmaf(P.h * P.h, STOP_KN, P.h * STOP_CN);
          float lam = 0.f;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float l0 = rec[(3 * k + 2) * BLOCK];
            const float lk = l0 - kimp * fmaf(rec[(3 * k) * BLOCK], qdd_ankle, rec[(3 * k + 1) * BLOCK] * qdd);
            lam += (l0 > 0.f && lk > 0.f) ? lk : 0.f;
          }
          fstop = mk(rec[12 * BLOCK], rec[13 * BLOCK], rec[14 * BLOCK]) * lam;
        }
      }
    }
  });
  if constexpr (LEG && !CL) {
    if (keep && (P.flags & BEZ_FLAG_ANKLE_STOP)) {   // the calf's row was stored / accumulated above (first substep: stored): add; the foot's share joins fend
      float* pc = co.base + (size_t)(link_body<CL>(FIRST + 3) * 3) * co.n;
      pc[0] = fmaf(fstop.x, P.cf_w, pc[0]); pc[(size_t)co.n] = fmaf(fstop.y, P.cf_w, pc[(size_t)co.n]); pc[(size_t)2 * co.n] = fmaf(fstop.z, P.cf_w, pc[(size_t)2 * co.n]);
      fend = fend - fstop;
    }
  }
  if (keep) {
    constexpr bool foot = (Lend == BEZ_LFOOT_LINK || Lend == BEZ_RFOOT_LINK);
    if constexpr (CL && foot) {  // the foot plate itself only feels the ball / the other leg; the ground acts on the cleats
      cf_accum(co, link_body<CL>(Lend), fend, P.cf_w, first);
      link_ground_forces_cleats<Lend>(P, a, lds, lane, co, first);
    } else {
      V3 f = fend + cf_ground(P, link_ground_forces<Lend>(P, a, lds, lane));
      if constexpr (Lend == BEZ_LFOOT_LINK) co.lf = first ? f * P.cf_w : fma3(f, P.cf_w, co.lf);
      else if constexpr (Lend == BEZ_RFOOT_LINK) co.rf = first ? f * P.cf_w : fma3(f, P.cf_w, co.rf);
      else cf_accum(co, link_body<CL>(Lend), f, P.cf_w, first);
    }
  }
}

BEZ_DEV void quat_integrate(float q[4], V3 w, float h) {
  float x = q[0], y = q[1], z = q[2], s = q[3];
  float dx = fmaf(w.x, s, fmaf(w.y, z, -w.z * y));
  float dy = fmaf(-w.x, z, fmaf(w.y, s, w.z * x));
  float dz = fmaf(w.x, y, fmaf(-w.y, x, w.z * s));
  float dw = -fmaf(w.x, x, fmaf(w.y, y, w.z * z));
  float hh = 0.5f * h;
  x = fmaf(hh, dx, x); y = fmaf(hh, dy, y); z = fmaf(hh, dz, z); s = fmaf(hh, dw, s);
  float n = frsq(fmaf(x, x, fmaf(y, y, fmaf(z, z, s * s))));
  q[0] = x * n; q[1] = y * n; q[2] = z * n; q[3] = s * n;
}

// ---- one substep of the articulated-body dynamics for this lane's env.  When `keep` the net contact force per body of
// this substep is accumulated (`first`: it starts the mean): foot rows in `co`, all other rows in HBM.
template <bool CL>
BEZ_DEV void substep(const Params& P, const EnvDyn& D, EnvState& S, const float* target, CfOut& co, float* lds, int lane, bool keep, bool first, int e) {
  const M3 E0 = quat_to_mat(S.rq[0], S.rq[1], S.rq[2], S.rq[3]);
  const SV V0 = mksv(S.root_ang, S.root_lin);
  const V3 bc = S.ball_pos - S.root_pos;  // ball centre rel. O
  // (a) leg frames: which box, if any, does the ball penetrate deepest (legs first, torso box last; strict >), and the
  //     leg capsules for the self-collision
  BallSel sel;
  sel.link = -1; sel.depth = 0.f; sel.n = sel.P = sel.f0p = sel.x = sel.xb = mk(0, 0, 0); sel.A = sym3zero();
  V3 cap0[BEZ_NCAP], cap1[BEZ_NCAP]; SV capV[BEZ_NCAP];
  {
    M3 E = E0; V3 r = mk(0, 0, 0); SV V = V0, Sj, cbj;
    static_for<6>([&](auto I) {
      constexpr int L = 5 + decltype(I)::value;
      link_kinematics<L>(S.q[L - 1], S.qd[L - 1], E, r, V, Sj, cbj);
      if constexpr (link_has_box(L)) { test_box<link_box(L)>(E, r, bc, sel); link_capsules<L>(E, r, V, cap0, cap1, capV); }
    });
    E = E0; r = mk(0, 0, 0); V = V0;
    static_for<6>([&](auto I) {
      constexpr int L = 13 + decltype(I)::value;
      link_kinematics<L>(S.q[L - 1], S.qd[L - 1], E, r, V, Sj, cbj, quirk_rz<CL>(P.flags));
      if constexpr (link_has_box(L)) { test_box<link_box(L)>(E, r, bc, sel); link_capsules<L>(E, r, V, cap0, cap1, capV); }
    });
    test_torso_box(P, E0, mk(0, 0, 0), bc, sel);
  }
  SV selfw[BEZ_NL]; V3 selfcf[BEZ_NL]; float selff2 = 0.f;   // per-link bias wrenches / reported forces of the leg<->leg pairs, and the sum of their |force|^2
#pragma unroll
  for (int l = 0; l < BEZ_NL; ++l) { selfw[l] = svzero(); selfcf[l] = mk(0, 0, 0); }
  if (!(P.flags & BEZ_FLAG_NO_SELF_COLLISION)) {
    static_for<BEZ_NCPAIR>([&](auto I) {
      constexpr int ia = BEZ_CPAIR[decltype(I)::value][0], ib = BEZ_CPAIR[decltype(I)::value][1];
      constexpr int la = BEZ_CAP_LINK[ia], lb = BEZ_CAP_LINK[ib];
      V3 x, f, fn;
      if (self_pair(P, D.mu, (float)BEZ_CAP_R[ia], (float)BEZ_CAP_R[ib], cap0[ia], cap1[ia], cap0[ib], cap1[ib], capV[ia], capV[ib], x, f, fn)) {
        SV w = wrench_at(x, f);
        selfw[la] = selfw[la] - w; selfw[lb] = selfw[lb] + w; selff2 = fmaf(f.x, f.x, fmaf(f.y, f.y, fmaf(f.z, f.z, selff2)));
        V3 fr = (P.flags & BEZ_FLAG_CF_WITH_FRICTION) ? f : fn;
        selfcf[la] = selfcf[la] + fr; selfcf[lb] = selfcf[lb] - fr;
      }
    });
  }
  // (b) ball free body with its ground contact
  BallBody ball = ball_setup(P, D.mu, D.g, S.ball_pos.z, S.ball_ang, S.ball_lin);
  // (c) torso: own inertia + guard points (+ the ball when the torso box is the deepest), then the five chains
  Sym6 IA0 = sym6zero();
  SV pA0;
  {
    LinkInertia I0;
    link_inertia<0, CL>(D.mass_scale[0], D.g, E0, mk(0, 0, 0), V0, I0, pA0);
    add_link_inertia(IA0, I0);
    link_ground_points<0, CL>(P, D.mu, S.root_pos.z, E0, mk(0, 0, 0), V0, IA0, pA0, lds, lane, keep);
    if (sel.link == 0) {
      ball_link_contact(P, D.mu, S.ball_ang, S.ball_lin, ball, bc, V0, sel);
      if (sel.link == 0) { add_point_stiffness(IA0, sel.x, sel.A); pA0 = pA0 - wrench_at(sel.x, sel.f0p); }
    }
  }
  SV pS0 = svzero(); SelfSums Z; Z.am = Z.as = 0.f; Z.f2 = selff2;
  chain_up<1, 2, CL>(P, D, S, target, E0, V0, ball, bc, sel, selfw, IA0, pA0, pS0, Z, lds, lane, keep, e);    // neck, head
  chain_up<3, 2, CL>(P, D, S, target, E0, V0, ball, bc, sel, selfw, IA0, pA0, pS0, Z, lds, lane, keep, e);    // left arm
  chain_up<5, 6, CL>(P, D, S, target, E0, V0, ball, bc, sel, selfw, IA0, pA0, pS0, Z, lds, lane, keep, e);    // left leg
  chain_up<11, 2, CL>(P, D, S, target, E0, V0, ball, bc, sel, selfw, IA0, pA0, pS0, Z, lds, lane, keep, e);   // right arm
  chain_up<13, 6, CL>(P, D, S, target, E0, V0, ball, bc, sel, selfw, IA0, pA0, pS0, Z, lds, lane, keep, e);   // right leg
  // the leg<->leg contacts' common scale (known before the root solve), then their share of the torso's bias
  const float sc = self_scale(P, Z);
  pA0 = pA0 + pS0 * sc;
  // (d) root: I0^A a0 = -p0^A; urdfAsset.fixBaseLink (BEZ_FLAG_FIX_BASE): the torso is welded to the world, a0 = 0
  SV a0 = solve_spd6(IA0, svzero() - pA0);
  if (P.flags & BEZ_FLAG_FIX_BASE) a0 = svzero();
  // (e) pass 3 + joint integration + contact forces
  V3 fl = mk(0, 0, 0);
  {
    V3 f0 = mk(0, 0, 0);
    if (sel.link == 0) { fl = sel.f0p - mul(sel.A, point_of(a0, sel.x)); f0 = cf_along(P, fl, sel.n); }
    if (keep) cf_accum(co, 0, f0 + cf_ground(P, link_ground_forces<0>(P, a0, lds, lane)), P.cf_w, first);
  }
  chain_down<1, 2, CL>(P, S, a0, sc, sel, selfcf, fl, co, lds, lane, keep, first, e);
  chain_down<3, 2, CL>(P, S, a0, sc, sel, selfcf, fl, co, lds, lane, keep, first, e);
  chain_down<5, 6, CL>(P, S, a0, sc, sel, selfcf, fl, co, lds, lane, keep, first, e);
  chain_down<11, 2, CL>(P, S, a0, sc, sel, selfcf, fl, co, lds, lane, keep, first, e);
  chain_down<13, 6, CL>(P, S, a0, sc, sel, selfcf, fl, co, lds, lane, keep, first, e);
  // (f) ball: Mb ab = -pb - Jb^T fl
  SV ab = ball_minv(ball, svzero() - ball.pb - wrench_at(sel.xb, fl));
  if (keep) {
    V3 fb = -cf_along(P, fl, sel.n);
    if (ball.ground) fb = fb + cf_ground(P, hit_force(P, ball.ghit, ab));
    cf_accum(co, nb_of<CL>(), fb, P.cf_w, first);
  }
  // (g) integrate root (spatial -> classical acceleration of the torso origin) and ball
  V3 vdot = a0.l + cross(S.root_ang, S.root_lin);
  S.root_ang = fma3(a0.a, P.h, S.root_ang);
  S.root_lin = fma3(vdot, P.h, S.root_lin);
  S.root_pos = fma3(S.root_lin, P.h, S.root_pos);
  quat_integrate(S.rq, S.root_ang, P.h);
  float damp = fmaxf(1.0f - P.h * P.ball_damp, 0.f);
  S.ball_lin = fma3(ab.l, P.h, S.ball_lin);
  S.ball_ang = fma3(ab.a, P.h, S.ball_ang) * damp;
  S.ball_pos = fma3(S.ball_lin, P.h, S.ball_pos);
  quat_integrate(S.bq, S.ball_ang, P.h);
}

// ---- env logic ---------------------------------------------------------------------------------------

// kick_env.py:779-850 for this lane's env
BEZ_DEV void env_reset(const Params& P, EnvState& S, float* target, CfOut& co, uint32_t& episode, int64_t genv) {
  uint32_t k0 = (uint32_t)P.seed, k1 = (uint32_t)(P.seed >> 32);
  float u[36];
#pragma unroll
  for (int b = 0; b < 9; ++b) {
    uint32_t c[4] = {(uint32_t)genv, (uint32_t)((uint64_t)genv >> 32), episode, (uint32_t)b};
    philox4x32_10(c, k0, k1);
#pragma unroll
    for (int k = 0; k < 4; ++k) u[b * 4 + k] = (float)(c[k] >> 8) * (1.0f / 16777216.0f);
  }
#pragma unroll
  for (int j = 0; j < BEZ_ND; ++j) {
    float off = fmaf(0.3f, u[j], -0.15f);  // torch_rand_float: (upper-lower)*rand + lower
    float vel = fmaf(0.2f, u[BEZ_ND + j], -0.1f);
    float q = (float)BEZ_DOF_DEFAULT[j] + off;
    q = fmaxf(fminf(q, (float)BEZ_DOF_UPPER[j]), (float)BEZ_DOF_LOWER[j]);
    S.q[j] = q; S.qd[j] = vel;
    target[j] = (float)BEZ_DOF_DEFAULT[j];
  }
  episode += 1;
  S.root_pos = mk(P.bez_init[0], P.bez_init[1], P.bez_init[2]);
  S.ball_pos = mk(P.ball_init[0], P.ball_init[1], P.ball_init[2]);
#pragma unroll
  for (int i = 0; i < 4; ++i) { S.rq[i] = P.bez_init[3 + i]; S.bq[i] = P.ball_init[3 + i]; }
  S.root_lin = S.root_ang = S.ball_lin = S.ball_ang = mk(0, 0, 0);
  for (int b = 0; b < BEZ_NBE_MAX; ++b) cf_store(co, b, mk(0, 0, 0));
  co.lf = co.rf = mk(0, 0, 0);
}

// vec_task.py:317 + kick_env.py:413-418
BEZ_DEV void env_targets(const Params& P, const float* __restrict__ act, float* target) {
#pragma unroll
  for (int j = 0; j < BEZ_ND; ++j) {
    float a = fminf(fmaxf(act[j], -P.clip), P.clip);
    if (j < 2) a = 0.f;
    float t = a + (float)BEZ_DOF_DEFAULT[j];
    target[j] = fmaxf(fminf(t, (float)BEZ_DOF_UPPER[j]), (float)BEZ_DOF_LOWER[j]);
  }
}

// kick_env.py:966-1040 on one foot's net contact force (mutated in place like the reference does)
BEZ_DEV void feet_no_cleats(float* f, float* out) {
#pragma unroll
  for (int i = 0; i < 3; ++i) if (!(fabsf(f[i]) > 0.01f)) f[i] = 0.f;
  float x = (fabsf(f[0]) > 0.f) ? 1.f : 0.f;
  if (f[0] == 0.f) x = 2.f;
  float y = (fabsf(f[1]) > 0.f) ? 1.f : 3.f;
  if (f[1] == 0.f) y = 3.f;
  float sensor = (x == 1.f) ? 0.f : 4.f;
  if (x == 2.f) sensor = 8.f;
  float cs = y + sensor;
  float o0 = -1.f, o1 = -1.f, o2 = -1.f, o3 = -1.f;
  if (cs == 1.f) { o0 = 1.f; }
  else if (cs == 3.f) { o0 = 1.f; o2 = 1.f; }
  else if (cs == 5.f) { o1 = 1.f; }
  else if (cs == 7.f) { o1 = 1.f; o3 = 1.f; }
  else if (cs == 9.f) { o0 = 1.f; o1 = 1.f; }
  else if (cs == 11.f) { o0 = o1 = o2 = o3 = 1.f; }
  if (f[2] < 1.f) { o0 = o1 = o2 = o3 = -1.f; }
  out[0] = o0; out[1] = o1; out[2] = o2; out[3] = o3;
}

// compute_observations + compute_reward (kick_env.py:749-777, 724-747) for one env, everything except the joint slots
// obs[0:36], in three independent parts (the 8-wave kernel runs them in different waves; env_observe_core chains them):
//   obs_imu_orn  tail[0:8]  = imu(6) off_orn(2)         from the root state, prev_lin_vel and the goal
//   obs_feet     tail[8:18] = feet(8) ball_init(2)      from the two foot rows (or the 8 cleat rows) of the net contact force
//   reward_of    reward and the reset flag of the NEXT step
// `pn` = sum_j (default_j - q_j)^2.  `goal` = this env's goal xy (bez_kick: the configured point; bez_walk: redrawn at reset).
struct OrnOut { float ux, uy, gn, ang_goal; };  // unit vector / distance to the goal, orient task's heading error
// compute_off_orn (kick_env.py:941-960) / orient_env.py:719-735 compute_off_angle: the two orientation slots (t6, t7) and the
// goal direction / heading error the walk and orient rewards reuse
BEZ_DEV OrnOut obs_off_orn(const Params& P, V3 root_pos, const float* rq, float goal_x, float goal_y, float& t6, float& t7) {
  OrnOut o;
  float gx = goal_x - root_pos.x, gy = goal_y - root_pos.y;
  o.gn = sqrtf(gx * gx + gy * gy);
  o.ux = gx / o.gn; o.uy = gy / o.gn;
  float qx = rq[0], qy = rq[1], qz = rq[2], qw = rq[3];
  // heading (cos yaw, sin yaw) with yaw = atan2(sy, cy) (get_euler_xyz [ext]): the unit vector (cy, sy)/|.| itself --
  // the % 2pi wrap and the atan2/sincos round trip of the reference only cost rounding (checked by the golden tests)
  float sy = 2.0f * (qw * qz + qx * qy), cy = qw * qw + qx * qx - qy * qy - qz * qz;
  float hn = 1.0f / sqrtf(sy * sy + cy * cy);
  float hs = sy * hn, hc = cy * hn;
  float cosv = hc * o.ux + hs * o.uy;
  float sinv = fabsf(o.ux * hs - o.uy * hc);
  t6 = sinv; t7 = -cosv;
  o.ang_goal = 0.f;
  if (P.task == BEZ_TASK_ORIENT) { o.ang_goal = P.goal_angle - atan2f(hs, hc); t6 = cosf(o.ang_goal); t7 = sinf(o.ang_goal); }
  return o;
}
BEZ_DEV OrnOut obs_imu_orn(const Params& P, V3 root_pos, const float* rq, V3 v, V3 w, float* prev, float goal_x, float goal_y, float* tail) {
  // IMU link = torso origin frame (soccerbot_stl.urdf:567-572)
  // compute_imu (kick_env.py:918-930), quaternion_to_matrix fed xyzw as (r,i,j,k) (quirk Q2)
  float pvx = P.use_prev ? prev[0] : v.x, pvy = P.use_prev ? prev[1] : v.y, pvz = P.use_prev ? prev[2] : v.z;
  float ax = (v.x - pvx) / P.dt - 0.f, ay = (v.y - pvy) / P.dt - 0.f, az = (v.z - pvz) / P.dt - (-1.f);
  float r = rq[0], i_ = rq[1], j_ = rq[2], k_ = rq[3];
  float two_s = 2.0f / (r * r + i_ * i_ + j_ * j_ + k_ * k_);
  float m00 = 1.f - two_s * (j_ * j_ + k_ * k_), m01 = two_s * (i_ * j_ - k_ * r), m02 = two_s * (i_ * k_ + j_ * r);
  float m10 = two_s * (i_ * j_ + k_ * r), m11 = 1.f - two_s * (i_ * i_ + k_ * k_), m12 = two_s * (j_ * k_ - i_ * r);
  float m20 = two_s * (i_ * k_ - j_ * r), m21 = two_s * (j_ * k_ + i_ * r), m22 = 1.f - two_s * (i_ * i_ + j_ * j_);
  const float LIN = (float)(2. * 9.81), ANG = 8.7266f;
  tail[0] = fminf(fmaxf(m00 * ax + m01 * ay + m02 * az, -LIN), LIN);
  tail[1] = fminf(fmaxf(m10 * ax + m11 * ay + m12 * az, -LIN), LIN);
  tail[2] = fminf(fmaxf(m20 * ax + m21 * ay + m22 * az, -LIN), LIN);
  tail[3] = fminf(fmaxf(w.x, -ANG), ANG); tail[4] = fminf(fmaxf(w.y, -ANG), ANG); tail[5] = fminf(fmaxf(w.z, -ANG), ANG);
  prev[0] = v.x; prev[1] = v.y; prev[2] = v.z;
  return obs_off_orn(P, root_pos, rq, goal_x, goal_y, tail[6], tail[7]);
}
// `cleats` = the 8 cleat rows of the net contact force (24 floats) with the cleats asset, else null
BEZ_DEV void obs_feet(const Params& P, CfOut& co, const float* cleats, float* feet, float* tail) {
  // feet (kick_env.py:538-576; with cleats kick_env.py:467-495,1044-1069: |force on the cleat| > 1 N)
  if (cleats) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float* f = cleats + 3 * k;
      feet[k] = sqrtf(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]) > 1.0f ? 1.f : -1.f;
    }
  } else {
    float fl[3] = {co.lf.x, co.lf.y, co.lf.z}, fr[3] = {co.rf.x, co.rf.y, co.rf.z};
    feet_no_cleats(fl, feet);
    feet_no_cleats(fr, feet + 4);
    co.lf = mk(fl[0], fl[1], fl[2]); co.rf = mk(fr[0], fr[1], fr[2]);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) tail[8 + i] = feet[i];
  tail[16] = P.ball_init[0]; tail[17] = P.ball_init[1];  // constant ball_init (quirk Q5, kick_env.py:776); bez_kick only
}
BEZ_DEV void reward_of(const Params& P, V3 root_pos, const float* rq, V3 v, V3 w, V3 ball_pos, V3 ball_lin, float pn, const OrnOut& o,
                       float& rew, int64_t& reset, int64_t progress, float goal_x, float goal_y) {
  if (P.task != BEZ_TASK_KICK) {
    // compute_bez_reward of walk_env.py:826-1031 / orient_env.py:843-1018
    const float qx = rq[0], qy = rq[1];
    const float ux = o.ux, uy = o.uy, gn = o.gn, ang_goal = o.ang_goal;
    const float vel_lin = sqrtf(dot(v, v)), vel_ang = sqrtf(dot(w, w)), vel_reward = sqrtf(dot(v, v) + dot(w, w)), pos_reward = sqrtf(pn);
    const float up_proj = 1.f - 2.f * (qx * qx + qy * qy);  // get_basis_vector(q, (0,0,1)).z
    const float dh = fabsf(1.f - up_proj);
    const float height_vel_pos = -((vel_reward * 0.05f + pos_reward * 0.05f) + dh);
    float reward, near;
    if (P.task == BEZ_TASK_WALK) {
      const float vfwd = ux * v.x + uy * v.y;
      const float vel_height = vfwd * 10.f - (dh + 5.f * (pos_reward * 0.05f));
      near = gn;
      reward = gn < 0.05f ? height_vel_pos : vel_height;
    } else {
      const float vel_height = fabsf(ang_goal) * -0.5f - (dh + 0.05f * (pos_reward * 0.05f));
      near = ang_goal;  // signed, as the reference compares it (orient_env.py:935)
      reward = ang_goal < 0.05f ? height_vel_pos : vel_height;
    }
    if (up_proj < 0.7f) { reset = 1; reward = -100.f; }
    const int state = (near < 0.05f) + (pos_reward < 0.15f) + (vel_ang < 0.1f) + (vel_lin < 0.1f);
    if (state == 4) { reset = 1; reward = 1000.0f - 1000.0f * ((float)progress / (float)P.max_len); }
    if (P.task == BEZ_TASK_WALK) {
      const float gnn = sqrtf(goal_x * goal_x + goal_y * goal_y);
      if (fabsf(atan2f(goal_y / gnn, goal_x / gnn) - atan2f(uy, ux)) > 1.5708f) { reset = 1; reward = -100.f; }
    } else {
      const float tx = root_pos.x - P.bez_init[0], ty = root_pos.y - P.bez_init[1];
      if (sqrtf(tx * tx + ty * ty) > 0.3f) { reset = 1; reward = -5.f; }
    }
    if (progress >= (int64_t)P.max_len) { reset = 1; reward = 0.f; }
    rew = reward;
    return;
  }
  // compute_bez_reward (kick_env.py:1224-1391)
  float bx = ball_pos.x, by = ball_pos.y;
  float dbx = bx - root_pos.x, dby = by - root_pos.y;
  float dbn = sqrtf(dbx * dbx + dby * dby);
  float vel_fwd = (dbx / dbn) * v.x + (dby / dbn) * v.y;
  float dgx = P.goal[0] - bx, dgy = P.goal[1] - by;
  float dgn = sqrtf(dgx * dgx + dgy * dgy);
  float b2gx = dgx / dgn, b2gy = dgy / dgn;
  float ball_fwd = b2gx * ball_lin.x + b2gy * ball_lin.y;
  float goal_angle_diff = fabsf(P.ang_init - atan2f(b2gy, b2gx));
  float vel_reward = sqrtf(dot(v, v) + dot(w, w));
  float pos_reward = sqrtf(pn);
  float height = fabsf(0.325f - root_pos.z);
  float kx = bx - P.ball_init[0], ky = by - P.ball_init[1];
  float kicked = sqrtf(kx * kx + ky * ky);
  float height_vel_pos = height + (vel_reward * 0.05f + pos_reward * 0.05f);
  float r_after = ball_fwd * 0.1f - height_vel_pos;
  float r_before = ball_fwd * 0.1f + (vel_fwd * 0.05f - height);
  float reward = kicked > 0.3f ? r_after : r_before;
  if (root_pos.z < 0.275f) { reset = 1; reward = -1.f; }
  float tx = root_pos.x - P.bez_init[0], ty = root_pos.y - P.bez_init[1];
  if (sqrtf(tx * tx + ty * ty) > 0.5f) { reset = 1; reward = -1.f; }
  if (goal_angle_diff > 1.5708f) { reset = 1; reward = -1.f; }
  if (dgn < 0.05f) { reset = 1; reward = 100.0f - 100.0f * ((float)progress / (float)P.max_len); }
  if (progress >= (int64_t)P.max_len) { reset = 1; reward = 0.f; }
  rew = reward;
}
// the three parts in sequence: writes tail[18] = imu(6) off_orn(2) feet(8) ball_init(2)
BEZ_DEV void env_observe_core(const Params& P, V3 root_pos, const float* rq, V3 v, V3 w, V3 ball_pos, V3 ball_lin, CfOut& co,
                              float* prev, float* feet, float* tail, float pn, float& rew, int64_t& reset, int64_t progress,
                              float goal_x, float goal_y, const float* cleats) {
  const OrnOut o = obs_imu_orn(P, root_pos, rq, v, w, prev, goal_x, goal_y, tail);
  obs_feet(P, co, cleats, feet, tail);
  reward_of(P, root_pos, rq, v, w, ball_pos, ball_lin, pn, o, rew, reset, progress, goal_x, goal_y);
}

BEZ_DEV void env_observe_reward(const Params& P, const EnvState& S, CfOut& co, float* prev, float* feet, float* obs,
                                float& rew, int64_t& reset, int64_t progress, float goal_x, float goal_y, const float* cleats) {
  float pn = 0.f;
#pragma unroll
  for (int j = 0; j < BEZ_ND; ++j) {
    obs[j] = S.q[j]; obs[BEZ_ND + j] = S.qd[j];  // kick_env.py:1409-1410
    float d = (float)BEZ_DOF_DEFAULT[j] - S.q[j];
    pn = fmaf(d, d, pn);
  }
  env_observe_core(P, S.root_pos, S.rq, S.root_lin, S.root_ang, S.ball_pos, S.ball_lin, co, prev, feet, obs + 36, pn, rew, reset, progress, goal_x, goal_y, cleats);
}

// ---- the fused kernel: PRE (targets) / SIM (substeps) / POST (bookkeeping, reset, obs, reward)
template <bool PRE, bool SIM, bool POST, bool DR, bool CL>
__global__ __launch_bounds__(BLOCK) void step_kernel(Params P) {
  __shared__ float lds[SIM ? LDS_SLOTS * BLOCK : 1];
  const int lane = threadIdx.x;
  const int e = blockIdx.x * BLOCK + lane;
  if (e >= P.n) return;
  const int n = P.n;
  float* st = P.state;
  EnvState S;
  load_state(st, n, e, S);
  float target[BEZ_ND];
  if (PRE) {
    float act[BEZ_ND];
#pragma unroll
    for (int j = 0; j < BEZ_ND; ++j) act[j] = P.actions[(size_t)e * BEZ_ND + j];
    env_targets(P, act, target);
    if (!SIM) {
#pragma unroll
      for (int j = 0; j < BEZ_ND; ++j) st[(size_t)(F_TARGET + j) * n + e] = target[j];
    }
  } else {
#pragma unroll
    for (int j = 0; j < BEZ_ND; ++j) target[j] = st[(size_t)(F_TARGET + j) * n + e];
  }
  CfOut co;
  co.base = st + (size_t)F_CF * n + e; co.n = n;
  co.lf = co.rf = mk(0, 0, 0);
  if (SIM) {
    EnvDyn D;
    D.mu = P.mu; D.g = mk(P.g[0], P.g[1], P.g[2]);
#pragma unroll
    for (int j = 0; j < BEZ_ND; ++j) { D.kp_scale[j] = 1.f; D.kd_scale[j] = 1.f; }
#pragma unroll
    for (int l = 0; l < BEZ_NL; ++l) D.mass_scale[l] = 1.f;
#pragma unroll
    for (int j = 0; j < BEZ_ND; ++j) { D.lo[j] = (float)BEZ_DOF_LOWER[j]; D.hi[j] = (float)BEZ_DOF_UPPER[j]; }
    if (DR) {
      if (P.dr_lower) {
#pragma unroll
        for (int j = 0; j < BEZ_ND; ++j) D.lo[j] = P.dr_lower[(size_t)e * BEZ_ND + j];
      }
      if (P.dr_upper) {
#pragma unroll
        for (int j = 0; j < BEZ_ND; ++j) D.hi[j] = P.dr_upper[(size_t)e * BEZ_ND + j];
      }
      if (P.dr_friction) D.mu = P.dr_friction[e];
      if (P.dr_gravity) D.g = mk(P.dr_gravity[(size_t)e * 3], P.dr_gravity[(size_t)e * 3 + 1], P.dr_gravity[(size_t)e * 3 + 2]);
      if (P.dr_kp) {
#pragma unroll
        for (int j = 0; j < BEZ_ND; ++j) D.kp_scale[j] = P.dr_kp[(size_t)e * BEZ_ND + j];
      }
      if (P.dr_kd) {
#pragma unroll
        for (int j = 0; j < BEZ_ND; ++j) D.kd_scale[j] = P.dr_kd[(size_t)e * BEZ_ND + j];
      }
      if (P.dr_mass) {
#pragma unroll
        for (int l = 0; l < BEZ_NL; ++l) D.mass_scale[l] = P.dr_mass[(size_t)e * BEZ_NL + l];
      }
    }
    const bool last_only = (P.flags & BEZ_FLAG_CF_LAST_SUBSTEP) != 0;
    for (int s = 0; s < P.substeps; ++s) {
      const bool last = (s == P.substeps - 1);
      substep<CL>(P, D, S, target, co, lds, lane, last_only ? last : true, last_only ? true : (s == 0), e);
    }
  } else if (POST && !CL) {
    co.lf = mk(co.base[(size_t)(BEZ_LFOOT_BODY * 3 + 0) * n], co.base[(size_t)(BEZ_LFOOT_BODY * 3 + 1) * n], co.base[(size_t)(BEZ_LFOOT_BODY * 3 + 2) * n]);
    co.rf = mk(co.base[(size_t)(BEZ_RFOOT_BODY * 3 + 0) * n], co.base[(size_t)(BEZ_RFOOT_BODY * 3 + 1) * n], co.base[(size_t)(BEZ_RFOOT_BODY * 3 + 2) * n]);
  }
  if (POST) {
    int64_t progress = P.progress[e], reset = P.reset[e];
    uint32_t episode = P.episode[e];
    if (!P.obs_only) {
      P.timeout[e] = (progress >= (int64_t)(P.max_len - 1)) ? 1 : 0;  // vec_task.py:331-332
      progress += 1;                                                  // kick_env.py:429
      if (reset != 0) {                                               // kick_env.py:433-435
        env_reset(P, S, target, co, episode, P.env_off + e);
        progress = 0; reset = 0;
        P.episode[e] = episode;
        if (P.task != BEZ_TASK_KICK) { st[(size_t)F_GOAL * n + e] = reset_goal(P, 0); st[(size_t)(F_GOAL + 1) * n + e] = reset_goal(P, 1); }  // walk_env.py:570-575
      }
    }
    float prev[3], feet[8], obs[BEZ_NUM_OBS], rew;
#pragma unroll
    for (int i = 0; i < 3; ++i) prev[i] = st[(size_t)(F_PREV + i) * n + e];
    const float goal_x = P.task == BEZ_TASK_KICK ? P.goal[0] : st[(size_t)F_GOAL * n + e], goal_y = P.task == BEZ_TASK_KICK ? P.goal[1] : st[(size_t)(F_GOAL + 1) * n + e];
    float cleats[24];
    if (CL) {  // the 8 cleat rows as the substeps left them in HBM (or as the caller injected them)
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        cleats[k] = co.base[(size_t)(BEZ_LCLEAT_BODY_CL * 3 + k) * n];
        cleats[12 + k] = co.base[(size_t)(BEZ_RCLEAT_BODY_CL * 3 + k) * n];
      }
    }
    env_observe_reward(P, S, co, prev, feet, obs, rew, reset, progress, goal_x, goal_y, CL ? cleats : nullptr);
#pragma unroll
    for (int i = 0; i < 3; ++i) st[(size_t)(F_PREV + i) * n + e] = prev[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) st[(size_t)(F_FEET + i) * n + e] = feet[i];
#pragma unroll
    for (int i = 0; i < BEZ_NUM_OBS; ++i)
      if (i < P.nobs) P.obs[(size_t)e * P.nobs + i] = P.obs_noise ? obs_with_noise(P, (long long)e * P.nobs + i, obs[i]) : obs[i];
    P.rew[e] = rew; P.reset[e] = reset; P.progress[e] = progress;
    if (DR && P.dr_snap && e == 0) {
      const unsigned long long f = P.dr_state->frame;
      *P.dr_snap = DrSnap{P.dr_state->noise[2], P.dr_state->noise[3], (unsigned int)f, (unsigned int)(f >> 32)};
    }
  }
  if (SIM || POST) {
    store_state(st, n, e, S);
    if (!CL) {  // with cleats the foot rows live in HBM throughout (no in-place filter: kick_env.py:467-495)
      cf_store(co, BEZ_LFOOT_BODY, co.lf);
      cf_store(co, BEZ_RFOOT_BODY, co.rf);
    }
#pragma unroll
    for (int j = 0; j < BEZ_ND; ++j) st[(size_t)(F_TARGET + j) * n + e] = target[j];
  }
}

}  // namespace bez

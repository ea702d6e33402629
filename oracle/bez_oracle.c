/* bez_oracle.c -- CPU ORACLE for the bez_kick hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker / reported CPU baseline -- the product path (libbez_sim.so, HIP)
 * never links, loads or falls back to it.
 *
 * What it restates (reference file:line):
 *   step orchestration        bez_isaacgym/tasks/base/vec_task.py:303-349
 *   buffers                   bez_isaacgym/tasks/base/vec_task.py:226-249
 *   PD targets                bez_isaacgym/tasks/kick_env.py:410-419
 *   post-physics ordering     bez_isaacgym/tasks/kick_env.py:426-438
 *   reset                     bez_isaacgym/tasks/kick_env.py:779-850
 *   quaternion_to_matrix      bez_isaacgym/tasks/kick_env.py:857-885   (quirk Q2: wxyz formula on xyzw data)
 *   compute_imu               bez_isaacgym/tasks/kick_env.py:888-930   (quirk Q1: prev aliases current)
 *   compute_off_orn           bez_isaacgym/tasks/kick_env.py:933-962
 *   feet sensors (no cleats)  bez_isaacgym/tasks/kick_env.py:966-1040  (quirk Q3, in-place noise filter)
 *   compute_bez_reward        bez_isaacgym/tasks/kick_env.py:1198-1395
 *   compute_bez_observations  bez_isaacgym/tasks/kick_env.py:1398-1417
 * PARITY PINNING: the obs/reward/reset/target arithmetic above is pinned against golden vectors
 * produced by the reference's own TorchScript functions (tests/golden/, generator committed).
 * The rigid-body step (gym.simulate, vec_task.py:324) is Isaac Gym / PhysX, a closed binary that is
 * not under /root/reference and cannot run here: for that part there is nothing to pin against --
 * "parity unpinned".  The physics below is this build's own model (Featherstone ABA, implicit PD,
 * implicit spring-damper contact), written here in plain textbook form (dense 6x6
 * spatial algebra, runtime tree tables, general Rodrigues rotations) so that it is an independent
 * check of the hand-specialised fp32 HIP kernels, and validated itself against an independent
 * numpy CRBA/RNEA and physical invariants (tests/test_oracle_physics.py).
 *
 * Build: gcc -O2 -fPIC -shared [-DBEZ_REAL=float] [-fopenmp] -o libbez_oracle_{f64,f32}.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../bez_isaacgym_amd/csrc/bez_model_gen.h"
#include "../include/bez_sim.h"
#include "bez_oracle_shapes.inc"

#ifndef BEZ_REAL
#define BEZ_REAL double
#endif
typedef BEZ_REAL real;

#define NL BEZ_NL
#define ND BEZ_ND
#define NBMAX BEZ_NBE_MAX

/* model variant: default asset or the cleats asset (BEZ_FLAG_CLEATS) */
static int m_cl(const BezSimConfig* c) { return (c->flags & BEZ_FLAG_CLEATS) != 0; }
static int m_nb(const BezSimConfig* c) { return m_cl(c) ? BEZ_NB_CL : BEZ_NB; }              /* robot bodies; the ball row follows */
static int m_has_ball(const BezSimConfig* c) { return c->task == BEZ_TASK_KICK; }
static int m_nbe(const BezSimConfig* c) { return m_nb(c) + (m_has_ball(c) ? 1 : 0); }        /* rows of the exported tensors */
static int m_nobs(const BezSimConfig* c) { return c->task == BEZ_TASK_KICK ? BEZ_NUM_OBS : BEZ_NUM_OBS_WALK; }
static int m_link_body(const BezSimConfig* c, int l) { return m_cl(c) ? BEZ_LINK_BODY_CL[l] : BEZ_LINK_BODY[l]; }
static real m_mass(const BezSimConfig* c, int l) { return (real)(m_cl(c) ? BEZ_LINK_MASS_CL[l] : BEZ_LINK_MASS[l]); }
static const double* m_com(const BezSimConfig* c, int l) { return m_cl(c) ? BEZ_LINK_COM_CL[l] : BEZ_LINK_COM[l]; }
static const double* m_inertia(const BezSimConfig* c, int l) { return m_cl(c) ? BEZ_LINK_INERTIA_CL[l] : BEZ_LINK_INERTIA[l]; }
/* box asset (BEZ_FLAG_BOX_ASSET, asset.stl: False, kick_env.py:266-276): soccerbot_box.urdf -- the default asset's dynamics with
 * the URDF's own torso / head / forearm collision boxes (upper-body ground points, ball <-> torso box) */
static int m_box(const BezSimConfig* c) { return (c->flags & BEZ_FLAG_BOX_ASSET) != 0; }
/* foot / cleat points (the first 8): the stl assets'; upper-body points: the box asset's, with and without cleats */
static const double* m_pt_pos(const BezSimConfig* c, int i) { return (m_box(c) && i >= 8) ? BEZ_PT_POS_BOX[i] : (m_cl(c) ? BEZ_PT_POS_CL[i] : BEZ_PT_POS[i]); }
/* soccerbot_box_sensor.urdf (box + cleats) moves one joint origin: link BEZ_BOXCL_LINK sits at z = BEZ_BOXCL_LINK_Z in its parent */
static double m_link_z(const BezSimConfig* c, int l) { return (l == BEZ_BOXCL_LINK && m_box(c) && m_cl(c)) ? BEZ_BOXCL_LINK_Z : BEZ_LINK_XYZ[l][2]; }
static const double* m_box_center(const BezSimConfig* c, int b) { return (b == BEZ_TORSO_BOX && m_box(c)) ? BEZ_TORSO_BOX_CENTER_BOX : BEZ_BOX_CENTER[b]; }
static const double* m_box_half(const BezSimConfig* c, int b) { return (b == BEZ_TORSO_BOX && m_box(c)) ? BEZ_TORSO_BOX_HALF_BOX : BEZ_BOX_HALF[b]; }
static int m_pt_body(const BezSimConfig* c, int i) { return m_cl(c) ? BEZ_PT_BODY_CL[i] : BEZ_PT_BODY[i]; }
static int m_body_link(const BezSimConfig* c, int b) { return m_cl(c) ? BEZ_BODY_LINK_CL[b] : BEZ_BODY_LINK[b]; }
static const double* m_body_offset(const BezSimConfig* c, int b) { return m_cl(c) ? BEZ_BODY_OFFSET_CL[b] : BEZ_BODY_OFFSET[b]; }

/* ------------------------------------------------------------------ small linear algebra */
typedef struct { real v[3]; } V3;
typedef struct { real m[3][3]; } M3;
typedef struct { real v[6]; } SV;       /* spatial vector [ang; lin] */
typedef struct { real m[6][6]; } M6;

static V3 v3(real x, real y, real z) { V3 r = {{x, y, z}}; return r; }
static V3 v3add(V3 a, V3 b) { return v3(a.v[0] + b.v[0], a.v[1] + b.v[1], a.v[2] + b.v[2]); }
static V3 v3sub(V3 a, V3 b) { return v3(a.v[0] - b.v[0], a.v[1] - b.v[1], a.v[2] - b.v[2]); }
static V3 v3scale(V3 a, real s) { return v3(a.v[0] * s, a.v[1] * s, a.v[2] * s); }
static real v3dot(V3 a, V3 b) { return a.v[0] * b.v[0] + a.v[1] * b.v[1] + a.v[2] * b.v[2]; }
static V3 v3cross(V3 a, V3 b) {
  return v3(a.v[1] * b.v[2] - a.v[2] * b.v[1], a.v[2] * b.v[0] - a.v[0] * b.v[2], a.v[0] * b.v[1] - a.v[1] * b.v[0]);
}
static V3 m3mulv(const M3* A, V3 x) {
  V3 r;
  for (int i = 0; i < 3; ++i) r.v[i] = A->m[i][0] * x.v[0] + A->m[i][1] * x.v[1] + A->m[i][2] * x.v[2];
  return r;
}
static V3 m3Tmulv(const M3* A, V3 x) {
  V3 r;
  for (int i = 0; i < 3; ++i) r.v[i] = A->m[0][i] * x.v[0] + A->m[1][i] * x.v[1] + A->m[2][i] * x.v[2];
  return r;
}
static M3 m3mul(const M3* A, const M3* B) {
  M3 C;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      real s = 0;
      for (int k = 0; k < 3; ++k) s += A->m[i][k] * B->m[k][j];
      C.m[i][j] = s;
    }
  return C;
}
static M3 m3T(const M3* A) {
  M3 C;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C.m[i][j] = A->m[j][i];
  return C;
}
static M3 skew(V3 a) {
  M3 S = {{{0, -a.v[2], a.v[1]}, {a.v[2], 0, -a.v[0]}, {-a.v[1], a.v[0], 0}}};
  return S;
}
/* proper rotation matrix of an xyzw unit quaternion (body -> world) */
static M3 quat_to_mat(const real q[4]) {
  real x = q[0], y = q[1], z = q[2], w = q[3];
  M3 R = {{{1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)},
           {2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)},
           {2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)}}};
  return R;
}
/* Rodrigues: rotation by `ang` about unit axis a */
static M3 rot_axis(V3 a, real ang) {
  real c = cos(ang), s = sin(ang), t = 1 - c;
  M3 K = skew(a);
  M3 R;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R.m[i][j] = (i == j ? c : 0) + t * a.v[i] * a.v[j] + s * K.m[i][j];
  return R;
}
static void mat_to_quat(const M3* R, real q[4]) { /* xyzw */
  real tr = R->m[0][0] + R->m[1][1] + R->m[2][2];
  if (tr > 0) {
    real s = sqrt(tr + 1) * 2;
    q[3] = s / 4; q[0] = (R->m[2][1] - R->m[1][2]) / s; q[1] = (R->m[0][2] - R->m[2][0]) / s; q[2] = (R->m[1][0] - R->m[0][1]) / s;
  } else if (R->m[0][0] > R->m[1][1] && R->m[0][0] > R->m[2][2]) {
    real s = sqrt(1 + R->m[0][0] - R->m[1][1] - R->m[2][2]) * 2;
    q[3] = (R->m[2][1] - R->m[1][2]) / s; q[0] = s / 4; q[1] = (R->m[0][1] + R->m[1][0]) / s; q[2] = (R->m[0][2] + R->m[2][0]) / s;
  } else if (R->m[1][1] > R->m[2][2]) {
    real s = sqrt(1 + R->m[1][1] - R->m[0][0] - R->m[2][2]) * 2;
    q[3] = (R->m[0][2] - R->m[2][0]) / s; q[0] = (R->m[0][1] + R->m[1][0]) / s; q[1] = s / 4; q[2] = (R->m[1][2] + R->m[2][1]) / s;
  } else {
    real s = sqrt(1 + R->m[2][2] - R->m[0][0] - R->m[1][1]) * 2;
    q[3] = (R->m[1][0] - R->m[0][1]) / s; q[0] = (R->m[0][2] + R->m[2][0]) / s; q[1] = (R->m[1][2] + R->m[2][1]) / s; q[2] = s / 4;
  }
}

static SV sv(V3 a, V3 l) { SV r = {{a.v[0], a.v[1], a.v[2], l.v[0], l.v[1], l.v[2]}}; return r; }
static V3 sv_ang(SV s) { return v3(s.v[0], s.v[1], s.v[2]); }
static V3 sv_lin(SV s) { return v3(s.v[3], s.v[4], s.v[5]); }
static SV sv_add(SV a, SV b) { SV r; for (int i = 0; i < 6; ++i) r.v[i] = a.v[i] + b.v[i]; return r; }
static SV sv_scale(SV a, real s) { SV r; for (int i = 0; i < 6; ++i) r.v[i] = a.v[i] * s; return r; }
static real sv_dot(SV a, SV b) { real s = 0; for (int i = 0; i < 6; ++i) s += a.v[i] * b.v[i]; return s; }
static SV m6mulv(const M6* A, SV x) {
  SV r;
  for (int i = 0; i < 6; ++i) { real s = 0; for (int j = 0; j < 6; ++j) s += A->m[i][j] * x.v[j]; r.v[i] = s; }
  return r;
}
/* motion cross product  V x S */
static SV crm(SV V, SV S) {
  V3 w = sv_ang(V), v = sv_lin(V), sa = sv_ang(S), sl = sv_lin(S);
  return sv(v3cross(w, sa), v3add(v3cross(w, sl), v3cross(v, sa)));
}
/* force cross product  V x* F */
static SV crf(SV V, SV F) {
  V3 w = sv_ang(V), v = sv_lin(V), n = sv_ang(F), f = sv_lin(F);
  return sv(v3add(v3cross(w, n), v3cross(v, f)), v3cross(w, f));
}
static void m6_add_outer(M6* A, SV w, real k) { /* A += k w w^T */
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) A->m[i][j] += k * w.v[i] * w.v[j];
}
/* wrench of a force f applied at point x (both about the common reference point) */
static SV wrench_at(V3 x, V3 f) { return sv(v3cross(x, f), f); }
/* spatial rigid-body inertia about the reference point: mass m, COM at c, rotational inertia Ic about COM */
static M6 rb_inertia(real m, V3 c, const M3* Ic) {
  M6 I; memset(&I, 0, sizeof(I));
  M3 cx = skew(c);
  M3 cxT = m3T(&cx);
  M3 cc = m3mul(&cx, &cxT);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      I.m[i][j] = Ic->m[i][j] + m * cc.m[i][j];
      I.m[i][j + 3] = m * cx.m[i][j];
      I.m[i + 3][j] = m * cxT.m[i][j];
      I.m[i + 3][j + 3] = (i == j) ? m : 0;
    }
  return I;
}
/* Cholesky of an SPD 6x6 in place (lower triangle); chol6_subst solves with the factor */
static int chol6_factor(M6* A) {
  for (int j = 0; j < 6; ++j) {
    real d = A->m[j][j];
    for (int k = 0; k < j; ++k) d -= A->m[j][k] * A->m[j][k];
    if (!(d > 0)) return -1;
    d = sqrt(d);
    A->m[j][j] = d;
    for (int i = j + 1; i < 6; ++i) {
      real s = A->m[i][j];
      for (int k = 0; k < j; ++k) s -= A->m[i][k] * A->m[j][k];
      A->m[i][j] = s / d;
    }
  }
  return 0;
}
static void chol6_subst(const M6* A, SV* b, int nrhs) {
  for (int r = 0; r < nrhs; ++r) {
    real* x = b[r].v;
    for (int i = 0; i < 6; ++i) { real s = x[i]; for (int k = 0; k < i; ++k) s -= A->m[i][k] * x[k]; x[i] = s / A->m[i][i]; }
    for (int i = 5; i >= 0; --i) { real s = x[i]; for (int k = i + 1; k < 6; ++k) s -= A->m[k][i] * x[k]; x[i] = s / A->m[i][i]; }
  }
}
/* solve A x = b for SPD 6x6 by Cholesky (A is destroyed) */
static int chol6_solve(M6* A, SV* b, int nrhs) {
  if (chol6_factor(A)) return -1;
  chol6_subst(A, b, nrhs);
  return 0;
}
/* general 3x3 inverse by cofactors */
static M3 m3inv(const M3* A) {
  const real (*a)[3] = A->m;
  real c00 = a[1][1] * a[2][2] - a[1][2] * a[2][1], c01 = a[1][2] * a[2][0] - a[1][0] * a[2][2], c02 = a[1][0] * a[2][1] - a[1][1] * a[2][0];
  real det = a[0][0] * c00 + a[0][1] * c01 + a[0][2] * c02;
  real id = 1 / det;
  M3 R = {{{c00 * id, (a[0][2] * a[2][1] - a[0][1] * a[2][2]) * id, (a[0][1] * a[1][2] - a[0][2] * a[1][1]) * id},
           {c01 * id, (a[0][0] * a[2][2] - a[0][2] * a[2][0]) * id, (a[0][2] * a[1][0] - a[0][0] * a[1][2]) * id},
           {c02 * id, (a[0][1] * a[2][0] - a[0][0] * a[2][1]) * id, (a[0][0] * a[1][1] - a[0][1] * a[1][0]) * id}}};
  return R;
}

/* ------------------------------------------------------------------ Philox4x32-10 (reset noise) */
static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
/* k-th uniform in [0,1) of the reset draw of (global env id, episode) */
static float reset_uniform(uint64_t seed, int64_t genv, uint32_t episode, int k) {
  uint32_t c[4] = {(uint32_t)genv, (uint32_t)((uint64_t)genv >> 32), episode, (uint32_t)(k >> 2)};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  return (float)(c[k & 3] >> 8) * (1.0f / 16777216.0f);
}

/* bez_walk / bez_orient draw a goal in reset_idx and give THE FIRST SAMPLE to every env reset by that call
 * (`self.goal[env_ids, 0] = goal_x[0]`, walk_env.py:570-575): one draw per reset call, keyed by (seed, call counter, kind)
 * -- kind 0 = the reset inside post_physics_step of that step, 1 = an explicit reset_idx call. */
static void goal_draw(uint64_t seed, uint64_t counter, uint32_t kind, float out[2]) {
  uint32_t c[4] = {(uint32_t)counter, (uint32_t)(counter >> 32), 0x474f414cu, kind};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  for (int k = 0; k < 2; ++k) out[k] = fmaf(4.0f, (float)(c[k] >> 8) * (1.0f / 16777216.0f), -2.0f); /* torch_rand_float(-2, 2) */
}

/* ------------------------------------------------------------------ per-env state */
typedef struct {
  real root_pos[3], root_quat[4], root_lin[3], root_ang[3];
  real q[ND], qd[ND];
  real ball_pos[3], ball_quat[4], ball_lin[3], ball_ang[3];
  real target[ND];
  real prev_lin_vel[3];
  real contact_force[NBMAX][3]; /* per body (robot bodies, then the ball), world */
  real goal[2];                 /* bez_walk: per-env goal, redrawn at reset (walk_env.py:570-575) */
  real feet[8];
  real obs[BEZ_NUM_OBS];
  real rew;
  int64_t reset, progress, timeout, randomize;
  uint32_t episode;
  /* DR */
  real friction, kp_scale[ND], kd_scale[ND], mass_scale[NL], gravity[3], lim_lo[ND], lim_hi[ND];
  real ftrans[ND]; /* BEZ_FLAG_TGS_SOLVER: size of the spatial force each joint transmitted in the previous substep */
  real vmargin;    /* test aid: smallest distance of a speed-limit predictor from its decision boundary over the last control step, rad/s */
} Env;

typedef struct {
  BezSimConfig cfg;
  int n;
  Env* env;
  int64_t obs_calls; /* number of compute_observations passes so far (quirk Q1 bookkeeping) */
  uint64_t post_calls, reset_calls; /* keys of the shared goal draw (bez_walk / bez_orient) */
  /* device-side domain randomisation restated (bez_sim_set_randomization): vec_task.py:505-725 */
  int dr_on; BezDrConfig dr; uint64_t dr_frame, dr_last_rand; float dr_noise[4];
} Oracle;

/* kinematics of one env: link frames relative to O = root_pos, world axes */
typedef struct {
  M3 E[NL];  /* link -> world rotation */
  V3 r[NL];  /* link origin relative to O */
  V3 a[NL];  /* joint axis in world */
} Kin;

static void forward_kinematics(const BezSimConfig* c, const Env* e, Kin* k) {
  k->E[0] = quat_to_mat(e->root_quat);
  k->r[0] = v3(0, 0, 0);
  k->a[0] = v3(0, 0, 0);
  for (int l = 1; l < NL; ++l) {
    int p = BEZ_LINK_PARENT[l];
    V3 ax = v3((real)BEZ_LINK_AXIS_VEC[l][0], (real)BEZ_LINK_AXIS_VEC[l][1], (real)BEZ_LINK_AXIS_VEC[l][2]);
    V3 t = v3((real)BEZ_LINK_XYZ[l][0], (real)BEZ_LINK_XYZ[l][1], (real)m_link_z(c, l));
    k->r[l] = v3add(k->r[p], m3mulv(&k->E[p], t));
    M3 Rj = rot_axis(ax, e->q[l - 1]);
    k->E[l] = m3mul(&k->E[p], &Rj);
    k->a[l] = m3mulv(&k->E[p], ax);
  }
}

/* Result of one dynamics evaluation */
typedef struct {
  SV a0;         /* root spatial acceleration about O */
  real qdd[ND];
  V3 ball_lin_acc, ball_ang_acc; /* classical */
  real contact_force[NBMAX][3];
  real vmargin;
} Dyn;

typedef struct { int link, body; V3 x; real fn0, kn, ct, ftx0, fty0; } GroundHit;

/* Adds the implicit ground contact of a point at x (rel. O) on a body with velocity V (about O) whose
 * world height is z.  Returns 1 if active and fills hit.  IA/pA are the body's inertia and bias. */
static real m_ball_kn(const BezSimConfig* c) { return c->ball_kn > 0 ? (real)c->ball_kn : (real)c->contact_kn; }
static real m_ball_cn(const BezSimConfig* c) { return c->ball_cn > 0 ? (real)c->ball_cn : (real)c->contact_cn; }
static int ground_contact(const BezSimConfig* c, real ckn, real ccn, real mu, real h, V3 x, real z, SV V, M6* IA, SV* pA, GroundHit* hit) {
  real d = -z;
  if (!(d > 0)) return 0;
  V3 vp = v3add(sv_lin(V), v3cross(sv_ang(V), x));
  real kd = h * ckn + ccn;
  real fn0 = ckn * d - kd * vp.v[2];
  if (!(fn0 > 0)) return 0;
  real kn = h * kd;
  SV wn = wrench_at(x, v3(0, 0, 1));
  m6_add_outer(IA, wn, kn);
  *pA = sv_add(*pA, sv_scale(wn, -fn0));
  real vt = sqrt(vp.v[0] * vp.v[0] + vp.v[1] * vp.v[1]);
  real ct = mu * fn0 / fmax(vt, (real)c->contact_veps);
  if (ct > (real)c->contact_ct) ct = (real)c->contact_ct;
  real kt = h * ct;
  SV wx = wrench_at(x, v3(1, 0, 0)), wy = wrench_at(x, v3(0, 1, 0));
  real ftx0 = -ct * vp.v[0], fty0 = -ct * vp.v[1];
  m6_add_outer(IA, wx, kt);
  m6_add_outer(IA, wy, kt);
  *pA = sv_add(*pA, sv_scale(wx, -ftx0));
  *pA = sv_add(*pA, sv_scale(wy, -fty0));
  hit->x = x; hit->fn0 = fn0; hit->kn = kn; hit->ct = ct; hit->ftx0 = ftx0; hit->fty0 = fty0;
  return 1;
}


/* ---- leg <-> leg self-collision (kick_env.py:365-366: create_actor(..., collision_filter 0) enables it).
 * Each leg box is a capsule (bez_model_gen.h: BEZ_CAP_*); every left x right pair of BEZ_CPAIR is tested by the
 * closest points of the two segments.  A penetrating pair is a spring-damper + regularised Coulomb point contact with
 * equal and opposite forces on the two links (momentum is conserved exactly).  A contact between two links of the same
 * tree closes a kinematic loop that the ABA recursion cannot fold in, so the pairs' forces are evaluated from the state
 * at the start of the substep -- lambda0 = k (depth - h u_n) - c u_n, the spring at the gap the CURRENT closing speed
 * leads to -- and then scaled by ONE factor per env that stands for the implicit part (self_contact_scale below). */
static void segment_closest(V3 p1, V3 q1, V3 p2, V3 q2, V3* c1, V3* c2) {
  V3 d1 = v3sub(q1, p1), d2 = v3sub(q2, p2), r = v3sub(p1, p2);
  real a = v3dot(d1, d1), e = v3dot(d2, d2), f = v3dot(d2, r);
  real c = v3dot(d1, r), b = v3dot(d1, d2);
  real den = a * e - b * b; /* >= 0; segments here never degenerate to points (a, e > 0) */
  real s = den > (real)1e-12 ? (b * f - c * e) / den : 0;
  if (s < 0) s = 0;
  if (s > 1) s = 1;
  real t = (b * s + f) / e;
  if (t < 0) { t = 0; s = -c / a; if (s < 0) s = 0; if (s > 1) s = 1; }
  else if (t > 1) { t = 1; s = (b - c) / a; if (s < 0) s = 0; if (s > 1) s = 1; }
  *c1 = v3add(p1, v3scale(d1, s));
  *c2 = v3add(p2, v3scale(d2, t));
}
/* (Tried in round 6 and removed: testing the pairs inside physx.contact_offset, bez_kick.yaml:139, 2 cm before the capsules touch, with the same
 * look-ahead law -- box penetration along the reference policy's rollouts 25.8 -> 24.1 mm, but default-yaml training over eight seeds fell from
 * 26 / 34 / 1 / 23 / 32 / 37 / 32 / 21 to 35 / 3 / 9 / 1 / 17 / 2 / 21 / 18: profiles/r06_seed_table_contact_offset.txt.) */
static void self_collision(const BezSimConfig* c, real hs, real mu, const Kin* k, const SV* V, SV* pS, real cf[][3], int with_fric, real* f2) {
  /* hs = the substep the spring looks ahead by (0: the round-5 explicit law, kept for the oracle-only solver families) */
  for (int pr = 0; pr < BEZ_NCPAIR; ++pr) {
    const int ia = BEZ_CPAIR[pr][0], ib = BEZ_CPAIR[pr][1];
    const int la = BEZ_CAP_LINK[ia], lb = BEZ_CAP_LINK[ib];
    V3 a0 = v3add(k->r[la], m3mulv(&k->E[la], v3((real)BEZ_CAP_P0[ia][0], (real)BEZ_CAP_P0[ia][1], (real)BEZ_CAP_P0[ia][2])));
    V3 a1 = v3add(k->r[la], m3mulv(&k->E[la], v3((real)BEZ_CAP_P1[ia][0], (real)BEZ_CAP_P1[ia][1], (real)BEZ_CAP_P1[ia][2])));
    V3 b0 = v3add(k->r[lb], m3mulv(&k->E[lb], v3((real)BEZ_CAP_P0[ib][0], (real)BEZ_CAP_P0[ib][1], (real)BEZ_CAP_P0[ib][2])));
    V3 b1 = v3add(k->r[lb], m3mulv(&k->E[lb], v3((real)BEZ_CAP_P1[ib][0], (real)BEZ_CAP_P1[ib][1], (real)BEZ_CAP_P1[ib][2])));
    V3 ca, cb;
    segment_closest(a0, a1, b0, b1, &ca, &cb);
    V3 dl = v3sub(ca, cb);
    real d2 = v3dot(dl, dl);
    real rs = (real)BEZ_CAP_R[ia] + (real)BEZ_CAP_R[ib] + ((c->flags & BEZ_FLAG_HARD_CONTACT) ? 2 * (real)c->tune[6] : 0); /* EXPERIMENT: shape rest offsets */
    if (!(d2 < rs * rs) || !(d2 > (real)1e-12)) continue;
    real dist = sqrt(d2), depth = rs - dist;
    V3 n = v3scale(dl, 1 / dist);                                         /* from capsule b towards capsule a */
    V3 x = v3add(cb, v3scale(n, (real)BEZ_CAP_R[ib] - (real)0.5 * depth)); /* mid-point of the overlap, rel. O */
    V3 va = v3add(sv_lin(V[la]), v3cross(sv_ang(V[la]), x)), vb = v3add(sv_lin(V[lb]), v3cross(sv_ang(V[lb]), x));
    V3 u = v3sub(va, vb);
    real un = v3dot(u, n);
    real fmag = (real)c->self_kn * depth - (hs * (real)c->self_kn + (real)c->self_cn) * un;
    if (!(fmag > 0)) continue;
    V3 ut = v3sub(u, v3scale(n, un));
    real vt = sqrt(v3dot(ut, ut));
    real ct = mu * fmag / fmax(vt, (real)c->contact_veps);
    if (ct > (real)c->self_cn) ct = (real)c->self_cn; /* explicit: the stick viscosity is capped like the normal damper */
    V3 fn = v3scale(n, fmag);
    V3 f = v3add(fn, v3scale(ut, -ct));                                   /* force on link la; -f on link lb */
    pS[la] = sv_add(pS[la], sv_scale(wrench_at(x, f), -1));
    pS[lb] = sv_add(pS[lb], wrench_at(x, f));
    if (f2) *f2 += v3dot(f, f);
    V3 fr = with_fric ? f : fn;
    for (int i = 0; i < 3; ++i) { cf[m_link_body(c, la)][i] += fr.v[i]; cf[m_link_body(c, lb)][i] -= fr.v[i]; }
  }
}

/* The implicit part of the leg <-> leg contact (round 6).  The wrench pair of a contact between two links of the same tree does no work
 * on the torso: it is exactly the joint torques tau_d = J_d lambda on the joints between the two links (J_d = d gap / d q_d).  An implicit
 * spring-damper would be lambda = lambda0 - K (gap acceleration), K = h^2 k + h c, the gap acceleration being that of the solved step --
 * which pass 2 cannot know.  What IS known once pass 2 has run, per joint: g_d = 1/D_d, the held-parent acceleration qdd_hp_d the
 * saturation and speed-limit predictors use, and tau_d = S_d . (sum of the contact wrenches W_l on the links of joint d's subtree).  With
 * every pair force scaled by the SAME s (equal and opposite pairs stay equal and opposite: momentum is conserved whatever s is), the
 * lambda0-weighted implicit equation  s F2 = F2 - K (A_M + s A_S)  has the joint-space-diagonal estimates
 *     A_M = sum_d tau_d qdd_hp_d            (how fast the gap would close with the contact switched off),
 *     A_S = sum_d tau_d^2 g_d               (how fast the contact's own force opens it; F2 = sum of the pairs' |force|^2),
 * and  s = (F2 - K A_M) / (F2 + BEZ_SELF_IMPLICIT K A_S),  clamped to [0, 8].  Exact for one pair acting on joints that do not recoil on
 * each other; BEZ_SELF_IMPLICIT = 2 covers the recoil the diagonal leaves out (J M^-1 J^T against its diagonal part).  The scale is known
 * BEFORE the root solve, so the scaled wrenches enter as the exact linear correction of pass 2's bias recursion they always were. */
#define BEZ_SELF_IMPLICIT 2.0
static real self_contact_scale(const BezSimConfig* c, real h, const SV* S, const SV* W, real F2, const real* g, const real* qdd_hp) {
  SV Wsub[NL]; memcpy(Wsub, W, sizeof(Wsub));
  real AM = 0, AS = 0;
  for (int l = NL - 1; l >= 1; --l) {
    real t = sv_dot(S[l], Wsub[l]);
    AS += t * t * g[l]; AM += t * qdd_hp[l];
    Wsub[BEZ_LINK_PARENT[l]] = sv_add(Wsub[BEZ_LINK_PARENT[l]], Wsub[l]);
  }
  if (!(F2 > 0)) return 0;
  const real K = h * h * (real)c->self_kn + h * (real)c->self_cn;
  real sc = (F2 - K * AM) / (F2 + (real)BEZ_SELF_IMPLICIT * K * AS);
  if (!(sc > 0)) sc = 0;
  if (sc > 8) sc = 8;
  return sc;
}

/* ---- same-leg calf <-> foot-plate contact (BEZ_FLAG_ANKLE_STOP; kick_env.py:365-366: collision_filter 0 collides every non-adjacent
 * shape pair, soccerbot_stl.urdf:232-236 calf box, :272-276 foot plate).  Each bottom corner of the calf box is tested against the plane of
 * the plate's top face; gap g = n . (corner - foot origin) - z_top with n the foot's z axis.  The two bodies are joined by the ankle-pitch and
 * foot-roll joints only, so the contact force pair does work through those two joint rates alone:  g' = J_a qd_a + J_f qd_f  with
 * J_d = -n . (S_d at the corner), and the contact is the joint-space force tau_d = J_d lambda,  lambda = -k (g + h g') - c g' - (h^2 k + h c) sum_e J_e qdd_e
 * (the implicit spring-damper of the joint limits; the own-joint term goes into the joint's D, the cross term is dropped). */
typedef struct { int n; int la[8], lf[8]; real Ja[8], Jf[8], lam0[8]; V3 nrm[8]; } AnkleStop;
static real m_stop_kn(const BezSimConfig* c) { return c->tune[0] != 0 ? (real)c->tune[0] : (real)2e5; }
static real m_stop_cn(const BezSimConfig* c) { return c->tune[1] != 0 ? (real)c->tune[1] : (real)1e3; }
static void ankle_stop(const BezSimConfig* c, const Env* e, real h, const Kin* k, const SV* S, AnkleStop* A) {
  A->n = 0;
  const real kn = m_stop_kn(c), cn = m_stop_cn(c);
  for (int side = 0; side < 2; ++side) {
    const int bc = 2 + 5 * side, bf = 4 + 5 * side;              /* calf box, foot box */
    const int lc = BEZ_BOX_LINK[bc], lf = BEZ_BOX_LINK[bf], la = lf - 1;
    const real ztop = (real)(BEZ_BOX_CENTER[bf][2] + BEZ_BOX_HALF[bf][2]);
    V3 n = v3(k->E[lf].m[0][2], k->E[lf].m[1][2], k->E[lf].m[2][2]);
    for (int cx = -1; cx <= 1; cx += 2) for (int cy = -1; cy <= 1; cy += 2) {
      V3 pl = v3((real)(BEZ_BOX_CENTER[bc][0] + cx * BEZ_BOX_HALF[bc][0]), (real)(BEZ_BOX_CENTER[bc][1] + cy * BEZ_BOX_HALF[bc][1]),
                 (real)(BEZ_BOX_CENTER[bc][2] - BEZ_BOX_HALF[bc][2]));
      V3 x = v3add(k->r[lc], m3mulv(&k->E[lc], pl));
      real g = v3dot(n, v3sub(x, k->r[lf])) - ztop;
      if (!(g < 0)) continue;
      real Ja = -v3dot(n, v3add(sv_lin(S[la]), v3cross(sv_ang(S[la]), x)));
      real Jf = -v3dot(n, v3add(sv_lin(S[lf]), v3cross(sv_ang(S[lf]), x)));
      real gd = Ja * e->qd[la - 1] + Jf * e->qd[lf - 1];
      real lam0 = -kn * (g + h * gd) - cn * gd;
      if (!(lam0 > 0)) continue;
      int i = A->n++;
      A->la[i] = la; A->lf[i] = lf; A->Ja[i] = Ja; A->Jf[i] = Jf; A->lam0[i] = lam0; A->nrm[i] = n;
    }
  }
}

/* pass 1 of the ABA: kinematics, link velocities, bias accelerations, rigid-body inertias and bias forces (gravity included) */
typedef struct { Kin k; SV V[NL], S[NL], cb[NL], pA[NL]; M6 IA[NL]; } Pass1;
static void aba_pass1(const BezSimConfig* c, const Env* e, Pass1* P) {
  Kin* k = &P->k;
  SV *V = P->V, *S = P->S, *cb = P->cb, *pA = P->pA;
  M6* IA = P->IA;
  forward_kinematics(c, e, k);
  V3 g = v3(e->gravity[0], e->gravity[1], e->gravity[2]);
  V[0] = sv(v3(e->root_ang[0], e->root_ang[1], e->root_ang[2]), v3(e->root_lin[0], e->root_lin[1], e->root_lin[2]));
  memset(&S[0], 0, sizeof(SV));
  memset(&cb[0], 0, sizeof(SV));
  for (int l = 0; l < NL; ++l) {
    if (l > 0) {
      int p = BEZ_LINK_PARENT[l];
      S[l] = sv(k->a[l], v3cross(k->r[l], k->a[l]));
      SV vj = sv_scale(S[l], e->qd[l - 1]);
      V[l] = sv_add(V[p], vj);
      cb[l] = crm(V[l], vj);
    }
    real m = m_mass(c, l) * e->mass_scale[l];
    V3 cl = v3((real)m_com(c, l)[0], (real)m_com(c, l)[1], (real)m_com(c, l)[2]);
    V3 cw = v3add(k->r[l], m3mulv(&k->E[l], cl));
    const double* il = m_inertia(c, l);
    M3 Il = {{{(real)il[0], (real)il[3], (real)il[4]}, {(real)il[3], (real)il[1], (real)il[5]}, {(real)il[4], (real)il[5], (real)il[2]}}};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Il.m[i][j] *= e->mass_scale[l];
    M3 ET = m3T(&k->E[l]);
    M3 tmp = m3mul(&k->E[l], &Il);
    M3 Iw = m3mul(&tmp, &ET);
    IA[l] = rb_inertia(m, cw, &Iw);
    SV hmom = m6mulv(&IA[l], V[l]);
    pA[l] = crf(V[l], hmom);
    SV fg = wrench_at(cw, v3scale(g, m));
    pA[l] = sv_add(pA[l], sv_scale(fg, -1));
  }
}

/* One evaluation of the build's dynamics model for one env at substep size h.
 * `mode` 0 = full model; 1 = bare ABA (no PD / friction / limits / contact / armature: used by the
 * known-answer tests, with tau_in as the applied joint torques). */
/* force_lock != NULL: the speed-limit locks are given (+1 / -1 / 0 per joint) instead of predicted: the active-set reference (dynamics) */
static void dynamics_x(const BezSimConfig* c, const Env* e, real h, int mode, const real* tau_in, Dyn* out, const int* force_lock) {
  Pass1 P1;
  aba_pass1(c, e, &P1);
  const Kin k = P1.k;
  const int nb = m_nb(c); /* row of the ball */
  V3 g = v3(e->gravity[0], e->gravity[1], e->gravity[2]);
  SV *V = P1.V, *S = P1.S, *cb = P1.cb, *pA = P1.pA;
  SV pS[NL]; /* leg<->leg contact wrenches (as bias forces), propagated next to pA: the predictors below do not see them */
  real F2self = 0; /* sum over the pairs of |force|^2 */
  SV Wself[NL]; real cfs[NBMAX][3]; /* the same as wrenches ON the links, and the contact-force rows they report: both scaled by self_contact_scale */
  M6* IA = P1.IA;
  memset(pS, 0, sizeof(pS)); memset(Wself, 0, sizeof(Wself)); memset(cfs, 0, sizeof(cfs));
  memset(out->contact_force, 0, sizeof(out->contact_force));
  out->vmargin = (real)1e9;

  /* contacts (implicit spring-dampers folded into IA / pA) */
  GroundHit hits[BEZ_NPT + BEZ_NXPT];
  int nhit = 0;
  real mu = e->friction;
  /* ball as a free body about its own centre, classical accelerations */
  M6 Mb; memset(&Mb, 0, sizeof(Mb));
  SV pb; memset(&pb, 0, sizeof(pb));
  GroundHit bhit; int ball_ground = 0;
  int bl_link = -1; V3 bl_x = v3(0, 0, 0), bl_xb = v3(0, 0, 0), bl_n = v3(0, 0, 0); M3 bl_A; V3 bl_f0p = v3(0, 0, 0);
  memset(&bl_A, 0, sizeof(bl_A));
  if (mode == 0) {
    for (int i = 0; i < BEZ_NPT; ++i) {
      int l = BEZ_PT_LINK[i];
      V3 pl = v3((real)m_pt_pos(c, i)[0], (real)m_pt_pos(c, i)[1], (real)m_pt_pos(c, i)[2]);
      V3 x = v3add(k.r[l], m3mulv(&k.E[l], pl));
      real z = e->root_pos[2] + x.v[2];
      if (ground_contact(c, (real)c->contact_kn, (real)c->contact_cn, mu, h, x, z, V[l], &IA[l], &pA[l], &hits[nhit])) { hits[nhit].link = l; hits[nhit].body = m_pt_body(c, i); ++nhit; }
    }
    if ((c->flags & BEZ_FLAG_ALL_GROUND_SHAPES) && !m_cl(c) && !m_box(c)) { /* EXPERIMENT (get-up scenarios): the corners of every other collision shape */
      for (int i = 0; i < BEZ_NXPT; ++i) {
        int l = BEZ_XPT_LINK[i];
        V3 x = v3add(k.r[l], m3mulv(&k.E[l], v3((real)BEZ_XPT_POS[i][0], (real)BEZ_XPT_POS[i][1], (real)BEZ_XPT_POS[i][2])));
        real z = e->root_pos[2] + x.v[2];
        if (ground_contact(c, (real)c->contact_kn, (real)c->contact_cn, mu, h, x, z, V[l], &IA[l], &pA[l], &hits[nhit])) { hits[nhit].link = l; hits[nhit].body = BEZ_XPT_BODY[i]; ++nhit; }
      }
    }
    if (!(c->flags & BEZ_FLAG_NO_SELF_COLLISION)) {
      self_collision(c, h, mu, &k, V, pS, cfs, (c->flags & BEZ_FLAG_CF_WITH_FRICTION) != 0, &F2self);
      for (int l = 0; l < NL; ++l) Wself[l] = sv_scale(pS[l], -1);
    }
    /* ball */
    real R = (real)BEZ_BALL_RADIUS, mb = (real)BEZ_BALL_MASS, Ib = (real)BEZ_BALL_INERTIA;
    for (int i = 0; i < 3; ++i) { Mb.m[i][i] = Ib; Mb.m[i + 3][i + 3] = mb; }
    pb = sv(v3(0, 0, 0), v3scale(g, -mb));
    SV Vb = sv(v3(e->ball_ang[0], e->ball_ang[1], e->ball_ang[2]), v3(e->ball_lin[0], e->ball_lin[1], e->ball_lin[2]));
    ball_ground = ground_contact(c, m_ball_kn(c), m_ball_cn(c), mu, h, v3(0, 0, -R), e->ball_pos[2] - R, Vb, &Mb, &pb, &bhit);
    /* ball vs leg boxes: deepest penetration only */
    real best = 0;
    V3 bc = v3(e->ball_pos[0] - e->root_pos[0], e->ball_pos[1] - e->root_pos[1], e->ball_pos[2] - e->root_pos[2]); /* rel. O */
    V3 bn = v3(0, 0, 0), bP = v3(0, 0, 0);
    for (int b = 0; b < BEZ_NBOX; ++b) {
      int l = BEZ_BOX_LINK[b];
      V3 cl = v3((real)m_box_center(c, b)[0], (real)m_box_center(c, b)[1], (real)m_box_center(c, b)[2]);
      V3 he = v3((real)m_box_half(c, b)[0], (real)m_box_half(c, b)[1], (real)m_box_half(c, b)[2]);
      V3 ql = v3sub(m3Tmulv(&k.E[l], v3sub(bc, k.r[l])), cl); /* ball centre in box frame */
      V3 cp; int inside = 1;
      for (int i = 0; i < 3; ++i) {
        real t = ql.v[i];
        if (t > he.v[i]) { t = he.v[i]; inside = 0; }
        if (t < -he.v[i]) { t = -he.v[i]; inside = 0; }
        cp.v[i] = t;
      }
      V3 nl; real depth;
      if (!inside) {
        V3 dlt = v3sub(ql, cp);
        real dist = sqrt(v3dot(dlt, dlt));
        depth = R - dist;
        if (!(depth > 0)) continue;
        nl = v3scale(dlt, 1 / dist);
      } else { /* centre inside the box: push out through the nearest face */
        int ax = 0; real md = he.v[0] - fabs(ql.v[0]);
        for (int i = 1; i < 3; ++i) { real di = he.v[i] - fabs(ql.v[i]); if (di < md) { md = di; ax = i; } }
        nl = v3(0, 0, 0); nl.v[ax] = ql.v[ax] >= 0 ? 1 : -1;
        cp = ql; cp.v[ax] = nl.v[ax] * he.v[ax];
        depth = R + md;
      }
      if (depth > best) {
        best = depth; bl_link = l;
        bn = m3mulv(&k.E[l], nl);
        bP = v3add(k.r[l], m3mulv(&k.E[l], v3add(cp, cl)));
      }
    }
    if (bl_link >= 0) {
      int l = bl_link;
      V3 x = bP, xb = v3sub(bP, bc);
      V3 u = v3sub(v3add(sv_lin(V[l]), v3cross(sv_ang(V[l]), x)), v3add(sv_lin(Vb), v3cross(sv_ang(Vb), xb)));
      real un = v3dot(u, bn);
      real kd = h * m_ball_kn(c) + m_ball_cn(c);
      real fmag = m_ball_kn(c) * best + kd * un;
      if (fmag > 0) {
        V3 ut = v3sub(u, v3scale(bn, un));
        real vt = sqrt(v3dot(ut, ut));
        real ct = mu * fmag / fmax(vt, (real)c->contact_veps);
        if (ct > (real)c->contact_ct) ct = (real)c->contact_ct;
        real kn = h * kd, kt = h * ct;
        /* force on the LINK: f0 = -(fmag n + ct ut);  K = kn nn^T + kt (1 - nn^T) */
        V3 f0 = v3scale(v3add(v3scale(bn, fmag), v3scale(ut, ct)), -1);
        M3 K;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) K.m[i][j] = (kn - kt) * bn.v[i] * bn.v[j] + (i == j ? kt : 0);
        /* G = Jb Mb^-1 Jb^T, gb = Jb Mb^-1 pb  with Jb = [-[xb]x  1] */
        M6 Mc = Mb; SV rhs[4];
        for (int j = 0; j < 3; ++j) { V3 ej = v3(j == 0, j == 1, j == 2); rhs[j] = wrench_at(xb, ej); }
        rhs[3] = pb;
        chol6_solve(&Mc, rhs, 4);
        M3 G; V3 gb;
        for (int i = 0; i < 3; ++i) {
          V3 ei = v3(i == 0, i == 1, i == 2);
          SV wi = wrench_at(xb, ei);
          for (int j = 0; j < 3; ++j) G.m[i][j] = sv_dot(wi, rhs[j]);
          gb.v[i] = sv_dot(wi, rhs[3]);
        }
        M3 KG = m3mul(&K, &G);
        for (int i = 0; i < 3; ++i) KG.m[i][i] += 1;
        M3 inv = m3inv(&KG);
        bl_A = m3mul(&inv, &K);
        bl_f0p = m3mulv(&inv, v3sub(f0, m3mulv(&K, gb)));
        /* fold into the link: IA += Jl^T A Jl ; pA -= Jl^T f0' */
        SV Jc[3];
        for (int j = 0; j < 3; ++j) Jc[j] = wrench_at(x, v3(j == 0, j == 1, j == 2));
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
          for (int a = 0; a < 6; ++a) for (int b2 = 0; b2 < 6; ++b2) IA[l].m[a][b2] += Jc[i].v[a] * bl_A.m[i][j] * Jc[j].v[b2];
        pA[l] = sv_add(pA[l], sv_scale(wrench_at(x, bl_f0p), -1));
        bl_x = x; bl_xb = xb; bl_n = bn;
      } else {
        bl_link = -1;
      }
    }
  }

  /* same-leg calf <-> foot-plate contact as a coupled limit of the two ankle joints */
  AnkleStop AS; AS.n = 0;
  real stop_tau[NL], stop_k[NL];
  memset(stop_tau, 0, sizeof(stop_tau)); memset(stop_k, 0, sizeof(stop_k));
  if (mode == 0 && (c->flags & BEZ_FLAG_ANKLE_STOP)) {
    ankle_stop(c, e, h, &k, S, &AS);
    const real kimp = h * h * m_stop_kn(c) + h * m_stop_cn(c);
    for (int i = 0; i < AS.n; ++i) {
      stop_tau[AS.la[i]] += AS.Ja[i] * AS.lam0[i]; stop_k[AS.la[i]] += kimp * AS.Ja[i] * AS.Ja[i];
      stop_tau[AS.lf[i]] += AS.Jf[i] * AS.lam0[i]; stop_k[AS.lf[i]] += kimp * AS.Jf[i] * AS.Jf[i];
    }
  }
  /* pass 2: articulated inertias, leaves -> root.  Per joint the recursion keeps g = 1/D and w = (tau - S.pA)/D; pass 3 forms
   * qdd = w - g U.(a_parent + c).  A joint on its speed limit (kick_env.py:327: velocity 2 pi rad/s) is the same recursion with
   * g = 0 and w = the acceleration that puts its rate ON the limit at the end of the substep: a prescribed-rate joint, whose
   * reaction reaches the parent through pA like any other joint force.  (Round 5 clamped the rate after the step instead, which
   * took the link's momentum without reacting on anything.)  Which joints: those whose held-parent predictor -- the one the
   * drive saturation uses -- sees the rate beyond the limit.  The predictor can miss (the parent's own acceleration decides as
   * often as the joint's torque does): such a joint exceeds the limit for one substep and is caught by the next;
   * tools/vlimit_probe.py has the rates, `dynamics` below the exact reference. */
  SV U[NL]; real gj[NL], wj[NL], wS[NL], qdd_hp[NL];
  for (int l = NL - 1; l >= 1; --l) {
    int p = BEZ_LINK_PARENT[l], d = l - 1;
    U[l] = m6mulv(&IA[l], S[l]);
    real J = sv_dot(S[l], U[l]);
    real tau, D;
    int lock = 0; real qdd_fix = 0;
    const real Ucb = sv_dot(U[l], cb[l]);
    if (mode == 0) {
      J += (real)c->armature;
      real kp = (real)c->kp * e->kp_scale[d], kdm = (real)c->kd * e->kd_scale[d];
      real tau_pd0 = kp * (e->target[d] - e->q[d] - h * e->qd[d]) - kdm * e->qd[d];
      real k_pd = h * h * kp + h * kdm;
      real cf = (real)c->joint_friction / fmax(fabs(e->qd[d]), (real)c->jfric_veps);
      real k_f = h * cf, tau_f0 = -cf * e->qd[d];
      real k_l = 0, tau_l0 = 0;
      real lo = e->lim_lo[d], hi = e->lim_hi[d]; /* DR may jitter the PHYSICAL limits (bez_kick.yaml:206-219); targets keep the originals */
      if (e->q[d] < lo) { tau_l0 = (real)c->limit_k * (lo - e->q[d] - h * e->qd[d]) - (real)c->limit_d * e->qd[d]; k_l = h * h * (real)c->limit_k + h * (real)c->limit_d; }
      else if (e->q[d] > hi) { tau_l0 = (real)c->limit_k * (hi - e->q[d] - h * e->qd[d]) - (real)c->limit_d * e->qd[d]; k_l = h * h * (real)c->limit_k + h * (real)c->limit_d; }
      tau_l0 += stop_tau[l]; k_l += stop_k[l];
      /* effort-limit predictor: joint acceleration with the parent held (a_parent = 0) */
      real bias = sv_dot(S[l], pA[l]) + Ucb;
      real qdd_est = (tau_pd0 + tau_f0 + tau_l0 - bias) / (J + k_pd + k_f + k_l);
      real tau_drive = tau_pd0 - k_pd * qdd_est;
      real eff = (real)c->effort;
      if (tau_drive > eff) { tau = eff + tau_f0 + tau_l0; D = J + k_f + k_l; }
      else if (tau_drive < -eff) { tau = -eff + tau_f0 + tau_l0; D = J + k_f + k_l; }
      else { tau = tau_pd0 + tau_f0 + tau_l0; D = J + k_pd + k_f + k_l; }
      /* speed-limit predictor, the same kind: the rate the joint would have at the end of the substep with the parent held */
      real vl = (real)c->vel_limit, v_pred = e->qd[d] + h * (tau - bias) / D;
      if (d >= 2 && fabs(fabs(v_pred) - vl) < out->vmargin) out->vmargin = fabs(fabs(v_pred) - vl); /* (the head joints are never driven) */
      if (force_lock) { lock = force_lock[d] != 0; if (lock) qdd_fix = ((force_lock[d] > 0 ? vl : -vl) - e->qd[d]) / h; }
      else if (v_pred > vl) { lock = 1; qdd_fix = (vl - e->qd[d]) / h; }
      else if (v_pred < -vl) { lock = 1; qdd_fix = (-vl - e->qd[d]) / h; }
    } else {
      tau = tau_in ? tau_in[d] : 0;
      D = J;
    }
    const real u_main = tau - sv_dot(S[l], pA[l]), du = -sv_dot(S[l], pS[l]);
    if (lock) { gj[l] = 0; wj[l] = qdd_fix; qdd_hp[l] = qdd_fix; }
    else { gj[l] = 1 / D; wj[l] = u_main * gj[l]; qdd_hp[l] = (u_main - Ucb) * gj[l]; }
    wS[l] = du * gj[l];
    M6 Ia = IA[l];
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) Ia.m[i][j] -= U[l].v[i] * U[l].v[j] * gj[l];
    SV pa = sv_add(pA[l], sv_add(m6mulv(&Ia, cb[l]), sv_scale(U[l], wj[l])));
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) IA[p].m[i][j] += Ia.m[i][j];
    pA[p] = sv_add(pA[p], pa);
    pS[p] = sv_add(pS[p], sv_add(pS[l], sv_scale(U[l], wS[l])));
  }
  /* the leg<->leg forces' common scale, then their (linear) share of the joint accelerations and of the torso's bias */
  const real sc = mode == 0 ? self_contact_scale(c, h, S, Wself, F2self, gj, qdd_hp) : 0;
  for (int l = 1; l < NL; ++l) wj[l] += sc * wS[l];
  for (int b = 0; b < NBMAX; ++b) for (int i = 0; i < 3; ++i) out->contact_force[b][i] += sc * cfs[b][i];
  /* root */
  M6 I0 = IA[0];
  SV a0 = sv_scale(sv_add(pA[0], sv_scale(pS[0], sc)), -1);
  chol6_solve(&I0, &a0, 1);
  if (c->flags & BEZ_FLAG_FIX_BASE) memset(&a0, 0, sizeof(a0)); /* urdfAsset.fixBaseLink (kick_env.py:287): the torso is welded to the world */
  out->a0 = a0;
  /* pass 3 */
  SV acc[NL];
  acc[0] = a0;
  for (int l = 1; l < NL; ++l) {
    int p = BEZ_LINK_PARENT[l];
    SV ap = sv_add(acc[p], cb[l]);
    real qdd = wj[l] - gj[l] * sv_dot(U[l], ap);
    out->qdd[l - 1] = qdd;
    acc[l] = sv_add(ap, sv_scale(S[l], qdd));
  }
  /* contact forces actually applied (implicit part resolved with the link accelerations) */
  out->ball_lin_acc = v3(0, 0, 0); out->ball_ang_acc = v3(0, 0, 0);
  if (mode == 0) {
    /* Isaac Gym's net contact force tensor sums the solver's NORMAL contact impulses only [ext]; the
     * checkpoint's obs statistics show it (feet flags 1..3 are +1 17-45 % of the time, which needs fx = fy = 0
     * under load, kick_env.py:993-1038).  BEZ_FLAG_CF_WITH_FRICTION adds the friction part. */
    const int with_fric = (c->flags & BEZ_FLAG_CF_WITH_FRICTION) != 0;
    for (int i = 0; i < nhit; ++i) {
      const GroundHit* hh = &hits[i];
      SV a = acc[hh->link];
      V3 ap = v3add(sv_lin(a), v3cross(sv_ang(a), hh->x));
      int body = hh->body;
      if (with_fric) {
        out->contact_force[body][0] += hh->ftx0 - h * hh->ct * ap.v[0];
        out->contact_force[body][1] += hh->fty0 - h * hh->ct * ap.v[1];
      }
      out->contact_force[body][2] += hh->fn0 - hh->kn * ap.v[2];
    }
    for (int i = 0; i < AS.n; ++i) { /* calf <-> foot plate: +lambda n on the calf, -lambda n on the foot */
      const real kimp = h * h * m_stop_kn(c) + h * m_stop_cn(c);
      real lam = AS.lam0[i] - kimp * (AS.Ja[i] * out->qdd[AS.la[i] - 1] + AS.Jf[i] * out->qdd[AS.lf[i] - 1]);
      if (!(lam > 0)) continue;
      int bcalf = m_link_body(c, AS.la[i] - 1), bfoot = m_link_body(c, AS.lf[i]);
      for (int j = 0; j < 3; ++j) { out->contact_force[bcalf][j] += lam * AS.nrm[i].v[j]; out->contact_force[bfoot][j] -= lam * AS.nrm[i].v[j]; }
    }
    V3 fl = v3(0, 0, 0); /* force on the link from the ball */
    if (bl_link >= 0) {
      SV a = acc[bl_link];
      V3 ap = v3add(sv_lin(a), v3cross(sv_ang(a), bl_x));
      fl = v3sub(bl_f0p, m3mulv(&bl_A, ap));
      int body = m_link_body(c, bl_link);
      V3 fr = with_fric ? fl : v3scale(bl_n, v3dot(fl, bl_n));
      for (int i = 0; i < 3; ++i) { out->contact_force[body][i] += fr.v[i]; out->contact_force[nb][i] -= fr.v[i]; }
    }
    /* ball: Mb ab = -pb - Jb^T fl */
    SV rhs = sv_add(sv_scale(pb, -1), sv_scale(wrench_at(bl_xb, fl), -1));
    M6 Mc = Mb;
    chol6_solve(&Mc, &rhs, 1);
    out->ball_ang_acc = sv_ang(rhs); out->ball_lin_acc = sv_lin(rhs);
    if (ball_ground) {
      V3 ap = v3add(out->ball_lin_acc, v3cross(out->ball_ang_acc, bhit.x));
      if (with_fric) {
        out->contact_force[nb][0] += bhit.ftx0 - h * bhit.ct * ap.v[0];
        out->contact_force[nb][1] += bhit.fty0 - h * bhit.ct * ap.v[1];
      }
      out->contact_force[nb][2] += bhit.fn0 - bhit.kn * ap.v[2];
    }
  }
}

/* One evaluation of the model.  tune[22] = n > 0 (ORACLE ONLY; libbez_sim.so refuses every non-zero tune[] entry): the speed-limit locks
 * are found by an active-set iteration of up to n passes (a joint that ends beyond the limit is locked and the dynamics re-evaluated;
 * locks are only added) instead of by the predictor -- the reference the predictor's misses are measured against. */
static void dynamics(const BezSimConfig* c, const Env* e, real h, int mode, const real* tau_in, Dyn* out) {
  const int npass = (int)c->tune[22];
  if (mode != 0 || npass <= 0) { dynamics_x(c, e, h, mode, tau_in, out, NULL); return; }
  int lock[ND]; memset(lock, 0, sizeof(lock));
  for (int it = 0; it < npass; ++it) {
    dynamics_x(c, e, h, mode, tau_in, out, lock);
    int changed = 0;
    for (int d = 0; d < ND; ++d) {
      real v = e->qd[d] + h * out->qdd[d], vl = (real)c->vel_limit;
      if (!lock[d]) { if (v > vl * (1 + (real)1e-9)) { lock[d] = 1; changed = 1; } else if (v < -vl * (1 + (real)1e-9)) { lock[d] = -1; changed = 1; } }
    }
    if (!changed) break;
  }
}

/* ================================================================== rigid contact (BEZ_FLAG_HARD_CONTACT)
 * Velocity-level contact with Coulomb stiction and restitution 0, as the reference configures PhysX (plane static = dynamic
 * friction 1, restitution 0: bez_kick.yaml:13-16, kick_env.py:250-256; rest_offset 0, contact_offset 0.02, max depenetration
 * velocity: bez_kick.yaml:139-144).  PhysX's TGS solver itself is a closed binary; this is the textbook formulation it
 * belongs to (SURVEY.md 7 "Contact"): contact impulses p at points, solved by projected Gauss-Seidel on the Delassus matrix
 * W = J M~^-1 J^T whose columns are the articulated-body impulse responses (Featherstone, RBDA 11.2 style) of the SAME factorisation the forward dynamics used
 * -- M~ contains the implicit drive / limit / joint-friction terms, so the drives react to a contact impulse inside the step.
 * The drive's effort limit is then re-checked against the torque the solved step implies, and the step is repeated with
 * the corrected saturation set (active-set iteration; PhysX clamps the drive impulse inside its iterations).
 *   knobs (BezSimConfig.tune, 0 = default): [0] PGS sweeps (16)  [1] penetration ERP (0.2)  [2] detection margin m (0.02 = physx.contact_offset, bez_kick.yaml:139)
 *   [3] max depenetration speed m/s (1)  [4] constraint-force mixing (1e-6)  [5] effort active-set passes (3)  [6] contact-normal compliance */
#define HC_MAXC 48
#define HC_BALL NL   /* body index of the ball; -1 = the world */
typedef struct {
  int a, b;        /* bodies: link index, HC_BALL, or -1 (world, b only).  The impulse +p acts on a, -p on b */
  V3 xa, xb;       /* contact point relative to the body's reference point (O for links, the centre for the ball) */
  V3 d[3];         /* normal (from b towards a), two tangents */
  real phi, mu;    /* signed distance (negative: penetration), friction coefficient */
  int row_a, row_b;/* rows of the net-contact-force tensor (-1: none) */
  real p[3];
} HContact;

static V3 point_vel(SV V, V3 x) { return v3add(sv_lin(V), v3cross(sv_ang(V), x)); }

typedef struct { SV S[NL], U[NL]; real Dinv[NL]; M6 L0; int lock[NL]; /* joint rate prescribed (speed limit active): rigid for impulses */ } AbaFactor;
/* velocity response of every link to impulsive wrenches Pin[l] (about O): dV[l], and the joint-rate part dqd */
static void impulse_response(const AbaFactor* F, const SV* Pin, SV* dV, real* dqd) {
  SV pI[NL]; real u[NL];
  for (int l = 0; l < NL; ++l) pI[l] = sv_scale(Pin[l], -1);
  for (int l = NL - 1; l >= 1; --l) {
    int p = BEZ_LINK_PARENT[l];
    if (F->lock[l]) { u[l] = 0; pI[p] = sv_add(pI[p], pI[l]); continue; }
    u[l] = -sv_dot(F->S[l], pI[l]);
    pI[p] = sv_add(pI[p], sv_add(pI[l], sv_scale(F->U[l], u[l] * F->Dinv[l])));
  }
  dV[0] = sv_scale(pI[0], -1);
  chol6_subst(&F->L0, &dV[0], 1);
  for (int l = 1; l < NL; ++l) {
    int p = BEZ_LINK_PARENT[l];
    real r = F->lock[l] ? 0 : (u[l] - sv_dot(F->U[l], dV[p])) * F->Dinv[l];
    if (dqd) dqd[l - 1] = r;
    dV[l] = sv_add(dV[p], sv_scale(F->S[l], r));
  }
}

static void tangent_basis(V3 n, V3* t1, V3* t2) {
  V3 a = fabs(n.v[0]) < (real)0.7 ? v3(1, 0, 0) : v3(0, 1, 0);
  V3 t = v3sub(a, v3scale(n, v3dot(a, n)));
  *t1 = v3scale(t, 1 / sqrt(v3dot(t, t)));
  *t2 = v3cross(n, *t1);
}

typedef struct {
  SV a0; real qdd[ND];                 /* contact-free accelerations */
  SV dV0; real dqd[ND];                /* velocity change by the contact impulses */
  V3 ball_lin_acc, ball_dlin, ball_dang;
  real contact_force[NBMAX][3];
} DynH;

static real hc_knob(const BezSimConfig* c, int i, real dflt) { return c->tune[i] != 0 ? (real)c->tune[i] : dflt; }

static void dynamics_hard(const BezSimConfig* c, const Env* e, real h, DynH* out) {
  Pass1 P1;
  aba_pass1(c, e, &P1);
  const Kin* k = &P1.k;
  const int nb = m_nb(c);
  const SV* V = P1.V; const SV* cb = P1.cb;
  const real mu = e->friction;
  const real margin = hc_knob(c, 2, (real)0.02), erp = hc_knob(c, 1, (real)0.2), vdepen = hc_knob(c, 3, (real)1.0);
  const real cfm = hc_knob(c, 4, (real)1e-6);
  const int sweeps = (int)hc_knob(c, 0, 16), npass = (int)hc_knob(c, 5, 3);
  const real R = (real)BEZ_BALL_RADIUS, mb = (real)BEZ_BALL_MASS, Ib = (real)BEZ_BALL_INERTIA;
  const real ro_r = (real)c->tune[6], ro_b = 2 * (real)c->tune[6]; /* EXPERIMENT: shape rest offsets (asset thickness) */
  V3 g = v3(e->gravity[0], e->gravity[1], e->gravity[2]);
  memset(out, 0, sizeof(*out));

  /* explicit leg <-> leg penalty contact, as in the compliant model */
  SV pS[NL]; memset(pS, 0, sizeof(pS));
  if (!(c->flags & BEZ_FLAG_NO_SELF_COLLISION))
    self_collision(c, 0, mu, k, V, pS, out->contact_force, (c->flags & BEZ_FLAG_CF_WITH_FRICTION) != 0, NULL);

  /* ---- contact detection */
  HContact C[HC_MAXC]; int nc = 0;
  for (int i = 0; i < BEZ_NPT; ++i) { /* ground points of the feet / cleats and of the upper body */
    int l = BEZ_PT_LINK[i];
    V3 pl = v3((real)m_pt_pos(c, i)[0], (real)m_pt_pos(c, i)[1], (real)m_pt_pos(c, i)[2]);
    V3 x = v3add(k->r[l], m3mulv(&k->E[l], pl));
    real z = e->root_pos[2] + x.v[2] - ro_r;
    if (!(z < margin)) continue;
    HContact* q = &C[nc++];
    q->a = l; q->b = -1; q->xa = x; q->xb = v3(0, 0, 0); q->d[0] = v3(0, 0, 1); q->phi = z; q->mu = mu; q->row_a = m_pt_body(c, i); q->row_b = -1;
  }
  V3 bc = v3(e->ball_pos[0] - e->root_pos[0], e->ball_pos[1] - e->root_pos[1], e->ball_pos[2] - e->root_pos[2]); /* ball centre rel. O */
  if (m_has_ball(c)) {
    if (e->ball_pos[2] - R - ro_b < margin) {
      HContact* q = &C[nc++];
      q->a = HC_BALL; q->b = -1; q->xa = v3(0, 0, -R); q->xb = v3(0, 0, 0); q->d[0] = v3(0, 0, 1); q->phi = e->ball_pos[2] - R - ro_b; q->mu = mu; q->row_a = nb; q->row_b = -1;
    }
    /* ball vs the leg boxes and the torso box: every box within the margin is a contact */
    for (int b = 0; b < BEZ_NBOX && nc < HC_MAXC; ++b) {
      int l = BEZ_BOX_LINK[b];
      V3 cl = v3((real)m_box_center(c, b)[0], (real)m_box_center(c, b)[1], (real)m_box_center(c, b)[2]);
      V3 he = v3((real)m_box_half(c, b)[0], (real)m_box_half(c, b)[1], (real)m_box_half(c, b)[2]);
      V3 ql = v3sub(m3Tmulv(&k->E[l], v3sub(bc, k->r[l])), cl);
      V3 cp; int inside = 1;
      for (int i = 0; i < 3; ++i) {
        real t = ql.v[i];
        if (t > he.v[i]) { t = he.v[i]; inside = 0; }
        if (t < -he.v[i]) { t = -he.v[i]; inside = 0; }
        cp.v[i] = t;
      }
      V3 nl; real depth;
      if (!inside) {
        V3 dlt = v3sub(ql, cp);
        real dist = sqrt(v3dot(dlt, dlt));
        depth = R + ro_r + ro_b - dist;
        if (!(depth > -margin)) continue;
        nl = v3scale(dlt, 1 / dist);
      } else {
        int ax = 0; real md = he.v[0] - fabs(ql.v[0]);
        for (int i = 1; i < 3; ++i) { real di = he.v[i] - fabs(ql.v[i]); if (di < md) { md = di; ax = i; } }
        nl = v3(0, 0, 0); nl.v[ax] = ql.v[ax] >= 0 ? 1 : -1;
        cp = ql; cp.v[ax] = nl.v[ax] * he.v[ax];
        depth = R + ro_r + ro_b + md;
      }
      V3 bP = v3add(k->r[l], m3mulv(&k->E[l], v3add(cp, cl)));
      HContact* q = &C[nc++];
      q->a = HC_BALL; q->b = l; q->xa = v3sub(bP, bc); q->xb = bP; q->d[0] = m3mulv(&k->E[l], nl); q->phi = -depth; q->mu = mu;
      q->row_a = nb; q->row_b = m_link_body(c, l);
    }
  }
  for (int i = 0; i < nc; ++i) { tangent_basis(C[i].d[0], &C[i].d[1], &C[i].d[2]); C[i].p[0] = C[i].p[1] = C[i].p[2] = 0; }

  /* ---- drive terms per joint (implicit PD, joint friction, limit spring), and the first saturation guess */
  real tau_pd0[ND], k_pd[ND], tau_o[ND], k_o[ND];
  int sat[ND];
  for (int d = 0; d < ND; ++d) {
    real kp = (real)c->kp * e->kp_scale[d], kdm = (real)c->kd * e->kd_scale[d];
    tau_pd0[d] = kp * (e->target[d] - e->q[d] - h * e->qd[d]) - kdm * e->qd[d];
    k_pd[d] = h * h * kp + h * kdm;
    real cf = (real)c->joint_friction / fmax(fabs(e->qd[d]), (real)c->jfric_veps);
    real k_l = 0, tau_l0 = 0;
    real lo = e->lim_lo[d], hi = e->lim_hi[d];
    if (e->q[d] < lo) { tau_l0 = (real)c->limit_k * (lo - e->q[d] - h * e->qd[d]) - (real)c->limit_d * e->qd[d]; k_l = h * h * (real)c->limit_k + h * (real)c->limit_d; }
    else if (e->q[d] > hi) { tau_l0 = (real)c->limit_k * (hi - e->q[d] - h * e->qd[d]) - (real)c->limit_d * e->qd[d]; k_l = h * h * (real)c->limit_k + h * (real)c->limit_d; }
    tau_o[d] = -cf * e->qd[d] + tau_l0;
    k_o[d] = h * cf + k_l;
    sat[d] = 2; /* 2 = not decided yet: pass 0 uses the held-parent predictor of the compliant model */
  }

  AbaFactor F;
  memcpy(F.S, P1.S, sizeof(F.S));
  memset(F.lock, 0, sizeof(F.lock));
  real qdd_fix[ND]; /* prescribed joint acceleration of a speed-limited joint */
  const int consistent_vlim = c->tune[7] == 1; /* [7] != 0: the joint speed limit is a constraint (prescribed-rate joint), not a post-hoc clamp */
  for (int pass = 0; pass < npass; ++pass) {
    /* pass 2 */
    M6 IA[NL]; SV pA[NL], pSl[NL]; real u[NL];
    memcpy(IA, P1.IA, sizeof(IA)); memcpy(pA, P1.pA, sizeof(pA)); memcpy(pSl, pS, sizeof(pSl));
    for (int l = NL - 1; l >= 1; --l) {
      int p = BEZ_LINK_PARENT[l], d = l - 1;
      F.U[l] = m6mulv(&IA[l], F.S[l]);
      if (F.lock[l]) { /* hybrid dynamics: known joint acceleration, the articulated inertia passes through unprojected */
        SV pa = sv_add(pA[l], m6mulv(&IA[l], sv_add(cb[l], sv_scale(F.S[l], qdd_fix[d]))));
        for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) IA[p].m[i][j] += IA[l].m[i][j];
        pA[p] = sv_add(pA[p], pa);
        pSl[p] = sv_add(pSl[p], pSl[l]);
        F.Dinv[l] = 0; u[l] = 0;
        if (sat[d] == 2) sat[d] = 0;
        continue;
      }
      real J = sv_dot(F.S[l], F.U[l]) + (real)c->armature;
      if (sat[d] == 2) {
        real bias = sv_dot(F.S[l], pA[l]) + sv_dot(F.U[l], cb[l]);
        real qdd_est = (tau_pd0[d] + tau_o[d] - bias) / (J + k_pd[d] + k_o[d]);
        real tau_drive = tau_pd0[d] - k_pd[d] * qdd_est;
        sat[d] = tau_drive > (real)c->effort ? 1 : (tau_drive < -(real)c->effort ? -1 : 0);
      }
      real tau, D;
      if (sat[d] != 0) { tau = sat[d] * (real)c->effort + tau_o[d]; D = J + k_o[d]; }
      else { tau = tau_pd0[d] + tau_o[d]; D = J + k_pd[d] + k_o[d]; }
      F.Dinv[l] = 1 / D;
      real u_main = tau - sv_dot(F.S[l], pA[l]), du = -sv_dot(F.S[l], pSl[l]);
      u[l] = u_main + du;
      M6 Ia = IA[l];
      for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) Ia.m[i][j] -= F.U[l].v[i] * F.U[l].v[j] * F.Dinv[l];
      SV pa = sv_add(pA[l], sv_add(m6mulv(&Ia, cb[l]), sv_scale(F.U[l], u_main * F.Dinv[l])));
      for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) IA[p].m[i][j] += Ia.m[i][j];
      pA[p] = sv_add(pA[p], pa);
      pSl[p] = sv_add(pSl[p], sv_add(pSl[l], sv_scale(F.U[l], du * F.Dinv[l])));
    }
    F.L0 = IA[0];
    chol6_factor(&F.L0);
    SV a0 = sv_scale(sv_add(pA[0], pSl[0]), -1);
    chol6_subst(&F.L0, &a0, 1);
    out->a0 = a0;
    /* pass 3 */
    SV acc[NL];
    acc[0] = a0;
    for (int l = 1; l < NL; ++l) {
      int p = BEZ_LINK_PARENT[l];
      SV ap = sv_add(acc[p], cb[l]);
      real qdd = F.lock[l] ? qdd_fix[l - 1] : (u[l] - sv_dot(F.U[l], ap)) * F.Dinv[l];
      out->qdd[l - 1] = qdd;
      acc[l] = sv_add(ap, sv_scale(F.S[l], qdd));
    }
    out->ball_lin_acc = g;
    memset(&out->dV0, 0, sizeof(SV)); memset(out->dqd, 0, sizeof(out->dqd));
    out->ball_dlin = out->ball_dang = v3(0, 0, 0);

    if (nc > 0) {
      /* contact-free end-of-step velocities */
      SV Vf[NL + 1];
      for (int l = 0; l < NL; ++l) Vf[l] = sv_add(V[l], sv_scale(acc[l], h));
      Vf[HC_BALL] = sv(v3(e->ball_ang[0], e->ball_ang[1], e->ball_ang[2]),
                       v3(e->ball_lin[0] + h * g.v[0], e->ball_lin[1] + h * g.v[1], e->ball_lin[2] + h * g.v[2]));
      static const SV SV0 = {{0, 0, 0, 0, 0, 0}};
      const int m3n = 3 * nc;
      real W[3 * HC_MAXC][3 * HC_MAXC], bf[3 * HC_MAXC];
      for (int i = 0; i < nc; ++i) for (int kx = 0; kx < 3; ++kx) {
        const HContact* ci = &C[i];
        V3 dir = ci->d[kx];
        SV Pin[NL], dV[NL + 1];
        for (int l = 0; l < NL; ++l) Pin[l] = SV0;
        dV[HC_BALL] = SV0;
        int tree = 0;
        if (ci->a == HC_BALL) dV[HC_BALL] = sv(v3scale(v3cross(ci->xa, dir), 1 / Ib), v3scale(dir, 1 / mb));
        else { Pin[ci->a] = sv_add(Pin[ci->a], wrench_at(ci->xa, dir)); tree = 1; }
        if (ci->b >= 0) { Pin[ci->b] = sv_add(Pin[ci->b], sv_scale(wrench_at(ci->xb, dir), -1)); tree = 1; }
        if (tree) impulse_response(&F, Pin, dV, NULL); else for (int l = 0; l < NL; ++l) dV[l] = SV0;
        for (int j = 0; j < nc; ++j) {
          const HContact* cj = &C[j];
          V3 du = point_vel(dV[cj->a], cj->xa);
          if (cj->b >= 0) du = v3sub(du, point_vel(dV[cj->b], cj->xb));
          for (int mx = 0; mx < 3; ++mx) W[3 * j + mx][3 * i + kx] = v3dot(cj->d[mx], du);
        }
      }
      for (int j = 0; j < nc; ++j) {
        const HContact* cj = &C[j];
        V3 uf = point_vel(Vf[cj->a], cj->xa);
        if (cj->b >= 0) uf = v3sub(uf, point_vel(Vf[cj->b], cj->xb));
        for (int mx = 0; mx < 3; ++mx) bf[3 * j + mx] = v3dot(cj->d[mx], uf);
        /* restitution 0; penetration is pushed out with ERP (capped), a gap may close within the step */
        real vt = cj->phi < 0 ? fmin(-erp * cj->phi / h, vdepen) : -cj->phi / h;
        bf[3 * j] -= vt;
      }
      real pv[3 * HC_MAXC];
      for (int i = 0; i < nc; ++i) for (int kx = 0; kx < 3; ++kx) pv[3 * i + kx] = C[i].p[kx]; /* warm start from the previous pass */
      for (int it = 0; it < sweeps; ++it) {
        for (int i = 0; i < nc; ++i) {
          int r0 = 3 * i;
          real un = bf[r0];
          for (int j = 0; j < m3n; ++j) un += W[r0][j] * pv[j];
          real pn = pv[r0] - un / (W[r0][r0] + cfm);
          if (pn < 0) pn = 0;
          pv[r0] = pn;
          for (int kx = 1; kx < 3; ++kx) {
            int r = r0 + kx;
            real ut = bf[r];
            for (int j = 0; j < m3n; ++j) ut += W[r][j] * pv[j];
            pv[r] -= ut / (W[r][r] + cfm);
          }
          real lim = C[i].mu * pn, pt = sqrt(pv[r0 + 1] * pv[r0 + 1] + pv[r0 + 2] * pv[r0 + 2]);
          if (pt > lim) { real sc = pt > 0 ? lim / pt : 0; pv[r0 + 1] *= sc; pv[r0 + 2] *= sc; }
        }
      }
      for (int i = 0; i < nc; ++i) for (int kx = 0; kx < 3; ++kx) C[i].p[kx] = pv[3 * i + kx];
      /* apply the impulses */
      SV Pin[NL], dV[NL];
      for (int l = 0; l < NL; ++l) Pin[l] = SV0;
      V3 bl = v3(0, 0, 0), ba = v3(0, 0, 0);
      for (int i = 0; i < nc; ++i) {
        const HContact* ci = &C[i];
        V3 imp = v3add(v3scale(ci->d[0], ci->p[0]), v3add(v3scale(ci->d[1], ci->p[1]), v3scale(ci->d[2], ci->p[2])));
        if (ci->a == HC_BALL) { bl = v3add(bl, v3scale(imp, 1 / mb)); ba = v3add(ba, v3scale(v3cross(ci->xa, imp), 1 / Ib)); }
        else Pin[ci->a] = sv_add(Pin[ci->a], wrench_at(ci->xa, imp));
        if (ci->b >= 0) Pin[ci->b] = sv_add(Pin[ci->b], sv_scale(wrench_at(ci->xb, imp), -1));
      }
      impulse_response(&F, Pin, dV, out->dqd);
      out->dV0 = dV[0];
      out->ball_dlin = bl; out->ball_dang = ba;
    }
    /* effort limit against the torque the solved step implies */
    int changed = 0;
    for (int d = 0; d < ND; ++d) {
      if (F.lock[d + 1]) continue;
      if (consistent_vlim) {
        real vn = e->qd[d] + h * out->qdd[d] + out->dqd[d];
        if (fabs(vn) > (real)c->vel_limit) { F.lock[d + 1] = 1; qdd_fix[d] = ((vn > 0 ? (real)c->vel_limit : -(real)c->vel_limit) - e->qd[d]) / h; changed = 1; continue; }
      }
      real qdd_tot = out->qdd[d] + out->dqd[d] / h;
      real tau_u = tau_pd0[d] - k_pd[d] * qdd_tot; /* what the unsaturated implicit PD would apply at this acceleration */
      int ns = tau_u > (real)c->effort ? 1 : (tau_u < -(real)c->effort ? -1 : 0);
      if (ns != sat[d]) { sat[d] = ns; changed = 1; }
    }
    if (!changed) break;
  }
  /* net contact force rows: normal parts (BEZ_FLAG_CF_WITH_FRICTION: whole impulse), as forces over the substep */
  const int with_fric = (c->flags & BEZ_FLAG_CF_WITH_FRICTION) != 0;
  for (int i = 0; i < nc; ++i) {
    const HContact* ci = &C[i];
    V3 imp = v3scale(ci->d[0], ci->p[0]);
    if (with_fric) imp = v3add(imp, v3add(v3scale(ci->d[1], ci->p[1]), v3scale(ci->d[2], ci->p[2])));
    for (int a = 0; a < 3; ++a) {
      if (ci->row_a >= 0) out->contact_force[ci->row_a][a] += imp.v[a] / h;
      if (ci->row_b >= 0) out->contact_force[ci->row_b][a] -= imp.v[a] / h;
    }
  }
}

static void quat_integrate(real q[4], const real w[3], real h) {
  /* q <- normalize(q + h/2 * (w,0) (x) q), xyzw, world-frame angular velocity */
  real x = q[0], y = q[1], z = q[2], s = q[3];
  real dx = w[0] * s + w[1] * z - w[2] * y;
  real dy = -w[0] * z + w[1] * s + w[2] * x;
  real dz = w[0] * y - w[1] * x + w[2] * s;
  real dw = -w[0] * x - w[1] * y - w[2] * z;
  real hh = (real)0.5 * h;
  x += hh * dx; y += hh * dy; z += hh * dz; s += hh * dw;
  real n = 1 / sqrt(x * x + y * y + z * z + s * s);
  q[0] = x * n; q[1] = y * n; q[2] = z * n; q[3] = s * n;
}

static void substep_hard(const BezSimConfig* c, Env* e, real h, int first, real wgt) {
  DynH d;
  const int pre_clamp = c->tune[7] == 2; /* EXPERIMENT: speed limit applied to the incoming joint rates only */
  if (pre_clamp) for (int j = 0; j < ND; ++j) { real vl = (real)c->vel_limit; if (e->qd[j] > vl) e->qd[j] = vl; if (e->qd[j] < -vl) e->qd[j] = -vl; }
  dynamics_hard(c, e, h, &d);
  for (int j = 0; j < ND; ++j) {
    real v = e->qd[j] + h * d.qdd[j] + d.dqd[j];
    real vl = pre_clamp ? (real)1e9 : (real)c->vel_limit;
    if (v > vl) v = vl;
    if (v < -vl) v = -vl;
    e->qd[j] = v;
    e->q[j] += h * v;
  }
  V3 w = v3(e->root_ang[0], e->root_ang[1], e->root_ang[2]), v = v3(e->root_lin[0], e->root_lin[1], e->root_lin[2]);
  V3 vdot = v3add(sv_lin(d.a0), v3cross(w, v));
  V3 wdot = sv_ang(d.a0);
  for (int i = 0; i < 3; ++i) {
    e->root_ang[i] += h * wdot.v[i] + d.dV0.v[i];
    e->root_lin[i] += h * vdot.v[i] + d.dV0.v[3 + i];
    e->root_pos[i] += h * e->root_lin[i];
  }
  quat_integrate(e->root_quat, e->root_ang, h);
  real damp = 1 - h * (real)c->ball_ang_damping;
  if (damp < 0) damp = 0;
  for (int i = 0; i < 3; ++i) {
    e->ball_lin[i] += h * d.ball_lin_acc.v[i] + d.ball_dlin.v[i];
    e->ball_ang[i] = (e->ball_ang[i] + d.ball_dang.v[i]) * damp;
    e->ball_pos[i] += h * e->ball_lin[i];
  }
  quat_integrate(e->ball_quat, e->ball_ang, h);
  for (int b = 0; b < NBMAX; ++b) for (int i = 0; i < 3; ++i) e->contact_force[b][i] = first ? d.contact_force[b][i] * wgt : e->contact_force[b][i] + d.contact_force[b][i] * wgt;
}

#include "bez_oracle_tgs.inc"

static void substep(const BezSimConfig* c, Env* e, real h, int first, real wgt) {
  if (c->flags & BEZ_FLAG_TGS_SOLVER) { substep_tgs(c, e, h, first, wgt); return; }
  if (c->flags & BEZ_FLAG_HARD_CONTACT) { substep_hard(c, e, h, first, wgt); return; }
  Dyn d;
  dynamics(c, e, h, 0, NULL, &d);
  /* joints: semi-implicit Euler.  The speed limit (kick_env.py:327) is inside the dynamics (a prescribed-rate joint in pass 2):
   * no rate is edited here */
  for (int j = 0; j < ND; ++j) {
    real v = e->qd[j] + h * d.qdd[j];
    e->qd[j] = v;
    e->q[j] += h * v;
  }
  /* root: spatial -> classical acceleration of the torso origin */
  V3 w = v3(e->root_ang[0], e->root_ang[1], e->root_ang[2]), v = v3(e->root_lin[0], e->root_lin[1], e->root_lin[2]);
  V3 vdot = v3add(sv_lin(d.a0), v3cross(w, v));
  V3 wdot = sv_ang(d.a0);
  for (int i = 0; i < 3; ++i) {
    e->root_ang[i] += h * wdot.v[i];
    e->root_lin[i] += h * vdot.v[i];
    e->root_pos[i] += h * e->root_lin[i];
  }
  quat_integrate(e->root_quat, e->root_ang, h);
  /* ball */
  real damp = 1 - h * (real)c->ball_ang_damping;
  if (damp < 0) damp = 0;
  for (int i = 0; i < 3; ++i) {
    e->ball_lin[i] += h * d.ball_lin_acc.v[i];
    e->ball_ang[i] = (e->ball_ang[i] + h * d.ball_ang_acc.v[i]) * damp;
    e->ball_pos[i] += h * e->ball_lin[i];
  }
  quat_integrate(e->ball_quat, e->ball_ang, h);
  e->vmargin = first ? d.vmargin : (d.vmargin < e->vmargin ? d.vmargin : e->vmargin);
  /* physx.contact_collection 2 = CC_ALL_SUBSTEPS (bez_kick.yaml:147): the tensor holds the mean force over the control step [ext] */
  for (int b = 0; b < NBMAX; ++b) for (int i = 0; i < 3; ++i) e->contact_force[b][i] = first ? d.contact_force[b][i] * wgt : e->contact_force[b][i] + d.contact_force[b][i] * wgt;
}

/* ------------------------------------------------------------------ env logic (reference restatement) */
static void env_reset(const BezSimConfig* c, Env* e, int64_t genv, const float* goal) {
  if (c->task != BEZ_TASK_KICK) { e->goal[0] = goal[0]; e->goal[1] = goal[1]; }
  /* kick_env.py:786-791: q = clamp(default + U(-.15,.15), lo, hi); qd = U(-.1,.1) */
  for (int j = 0; j < ND; ++j) {
    float up = reset_uniform(c->seed, genv, e->episode, j);
    float uv = reset_uniform(c->seed, genv, e->episode, ND + j);
    float off = fmaf(0.3f, up, -0.15f); /* torch_rand_float: (upper-lower)*rand + lower */
    float vel = fmaf(0.2f, uv, -0.1f);
    real q = (real)((float)BEZ_DOF_DEFAULT[j] + off);
    real lo = (real)(float)BEZ_DOF_LOWER[j], hi = (real)(float)BEZ_DOF_UPPER[j];
    if (q > hi) q = hi;
    if (q < lo) q = lo;
    e->q[j] = q;
    e->qd[j] = (real)vel;
    e->target[j] = (real)(float)BEZ_DOF_DEFAULT[j]; /* kick_env.py:839-842 */
  }
  e->episode += 1;
  /* kick_env.py:163-166,831-837: root states <- initial (zero twist) */
  for (int i = 0; i < 3; ++i) { e->root_pos[i] = c->bez_init[i]; e->ball_pos[i] = c->ball_init[i]; e->root_lin[i] = e->root_ang[i] = e->ball_lin[i] = e->ball_ang[i] = 0; }
  for (int i = 0; i < 4; ++i) { e->root_quat[i] = c->bez_init[3 + i]; e->ball_quat[i] = c->ball_init[3 + i]; }
  memset(e->contact_force, 0, sizeof(e->contact_force));
  memset(e->ftrans, 0, sizeof(e->ftrans));
  e->progress = 0; /* kick_env.py:849-850 */
  e->reset = 0;
}

static void env_pre_physics(const BezSimConfig* c, Env* e, const float* act) {
  /* vec_task.py:317 + kick_env.py:413-418, evaluated in fp32 like the reference */
  for (int j = 0; j < ND; ++j) {
    float a = act[j];
    if (a > c->clip_actions) a = c->clip_actions;
    if (a < -c->clip_actions) a = -c->clip_actions;
    if (j < 2) a = 0.0f;
    float t = a + (float)BEZ_DOF_DEFAULT[j];
    float lo = (float)BEZ_DOF_LOWER[j], hi = (float)BEZ_DOF_UPPER[j];
    if (t > hi) t = hi; /* tensor_clamp = max(min(t, upper), lower) */
    if (t < lo) t = lo;
    e->target[j] = (real)t;
  }
}

/* yaw of get_euler_xyz (isaacgym.torch_utils [ext]), wrapped to [0, 2pi) */
static real euler_yaw(const real q[4]) {
  real x = q[0], y = q[1], z = q[2], w = q[3];
  real siny = 2 * (w * z + x * y), cosy = w * w + x * x - y * y - z * z;
  real yaw = atan2(siny, cosy);
  real twopi = (real)6.283185307179586;
  yaw = fmod(yaw, twopi);
  if (yaw < 0) yaw += twopi;
  return yaw;
}

static void feet_no_cleats(real f[3], real out[4]) {
  /* kick_env.py:987-990: noise filter written back into the sim tensor */
  for (int i = 0; i < 3; ++i) if (!(fabs(f[i]) > (real)0.01)) f[i] = 0;
  /* kick_env.py:993-1006: "sign" codes never test the sign (quirk Q3) */
  real x = (fabs(f[0]) > 0) ? 1 : 0;
  if (f[0] == 0) x = 2;
  real y = (fabs(f[1]) > 0) ? 1 : 3;
  if (f[1] == 0) y = 3;
  real sensor = (x == 1) ? 0 : 4;
  if (x == 2) sensor = 8;
  real cs = y + sensor;
  static const real tab[12][4] = {{0}, {1, -1, -1, -1}, {-1, -1, 1, -1}, {1, -1, 1, -1}, {0}, {-1, 1, -1, -1}, {-1, -1, -1, 1},
                                  {-1, 1, -1, 1}, {0}, {1, 1, -1, -1}, {-1, -1, 1, 1}, {1, 1, 1, 1}};
  int ci = (int)cs;
  for (int i = 0; i < 4; ++i) out[i] = -1;
  if (ci == 1 || ci == 2 || ci == 3 || ci == 5 || ci == 6 || ci == 7 || ci == 9 || ci == 10 || ci == 11)
    for (int i = 0; i < 4; ++i) out[i] = tab[ci][i];
  if (f[2] < 1) for (int i = 0; i < 4; ++i) out[i] = -1; /* kick_env.py:1036-1038 */
}

/* use_prev: take the finite difference against the stored prev_lin_vel.  The reference does so only on
 * the first compute_imu call of a process (prev = zeros, kick_env.py:183); afterwards prev aliases the
 * live velocity tensor (kick_env.py:930,441) and the difference is identically zero (quirk Q1). */
/* compute_feet_sensors_cleats (kick_env.py:1044-1069): +1 where the cleat's contact force norm exceeds 1 N */
static void feet_cleats(const Env* e, real out[8]) {
  for (int k = 0; k < 8; ++k) {
    const real* f = e->contact_force[(k < 4 ? BEZ_LCLEAT_BODY_CL : BEZ_RCLEAT_BODY_CL - 4) + k];
    out[k] = sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]) > 1 ? 1 : -1;
  }
}

static real wrap_pi(real a) { return atan2(sin(a), cos(a)); } /* isaacgym.torch_utils.normalize_angle [ext] */

static void env_observe_reward(const BezSimConfig* c, Env* e, int use_prev) {
  /* IMU link (body 1) is rigidly at the torso origin with identity offset (soccerbot_stl.urdf:567-572),
   * so its pose/twist equal the root state. */
  const real* q = e->root_quat; /* xyzw */
  real vimu[3] = {e->root_lin[0], e->root_lin[1], e->root_lin[2]};
  real wimu[3] = {e->root_ang[0], e->root_ang[1], e->root_ang[2]};
  /* compute_imu, kick_env.py:918-930 */
  real acc[3];
  real dt = (real)c->dt;
  real gunit[3] = {0, 0, -1}; /* kick_env.py:217 */
  for (int i = 0; i < 3; ++i) {
    real prev = use_prev ? e->prev_lin_vel[i] : vimu[i];
    acc[i] = (vimu[i] - prev) / dt - gunit[i];
  }
  /* quaternion_to_matrix with (r,i,j,k) <- (x,y,z,w): quirk Q2 */
  real r = q[0], i_ = q[1], j_ = q[2], k_ = q[3];
  real two_s = 2 / (r * r + i_ * i_ + j_ * j_ + k_ * k_);
  real Rm[3][3] = {{1 - two_s * (j_ * j_ + k_ * k_), two_s * (i_ * j_ - k_ * r), two_s * (i_ * k_ + j_ * r)},
                   {two_s * (i_ * j_ + k_ * r), 1 - two_s * (i_ * i_ + k_ * k_), two_s * (j_ * k_ - i_ * r)},
                   {two_s * (i_ * k_ - j_ * r), two_s * (j_ * k_ + i_ * r), 1 - two_s * (i_ * i_ + j_ * j_)}};
  real imu[6];
  for (int a = 0; a < 3; ++a) {
    real s = Rm[a][0] * acc[0] + Rm[a][1] * acc[1] + Rm[a][2] * acc[2];
    real lim = (real)(2. * 9.81);
    imu[a] = s > lim ? lim : (s < -lim ? -lim : s);
    real wl = (real)8.7266;
    imu[3 + a] = wimu[a] > wl ? wl : (wimu[a] < -wl ? -wl : wimu[a]);
  }
  for (int i = 0; i < 3; ++i) e->prev_lin_vel[i] = vimu[i]; /* kick_env.py:441,930 */
  /* compute_off_orn, kick_env.py:941-960 */
  const real goal_x = c->task == BEZ_TASK_KICK ? (real)c->goal[0] : e->goal[0], goal_y = c->task == BEZ_TASK_KICK ? (real)c->goal[1] : e->goal[1];
  real gx = goal_x - e->root_pos[0], gy = goal_y - e->root_pos[1];
  real gn = sqrt(gx * gx + gy * gy);
  real ux = gx / gn, uy = gy / gn;
  real yaw = euler_yaw(q);
  real hx = cos(yaw), hy = sin(yaw);
  real cosv = hx * ux + hy * uy;
  real sinv = fabs(ux * hy - uy * hx); /* || cross(u3, h3) || : unsigned (quirk Q4) */
  real o42 = sinv, o43 = -cosv;
  const real ang_goal = (real)c->goal_angle - wrap_pi(yaw); /* orient_env.py:719-735 compute_off_angle */
  if (c->task == BEZ_TASK_ORIENT) { o42 = cos(ang_goal); o43 = sin(ang_goal); }
  /* feet, kick_env.py:538-576 (cleats: kick_env.py:467-495) */
  if (m_cl(c)) {
    feet_cleats(e, e->feet);
  } else {
    real lf[4], rf[4];
    feet_no_cleats(e->contact_force[BEZ_LFOOT_BODY], lf);
    feet_no_cleats(e->contact_force[BEZ_RFOOT_BODY], rf);
    for (int i = 0; i < 4; ++i) { e->feet[i] = lf[i]; e->feet[4 + i] = rf[i]; }
  }
  /* compute_bez_observations, kick_env.py:1409-1415 (tail = constant ball_init, quirk Q5) */
  for (int j = 0; j < ND; ++j) { e->obs[j] = e->q[j]; e->obs[ND + j] = e->qd[j]; }
  for (int i = 0; i < 6; ++i) e->obs[36 + i] = imu[i];
  e->obs[42] = o42; e->obs[43] = o43;
  for (int i = 0; i < 8; ++i) e->obs[44 + i] = e->feet[i];
  e->obs[52] = c->ball_init[0]; e->obs[53] = c->ball_init[1]; /* bez_kick only; bez_walk / bez_orient stop at 52 (walk_env.py:1033-1050) */
  if (c->task != BEZ_TASK_KICK) {
    /* compute_bez_reward of walk_env.py:826-1031 / orient_env.py:843-1018 */
    real vn2 = 0, wn2 = 0, pn2 = 0;
    for (int i = 0; i < 3; ++i) { vn2 += vimu[i] * vimu[i]; wn2 += wimu[i] * wimu[i]; }
    for (int j = 0; j < ND; ++j) { real d = (real)(float)BEZ_DOF_DEFAULT[j] - e->q[j]; pn2 += d * d; }
    real vel_reward = sqrt(vn2 + wn2), vel_lin = sqrt(vn2), vel_ang = sqrt(wn2), pos_reward = sqrt(pn2);
    real up_proj = 1 - 2 * (q[0] * q[0] + q[1] * q[1]); /* get_basis_vector(q, (0,0,1)).z = quat_rotate [ext] */
    real dh = fabs(1 - up_proj);
    real height_vel_pos = -((vel_reward * (real)0.05 + pos_reward * (real)0.05) + dh);
    real reward, near;
    if (c->task == BEZ_TASK_WALK) {
      real vfwd = ux * vimu[0] + uy * vimu[1];
      real vel_height = vfwd * 10 - (dh + 5 * (pos_reward * (real)0.05));
      near = gn;
      reward = gn < (real)0.05 ? height_vel_pos : vel_height;
    } else {
      real vel_height = fabs(ang_goal) * (real)-0.5 - (dh + (real)0.05 * (pos_reward * (real)0.05));
      near = ang_goal; /* signed, as the reference compares it (orient_env.py:935) */
      reward = ang_goal < (real)0.05 ? height_vel_pos : vel_height;
    }
    int64_t reset = e->reset;
    if (up_proj < (real)0.7) { reset = 1; reward = -100; }
    int state = (near < (real)0.05) + (pos_reward < (real)0.15) + (vel_ang < (real)0.1) + (vel_lin < (real)0.1);
    if (state == 4) { reset = 1; reward = (real)1000.0 - (real)1000.0 * ((real)e->progress / (real)c->max_episode_length); }
    if (c->task == BEZ_TASK_WALK) { /* walk_env.py:966-990: heading to the goal turned by more than 90 deg since the start (0,0) */
      real a_init = atan2(goal_y / sqrt(goal_x * goal_x + goal_y * goal_y), goal_x / sqrt(goal_x * goal_x + goal_y * goal_y));
      if (fabs(a_init - atan2(uy, ux)) > (real)1.5708) { reset = 1; reward = -100; }
    } else { /* orient_env.py:1000-1009: wandered more than 0.3 m from the start */
      real tx = e->root_pos[0] - (real)c->bez_init[0], ty = e->root_pos[1] - (real)c->bez_init[1];
      if (sqrt(tx * tx + ty * ty) > (real)0.3) { reset = 1; reward = -5; }
    }
    if (e->progress >= c->max_episode_length) { reset = 1; reward = 0; }
    e->rew = reward;
    e->reset = reset;
    return;
  }

  /* compute_bez_reward, kick_env.py:1224-1391 */
  real bx = e->ball_pos[0], by = e->ball_pos[1];
  real dbx = bx - e->root_pos[0], dby = by - e->root_pos[1];
  real dbn = sqrt(dbx * dbx + dby * dby);
  real vel_fwd = (dbx / dbn) * vimu[0] + (dby / dbn) * vimu[1];
  real dgx = (real)c->goal[0] - bx, dgy = (real)c->goal[1] - by;
  real dgn = sqrt(dgx * dgx + dgy * dgy);
  real b2gx = dgx / dgn, b2gy = dgy / dgn;
  real ball_fwd = b2gx * e->ball_lin[0] + b2gy * e->ball_lin[1];
  real igx = (real)c->goal[0] - (real)c->ball_init[0], igy = (real)c->goal[1] - (real)c->ball_init[1];
  real ign = sqrt(igx * igx + igy * igy);
  real ang_now = atan2(b2gy, b2gx), ang_init = atan2(igy / ign, igx / ign);
  real goal_angle_diff = fabs(ang_init - ang_now);
  real vn = 0;
  for (int i = 0; i < 3; ++i) vn += vimu[i] * vimu[i] + wimu[i] * wimu[i];
  real vel_reward = sqrt(vn);
  real pn = 0;
  for (int j = 0; j < ND; ++j) { real d = (real)(float)BEZ_DOF_DEFAULT[j] - e->q[j]; pn += d * d; }
  real pos_reward = sqrt(pn);
  real height = fabs((real)0.325 - e->root_pos[2]);
  real kx = bx - (real)c->ball_init[0], ky = by - (real)c->ball_init[1];
  real kicked = sqrt(kx * kx + ky * ky);
  real vel_pos = vel_reward * (real)0.05 + pos_reward * (real)0.05;
  real height_vel_pos = height * 1 + vel_pos;
  real r_after = ball_fwd * (real)0.1 - height_vel_pos;
  real r_before = ball_fwd * (real)0.1 + (vel_fwd * (real)0.05 - height);
  real reward = kicked > (real)0.3 ? r_after : r_before;
  int64_t reset = e->reset;
  if (e->root_pos[2] < (real)0.275) { reset = 1; reward = -1; }
  real tx = e->root_pos[0] - (real)c->bez_init[0], ty = e->root_pos[1] - (real)c->bez_init[1];
  if (sqrt(tx * tx + ty * ty) > (real)0.5) { reset = 1; reward = -1; }
  if (goal_angle_diff > (real)1.5708) { reset = 1; reward = -1; }
  if (dgn < (real)0.05) { reset = 1; reward = (real)100.0 - (real)100.0 * ((real)e->progress / (real)c->max_episode_length); }
  if (e->progress >= c->max_episode_length) { reset = 1; reward = 0; }
  e->rew = reward;
  e->reset = reset;
}

static int oracle_use_prev(const Oracle* o) { return !(o->cfg.flags & BEZ_FLAG_IMU_PREV_ALIAS) || o->obs_calls == 0; }

static void env_post_physics(const BezSimConfig* c, Env* e, int64_t genv, int use_prev, const float* goal) {
  e->timeout = (e->progress >= c->max_episode_length - 1) ? 1 : 0; /* vec_task.py:331-332 */
  e->progress += 1;                                                /* kick_env.py:429 */
  if (e->reset != 0) env_reset(c, e, genv, goal);                  /* kick_env.py:433-435 */
  env_observe_reward(c, e, use_prev);                              /* kick_env.py:437-438 */
}

static void env_simulate(const BezSimConfig* c, Env* e) {
  real h = (real)c->dt / (real)c->substeps;
  const int last_only = (c->flags & BEZ_FLAG_CF_LAST_SUBSTEP) != 0;
  for (int s = 0; s < c->substeps; ++s) {
    if (last_only) substep(c, e, h, 1, 1);
    else substep(c, e, h, s == 0, (real)1 / (real)c->substeps);
  }
}

/* ------------------------------------------------------------------ domain randomisation (vec_task.py:505-725)
 * apply_randomizations is called from reset_idx (kick_env.py:781-782), i.e. once per control step in which some env resets,
 * after `randomize_buf += 1` (kick_env.py:430).  Restated for the whole batch in front of the env loop of that step:
 *   - an env with reset_buf != 0 and randomize_buf >= frequency redraws its friction / Kp / Kd / joint limits and clears
 *     randomize_buf (vec_task.py:525-530, 646-714); the draws are Philox words keyed by (seed, GLOBAL env id, episode) so
 *     that they depend neither on the shard nor on the order envs are visited in;
 *   - if any env resets and `frequency` frames have passed since the last time, gravity (vec_task.py:620-632) and the noise
 *     parameters of the observation / action lambdas (vec_task.py:544-618) are refreshed, scaled by the linear schedule
 *     min(frame, schedule_steps) / schedule_steps where frame = gym.get_frame_count (vec_task.py:521).
 * Word map of an env's draw: 0 friction; 1..18 stiffness; 19..36 damping; 37..72 lower (pairs: Box-Muller); 73..108 upper. */
static float dr_word_uniform(uint64_t seed, int64_t key, uint32_t key2, uint32_t tag, int k) {
  uint32_t c[4] = {(uint32_t)key, (uint32_t)((uint64_t)key >> 32), key2, tag + (uint32_t)(k >> 2)};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  return (float)(c[k & 3] >> 8) * (1.0f / 16777216.0f);
}
#define DR_TAG_ENV 0x44520000u
#define DR_TAG_GRAVITY 0x47520000u
static float dr_sched(const BezDrRange* r, uint64_t frame) {
  if (r->schedule_steps <= 0) return 1.0f;
  uint64_t f = frame < (uint64_t)r->schedule_steps ? frame : (uint64_t)r->schedule_steps;
  return (float)f / (float)r->schedule_steps;
}
static float dr_scaling(const BezDrRange* r, float s, float u) { /* uniform `scaling`: range interpolated from [1, 1] */
  float lo = fmaf(r->a, s, 1.0f - s), hi = fmaf(r->b, s, 1.0f - s);
  return fmaf(u, hi - lo, lo);
}
static float dr_normal(float u1, float u2) { return sqrtf(-2.0f * logf(1.0f - u1)) * cosf(6.2831853f * u2); }
static void dr_resample_env(Oracle* o, Env* e, int64_t genv) {
  const BezDrConfig* d = &o->dr;
  const uint64_t seed = o->cfg.seed, fr = o->dr_frame;
  if (d->friction.enabled) {
    float u = dr_word_uniform(seed, genv, e->episode, DR_TAG_ENV, 0);
    if (d->friction_buckets > 1) u = rintf(u * (float)(d->friction_buckets - 1)) / (float)(d->friction_buckets - 1);
    e->friction = (real)(o->cfg.plane_friction * dr_scaling(&d->friction, dr_sched(&d->friction, fr), u));
  }
  for (int j = 0; j < ND; ++j) {
    if (d->stiffness.enabled) e->kp_scale[j] = (real)dr_scaling(&d->stiffness, dr_sched(&d->stiffness, fr), dr_word_uniform(seed, genv, e->episode, DR_TAG_ENV, 1 + j));
    if (d->damping.enabled) e->kd_scale[j] = (real)dr_scaling(&d->damping, dr_sched(&d->damping, fr), dr_word_uniform(seed, genv, e->episode, DR_TAG_ENV, 19 + j));
    if (d->lower.enabled) {
      float s = dr_sched(&d->lower, fr), z = dr_normal(dr_word_uniform(seed, genv, e->episode, DR_TAG_ENV, 37 + 2 * j), dr_word_uniform(seed, genv, e->episode, DR_TAG_ENV, 38 + 2 * j));
      e->lim_lo[j] = (real)((float)BEZ_DOF_LOWER[j] + fmaf(z, d->lower.b * s, d->lower.a * s));
    }
    if (d->upper.enabled) {
      float s = dr_sched(&d->upper, fr), z = dr_normal(dr_word_uniform(seed, genv, e->episode, DR_TAG_ENV, 73 + 2 * j), dr_word_uniform(seed, genv, e->episode, DR_TAG_ENV, 74 + 2 * j));
      e->lim_hi[j] = (real)((float)BEZ_DOF_UPPER[j] + fmaf(z, d->upper.b * s, d->upper.a * s));
    }
  }
}
static void dr_refresh_global(Oracle* o) {
  const BezDrConfig* d = &o->dr;
  const uint64_t fr = o->dr_frame;
  if (d->gravity.enabled) {
    float s = dr_sched(&d->gravity, fr), g[3];
    for (int k = 0; k < 3; ++k) {
      float z = dr_normal(dr_word_uniform(o->cfg.seed, (int64_t)fr, 0, DR_TAG_GRAVITY, 2 * k), dr_word_uniform(o->cfg.seed, (int64_t)fr, 0, DR_TAG_GRAVITY, 2 * k + 1));
      g[k] = o->cfg.gravity[k] + fmaf(z, d->gravity.b * s, d->gravity.a * s);
    }
    for (int i = 0; i < o->n; ++i) for (int k = 0; k < 3; ++k) o->env[i].gravity[k] = (real)g[k];
  }
  float so = dr_sched(&d->observations, fr), sa = dr_sched(&d->actions, fr);
  o->dr_noise[0] = d->observations.enabled ? d->observations.a * so : 0.0f; o->dr_noise[1] = d->observations.enabled ? d->observations.b * so : 0.0f;
  o->dr_noise[2] = d->actions.enabled ? d->actions.a * sa : 0.0f; o->dr_noise[3] = d->actions.enabled ? d->actions.b * sa : 0.0f;
  o->dr_last_rand = fr;
}
/* in front of the env loop of every step that contains the post-physics */
static void oracle_dr_step(Oracle* o) {
  if (!o->dr_on) return;
  o->dr_frame += 1; /* gym.get_frame_count: this step's simulate has run */
  int any = 0;
  for (int i = 0; i < o->n; ++i) {
    Env* e = &o->env[i];
    int64_t rb = e->randomize + 1; /* kick_env.py:430 */
    if (e->reset != 0) {
      any = 1;
      if (rb >= o->dr.frequency) { dr_resample_env(o, e, o->cfg.env_id_offset + i); rb = 0; }
    }
    e->randomize = rb;
  }
  if (any && o->dr_frame - o->dr_last_rand >= (uint64_t)o->dr.frequency) dr_refresh_global(o);
}

/* ------------------------------------------------------------------ exported API (ctypes) */
void* bez_oracle_create(const BezSimConfig* cfg) {
  Oracle* o = (Oracle*)calloc(1, sizeof(Oracle));
  o->cfg = *cfg;
  if (!m_has_ball(cfg)) { /* walk_env.py / orient_env.py create no ball actor: the free body is parked out of reach */
    o->cfg.ball_init[0] = 1000.0f; o->cfg.ball_init[1] = 0.0f; o->cfg.ball_init[2] = (float)BEZ_BALL_RADIUS;
  }
  o->n = cfg->num_envs;
  o->env = (Env*)calloc((size_t)o->n, sizeof(Env));
  float goal0[2];
  goal_draw(cfg->seed, 0, 1, goal0); /* KickEnv/WalkEnv.__init__ ends with reset_idx(all): explicit reset call number 0 */
  o->reset_calls = 1;
  for (int i = 0; i < o->n; ++i) {
    Env* e = &o->env[i];
    e->reset = 1; /* vec_task.py:241 */
    e->friction = cfg->plane_friction;
    for (int j = 0; j < ND; ++j) { e->kp_scale[j] = 1; e->kd_scale[j] = 1; e->lim_lo[j] = (real)(float)BEZ_DOF_LOWER[j]; e->lim_hi[j] = (real)(float)BEZ_DOF_UPPER[j]; }
    for (int l = 0; l < NL; ++l) e->mass_scale[l] = 1;
    for (int k = 0; k < 3; ++k) e->gravity[k] = cfg->gravity[k];
    for (int k = 0; k < 8; ++k) e->feet[k] = -1; /* kick_env.py:185 */
    e->root_quat[3] = 1; e->ball_quat[3] = 1;
    /* vec_task.py:193 allocate_buffers sets reset_buf = 1, then KickEnv.__init__ ends with
     * reset_idx(all) (kick_env.py:238) which clears it: the first step() does NOT re-reset. */
    e->goal[0] = cfg->goal[0]; e->goal[1] = cfg->goal[1];
    env_reset(&o->cfg, e, cfg->env_id_offset + i, goal0);
  }
  return o;
}
void bez_oracle_destroy(void* h) { Oracle* o = (Oracle*)h; free(o->env); free(o); }
int bez_oracle_real_size(void) { return (int)sizeof(real); }

void bez_oracle_get_root_states(void* h, float* out) {
  Oracle* o = (Oracle*)h;
  const int na = m_has_ball(&o->cfg) ? 2 : 1; /* actors per env */
  for (int i = 0; i < o->n; ++i) {
    const Env* e = &o->env[i];
    float* r = out + (size_t)i * 13 * na;
    for (int k = 0; k < 3; ++k) { r[k] = (float)e->root_pos[k]; r[7 + k] = (float)e->root_lin[k]; r[10 + k] = (float)e->root_ang[k]; }
    for (int k = 0; k < 4; ++k) r[3 + k] = (float)e->root_quat[k];
    if (na == 1) continue;
    r += 13;
    for (int k = 0; k < 3; ++k) { r[k] = (float)e->ball_pos[k]; r[7 + k] = (float)e->ball_lin[k]; r[10 + k] = (float)e->ball_ang[k]; }
    for (int k = 0; k < 4; ++k) r[3 + k] = (float)e->ball_quat[k];
  }
}
void bez_oracle_set_root_states(void* h, const float* in) {
  Oracle* o = (Oracle*)h;
  const int na = m_has_ball(&o->cfg) ? 2 : 1;
  for (int i = 0; i < o->n; ++i) {
    Env* e = &o->env[i];
    const float* r = in + (size_t)i * 13 * na;
    for (int k = 0; k < 3; ++k) { e->root_pos[k] = r[k]; e->root_lin[k] = r[7 + k]; e->root_ang[k] = r[10 + k]; }
    for (int k = 0; k < 4; ++k) e->root_quat[k] = r[3 + k];
    if (na == 1) continue;
    r += 13;
    for (int k = 0; k < 3; ++k) { e->ball_pos[k] = r[k]; e->ball_lin[k] = r[7 + k]; e->ball_ang[k] = r[10 + k]; }
    for (int k = 0; k < 4; ++k) e->ball_quat[k] = r[3 + k];
  }
}
void bez_oracle_get_dof_state(void* h, float* out) {
  Oracle* o = (Oracle*)h;
  for (int i = 0; i < o->n; ++i) for (int j = 0; j < ND; ++j) { out[((size_t)i * ND + j) * 2] = (float)o->env[i].q[j]; out[((size_t)i * ND + j) * 2 + 1] = (float)o->env[i].qd[j]; }
}
void bez_oracle_set_dof_state(void* h, const float* in) {
  Oracle* o = (Oracle*)h;
  for (int i = 0; i < o->n; ++i) for (int j = 0; j < ND; ++j) { o->env[i].q[j] = in[((size_t)i * ND + j) * 2]; o->env[i].qd[j] = in[((size_t)i * ND + j) * 2 + 1]; }
}
void bez_oracle_get_targets(void* h, float* out) { Oracle* o = (Oracle*)h; for (int i = 0; i < o->n; ++i) for (int j = 0; j < ND; ++j) out[(size_t)i * ND + j] = (float)o->env[i].target[j]; }
void bez_oracle_set_targets(void* h, const float* in) { Oracle* o = (Oracle*)h; for (int i = 0; i < o->n; ++i) for (int j = 0; j < ND; ++j) o->env[i].target[j] = in[(size_t)i * ND + j]; }
void bez_oracle_get_contact_forces(void* h, float* out) {
  Oracle* o = (Oracle*)h;
  const int nbe = m_nbe(&o->cfg);
  for (int i = 0; i < o->n; ++i) for (int b = 0; b < nbe; ++b) for (int k = 0; k < 3; ++k) out[((size_t)i * nbe + b) * 3 + k] = (float)o->env[i].contact_force[b][k];
}
void bez_oracle_set_contact_forces(void* h, const float* in) {
  Oracle* o = (Oracle*)h;
  const int nbe = m_nbe(&o->cfg);
  for (int i = 0; i < o->n; ++i) for (int b = 0; b < nbe; ++b) for (int k = 0; k < 3; ++k) o->env[i].contact_force[b][k] = in[((size_t)i * nbe + b) * 3 + k];
}
void bez_oracle_get_goal(void* h, float* out) { Oracle* o = (Oracle*)h; for (int i = 0; i < o->n; ++i) for (int k = 0; k < 2; ++k) out[(size_t)i * 2 + k] = (float)o->env[i].goal[k]; }
void bez_oracle_set_goal(void* h, const float* in) { Oracle* o = (Oracle*)h; for (int i = 0; i < o->n; ++i) for (int k = 0; k < 2; ++k) o->env[i].goal[k] = in[(size_t)i * 2 + k]; }
int bez_oracle_num_bodies(void* h) { return m_nbe(&((Oracle*)h)->cfg); }
int bez_oracle_num_obs(void* h) { return m_nobs(&((Oracle*)h)->cfg); }
int bez_oracle_num_actors(void* h) { return m_has_ball(&((Oracle*)h)->cfg) ? 2 : 1; }
void bez_oracle_get_prev_lin_vel(void* h, float* out) { Oracle* o = (Oracle*)h; for (int i = 0; i < o->n; ++i) for (int k = 0; k < 3; ++k) out[(size_t)i * 3 + k] = (float)o->env[i].prev_lin_vel[k]; }
void bez_oracle_set_prev_lin_vel(void* h, const float* in) { Oracle* o = (Oracle*)h; for (int i = 0; i < o->n; ++i) for (int k = 0; k < 3; ++k) o->env[i].prev_lin_vel[k] = in[(size_t)i * 3 + k]; }
void bez_oracle_get_obs(void* h, float* out) { Oracle* o = (Oracle*)h; const int no = m_nobs(&o->cfg); for (int i = 0; i < o->n; ++i) for (int k = 0; k < no; ++k) out[(size_t)i * no + k] = (float)o->env[i].obs[k]; }
void bez_oracle_get_feet(void* h, float* out) { Oracle* o = (Oracle*)h; for (int i = 0; i < o->n; ++i) for (int k = 0; k < 8; ++k) out[(size_t)i * 8 + k] = (float)o->env[i].feet[k]; }
/* test aid (tests/: which envs sat within rounding of a speed-limit decision in the last control step) */
void bez_oracle_get_vlim_margin(void* h, float* out) { Oracle* o = (Oracle*)h; for (int i = 0; i < o->n; ++i) out[i] = (float)o->env[i].vmargin; }
void bez_oracle_get_rew(void* h, float* out) { Oracle* o = (Oracle*)h; for (int i = 0; i < o->n; ++i) out[i] = (float)o->env[i].rew; }
void bez_oracle_get_reset(void* h, int64_t* out) { Oracle* o = (Oracle*)h; for (int i = 0; i < o->n; ++i) out[i] = o->env[i].reset; }
void bez_oracle_set_reset(void* h, const int64_t* in) { Oracle* o = (Oracle*)h; for (int i = 0; i < o->n; ++i) o->env[i].reset = in[i]; }
void bez_oracle_get_progress(void* h, int64_t* out) { Oracle* o = (Oracle*)h; for (int i = 0; i < o->n; ++i) out[i] = o->env[i].progress; }
void bez_oracle_set_progress(void* h, const int64_t* in) { Oracle* o = (Oracle*)h; for (int i = 0; i < o->n; ++i) o->env[i].progress = in[i]; }
void bez_oracle_get_timeout(void* h, int64_t* out) { Oracle* o = (Oracle*)h; for (int i = 0; i < o->n; ++i) out[i] = o->env[i].timeout; }
void bez_oracle_set_env_params(void* h, int param, const float* v) {
  Oracle* o = (Oracle*)h;
  for (int i = 0; i < o->n; ++i) {
    Env* e = &o->env[i];
    switch (param) {
      case BEZ_PARAM_FRICTION: e->friction = v ? v[i] : o->cfg.plane_friction; break;
      case BEZ_PARAM_KP_SCALE: for (int j = 0; j < ND; ++j) e->kp_scale[j] = v ? v[(size_t)i * ND + j] : 1; break;
      case BEZ_PARAM_KD_SCALE: for (int j = 0; j < ND; ++j) e->kd_scale[j] = v ? v[(size_t)i * ND + j] : 1; break;
      case BEZ_PARAM_MASS_SCALE: for (int l = 0; l < NL; ++l) e->mass_scale[l] = v ? v[(size_t)i * NL + l] : 1; break;
      case BEZ_PARAM_GRAVITY: for (int k = 0; k < 3; ++k) e->gravity[k] = v ? v[(size_t)i * 3 + k] : o->cfg.gravity[k]; break;
      case BEZ_PARAM_DOF_LOWER: for (int j = 0; j < ND; ++j) e->lim_lo[j] = v ? v[(size_t)i * ND + j] : (real)(float)BEZ_DOF_LOWER[j]; break;
      case BEZ_PARAM_DOF_UPPER: for (int j = 0; j < ND; ++j) e->lim_hi[j] = v ? v[(size_t)i * ND + j] : (real)(float)BEZ_DOF_UPPER[j]; break;
      default: break;
    }
  }
}
/* rigid body states (N*22,13) by forward kinematics; velocities of body origins, world frame */
void bez_oracle_get_rigid_body_states(void* h, float* out) {
  Oracle* o = (Oracle*)h;
  for (int i = 0; i < o->n; ++i) {
    const Env* e = &o->env[i];
    Kin k;
    forward_kinematics(&o->cfg, e, &k);
    SV V[NL];
    V[0] = sv(v3(e->root_ang[0], e->root_ang[1], e->root_ang[2]), v3(e->root_lin[0], e->root_lin[1], e->root_lin[2]));
    for (int l = 1; l < NL; ++l) V[l] = sv_add(V[BEZ_LINK_PARENT[l]], sv_scale(sv(k.a[l], v3cross(k.r[l], k.a[l])), e->qd[l - 1]));
    const int nb = m_nb(&o->cfg), nbe = m_nbe(&o->cfg);
    for (int b = 0; b < nb; ++b) {
      int l = m_body_link(&o->cfg, b);
      V3 off = v3((real)m_body_offset(&o->cfg, b)[0], (real)m_body_offset(&o->cfg, b)[1], (real)m_body_offset(&o->cfg, b)[2]);
      V3 x = v3add(k.r[l], m3mulv(&k.E[l], off));
      V3 vel = v3add(sv_lin(V[l]), v3cross(sv_ang(V[l]), x));
      real qq[4];
      mat_to_quat(&k.E[l], qq);
      float* r = out + ((size_t)i * nbe + b) * 13;
      for (int a = 0; a < 3; ++a) { r[a] = (float)(e->root_pos[a] + x.v[a]); r[7 + a] = (float)vel.v[a]; r[10 + a] = (float)V[l].v[a]; }
      for (int a = 0; a < 4; ++a) r[3 + a] = (float)qq[a];
    }
    if (nbe == nb) continue;
    float* r = out + ((size_t)i * nbe + nb) * 13;
    for (int a = 0; a < 3; ++a) { r[a] = (float)e->ball_pos[a]; r[7 + a] = (float)e->ball_lin[a]; r[10 + a] = (float)e->ball_ang[a]; }
    for (int a = 0; a < 4; ++a) r[3 + a] = (float)e->ball_quat[a];
  }
}

void bez_oracle_pre_physics(void* h, const float* actions) {
  Oracle* o = (Oracle*)h;
  for (int i = 0; i < o->n; ++i) env_pre_physics(&o->cfg, &o->env[i], actions + (size_t)i * ND);
}
void bez_oracle_simulate(void* h) {
  Oracle* o = (Oracle*)h;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < o->n; ++i) env_simulate(&o->cfg, &o->env[i]);
}
void bez_oracle_post_physics(void* h) {
  Oracle* o = (Oracle*)h;
  oracle_dr_step(o);
  int up = oracle_use_prev(o);
  float goal[2];
  goal_draw(o->cfg.seed, o->post_calls++, 0, goal);
  for (int i = 0; i < o->n; ++i) env_post_physics(&o->cfg, &o->env[i], o->cfg.env_id_offset + i, up, goal);
  o->obs_calls += 1;
}
/* obs + reward only (no progress increment / reset handling): golden-vector checks of the jit functions */
void bez_oracle_observe_reward(void* h) {
  Oracle* o = (Oracle*)h;
  int up = oracle_use_prev(o);
  for (int i = 0; i < o->n; ++i) env_observe_reward(&o->cfg, &o->env[i], up);
  o->obs_calls += 1;
}
void bez_oracle_step(void* h, const float* actions) {
  Oracle* o = (Oracle*)h;
  oracle_dr_step(o);
  int up = oracle_use_prev(o);
  float goal[2];
  goal_draw(o->cfg.seed, o->post_calls++, 0, goal);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < o->n; ++i) {
    Env* e = &o->env[i];
    env_pre_physics(&o->cfg, e, actions + (size_t)i * ND);
    env_simulate(&o->cfg, e);
    env_post_physics(&o->cfg, e, o->cfg.env_id_offset + i, up, goal);
  }
  o->obs_calls += 1;
}
void bez_oracle_reset_idx(void* h, const int32_t* ids, int n) {
  Oracle* o = (Oracle*)h;
  float goal[2];
  goal_draw(o->cfg.seed, o->reset_calls++, 1, goal);
  for (int k = 0; k < n; ++k) env_reset(&o->cfg, &o->env[ids[k]], o->cfg.env_id_offset + ids[k], goal);
}
/* bez_sim_set_randomization: NULL = off.  The first call randomises every env at frame 0 (first_randomization, vec_task.py:521-523) */
void bez_oracle_set_randomization(void* h, const BezDrConfig* dr) {
  Oracle* o = (Oracle*)h;
  if (!dr) { o->dr_on = 0; return; }
  o->dr = *dr; o->dr_on = 1; o->dr_frame = 0;
  for (int i = 0; i < o->n; ++i) { dr_resample_env(o, &o->env[i], o->cfg.env_id_offset + i); o->env[i].randomize = 0; }
  dr_refresh_global(o);
}
void bez_oracle_get_randomize(void* h, int64_t* out) { Oracle* o = (Oracle*)h; for (int i = 0; i < o->n; ++i) out[i] = o->env[i].randomize; }
void bez_oracle_set_randomize(void* h, const int64_t* in) { Oracle* o = (Oracle*)h; for (int i = 0; i < o->n; ++i) o->env[i].randomize = in[i]; }
void bez_oracle_get_dr_noise(void* h, float* out) { Oracle* o = (Oracle*)h; for (int k = 0; k < 4; ++k) out[k] = o->dr_noise[k]; }
void bez_oracle_get_env_params(void* h, int param, float* v) {
  Oracle* o = (Oracle*)h;
  for (int i = 0; i < o->n; ++i) {
    const Env* e = &o->env[i];
    switch (param) {
      case BEZ_PARAM_FRICTION: v[i] = (float)e->friction; break;
      case BEZ_PARAM_KP_SCALE: for (int j = 0; j < ND; ++j) v[(size_t)i * ND + j] = (float)e->kp_scale[j]; break;
      case BEZ_PARAM_KD_SCALE: for (int j = 0; j < ND; ++j) v[(size_t)i * ND + j] = (float)e->kd_scale[j]; break;
      case BEZ_PARAM_MASS_SCALE: for (int l = 0; l < NL; ++l) v[(size_t)i * NL + l] = (float)e->mass_scale[l]; break;
      case BEZ_PARAM_GRAVITY: for (int k = 0; k < 3; ++k) v[(size_t)i * 3 + k] = (float)e->gravity[k]; break;
      case BEZ_PARAM_DOF_LOWER: for (int j = 0; j < ND; ++j) v[(size_t)i * ND + j] = (float)e->lim_lo[j]; break;
      case BEZ_PARAM_DOF_UPPER: for (int j = 0; j < ND; ++j) v[(size_t)i * ND + j] = (float)e->lim_hi[j]; break;
      default: break;
    }
  }
}
void bez_oracle_seed(void* h, uint64_t seed) { ((Oracle*)h)->cfg.seed = seed; }
void bez_oracle_set_flags(void* h, uint32_t flags) { /* the asset bits are fixed at creation, as in bez_sim_set_flags */
  Oracle* o = (Oracle*)h;
  const uint32_t asset = BEZ_FLAG_CLEATS | BEZ_FLAG_BOX_ASSET;
  o->cfg.flags = (flags & ~asset) | (o->cfg.flags & asset);
}
void bez_oracle_set_obs_calls(void* h, int64_t n) { ((Oracle*)h)->obs_calls = n; }

/* Known-answer hooks: bare dynamics of env 0 in double-precision interface.
 * mode 1: pure ABA with joint torques tau (no PD, friction, limits, armature, contact).
 * Outputs the root spatial acceleration about the torso origin (world axes, [ang; lin], spatial not
 * classical) and joint accelerations. */
void bez_oracle_forward_dynamics(void* h, int env, int mode, const double* tau, double* a0, double* qdd,
                                 double* ball_acc6, double* contact_force) {
  Oracle* o = (Oracle*)h;
  real t[ND];
  for (int j = 0; j < ND; ++j) t[j] = tau ? (real)tau[j] : 0;
  Dyn d;
  real hstep = (real)o->cfg.dt / (real)o->cfg.substeps;
  dynamics(&o->cfg, &o->env[env], hstep, mode, t, &d);
  for (int i = 0; i < 6; ++i) a0[i] = d.a0.v[i];
  for (int j = 0; j < ND; ++j) qdd[j] = d.qdd[j];
  if (ball_acc6) for (int i = 0; i < 3; ++i) { ball_acc6[i] = d.ball_ang_acc.v[i]; ball_acc6[3 + i] = d.ball_lin_acc.v[i]; }
  if (contact_force) for (int b = 0; b < BEZ_NBE; ++b) for (int k = 0; k < 3; ++k) contact_force[b * 3 + k] = d.contact_force[b][k]; /* default-asset layout (known-answer tests) */
}
/* full-precision state injection for the known-answer tests: s = pos3 quat4 lin3 ang3 q18 qd18 (49 doubles) */
void bez_oracle_set_env_state_f64(void* h, int env, const double* s) {
  Env* e = &((Oracle*)h)->env[env];
  for (int i = 0; i < 3; ++i) { e->root_pos[i] = (real)s[i]; e->root_lin[i] = (real)s[7 + i]; e->root_ang[i] = (real)s[10 + i]; }
  for (int i = 0; i < 4; ++i) e->root_quat[i] = (real)s[3 + i];
  for (int j = 0; j < ND; ++j) { e->q[j] = (real)s[13 + j]; e->qd[j] = (real)s[31 + j]; }
}
uint32_t bez_oracle_philox_word(uint64_t seed, int64_t genv, uint32_t episode, int k) {
  uint32_t c[4] = {(uint32_t)genv, (uint32_t)((uint64_t)genv >> 32), episode, (uint32_t)(k >> 2)};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  return c[k & 3];
}

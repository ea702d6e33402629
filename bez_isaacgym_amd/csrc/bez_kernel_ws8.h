// bez_kernel_ws8.h -- wave-specialised fused step kernel, 8 role waves per 64 environments (2 waves per SIMD).
//
// A lone wave issues one vector instruction every ~5 cycles on a CDNA4 SIMD whatever the other SIMDs do, so the step time of
// a kernel with one wave per chain (round 1's 4-wave kernel) is the instruction count of its longest role: a leg (forward kinematics, link
// inertias, ball/box tests, foot ground contact, the articulated-inertia recursion, the leg<->leg correction, pass 3).  Here
// the leg roles keep only what is serial in the joint chain, and everything that merely NEEDS the leg's kinematics is
// recomputed from the published joint state by helper waves that share the SIMDs (a second wave on a SIMD issues in the slots
// the first leaves empty):
//     role 0  left leg   (links 5..10)    pass 1 (kinematics, inertias), foot ground contact, pass 2, leg<->leg correction, pass 3
//     role 1  right leg  (links 13..18)   same
//     role 2  ball candidates among the hip / thigh / calf boxes of both legs (own forward kinematics down to the calves), then the
//             head (links 1,2) as a chain of its own
//     role 3  torso, ball, ball<->torso-box candidate, 6x6 root solve, integration, post-physics
//     role 4  ball candidate among the ankle / foot boxes of the left leg (own forward kinematics of the leg) -> X_CAND; then the left arm (links 3,4) as a chain of its own; then the sum of the head / arm blocks
//     role 5  the same for the right leg and the right arm (links 11,12)
//     role 6  leg<->leg capsule pairs of the left hip/thigh/calf capsules (own forward kinematics of both legs)
//     role 7  leg<->leg capsule pairs of the left ankle/foot capsules
// The deepest of the three ball candidates wins (left, right, torso on ties, the box order of the oracle); every owner decides
// locally from the three published depths whether it is the winner, and only the winner evaluates the contact operands (a leg
// wave skips that block altogether unless one of its 64 envs has a ball<->leg contact).
// All roles execute the same barriers (B0, per substep B1 B1c B2 B3 B4 B5).  512 threads, <= 256 VGPRs per wave.
// The action / observation staging block aliases the X_IA slots (actions are consumed before the first X_IA store, the
// observation rows are staged after the last X_IA load).
#pragma once
#include "bez_kernels.h"

// BEZ_WS_SUB: lanes per env.  1 (bez_step_ws8.hip, namespace w8): lane l of every role wave works on env l of a 64-env workgroup.
// 4 (bez_step_ws8q.hip, namespace w8q; the default for the default asset without per-env parameters, bez_sim.hip kernel_from_env):
// lane l works on env l / 4 of a 16-env workgroup -- 256 workgroups at 4096 envs, every CU of the chip -- and the four lanes of a
// quad split the 6 x 6 work of the legs' passes between them (bez_ws_quad.inc: 2 x 2 blocks of 3 x 3, quad_perm DPP); whatever is
// not split runs redundantly in the quad's lanes, whose LDS columns stay private (slot stride WS_LANES), so the roles' hand-overs
// are the same in both forms.
#ifndef BEZ_WS_SUB
#define BEZ_WS_SUB 1
#endif
namespace bez {
#if BEZ_WS_SUB == 1
namespace w8 {
#else
namespace w8q {
#endif

constexpr int WS_BLOCK = 512;
constexpr int WS_SUB = BEZ_WS_SUB;        // lanes per env
constexpr int WS_LANES = 64;              // lanes per wave = stride of an LDS slot
constexpr int WS_ENVS = WS_LANES / WS_SUB;  // envs per workgroup
static_assert(WS_SUB == 1 || WS_SUB == 4, "one lane or one quad per env");
constexpr int WS_SUB_SHIFT = WS_SUB == 4 ? 2 : 0, WS_ENV_SHIFT = 6 - WS_SUB_SHIFT;
BEZ_DEV constexpr int ws_env_of(int lane) { return lane >> WS_SUB_SHIFT; }   // workgroup-local env of a lane

enum : int {
  X_ROOT = 0,      // pos3 quat4 lin3 ang3
  X_BALL = 13,     // pos3 lin3 ang3
  X_A0 = 22,       // torso spatial acceleration
  X_FL = 28,       // ball<->link force on the link (3) + contact point rel. ball centre (3)
  X_PSUM = 34,     // per chain-owning role (0,1,2,4,5): sum of (default - q)^2 over its joints; slot 3 (no chain): X_RESETF
  X_CAND = 40,     // per leg: depth, link, n(3), P(3), V(6) of its deepest ball/box candidate among the lower boxes (roles 4 / 5) = 14 x 2
  X_TORSO = 68,    // depth of the torso-box candidate (role 3)
  X_SELFF2 = 69,   // per helper part: sum of |force|^2 over its active leg<->leg pairs (2)
  X_PAIRSEQ = 71,  // lanes 0 / 1 of this slot: the two leg waves' sequence words (ws_pair_publish); lanes 2 / 3: the package hand-over of the lane-group form
  X_SPARE = 72,    // 72..80 unused (lane-group form: the torso's contact rows across the root solve, 72..76)
  X_LEGQ = 81,     // per leg: q(6) qd(6) at the start of the substep (read by the helper roles)
  X_SELF = 105,    // 2 helper parts x [per leg box link (left 5, right 5): bias wrench (6) + reported force (3)] = 2 x 90
  X_CF = 285,      // net contact force rows, up to BEZ_NBE_MAX = 30 bodies (mean over substeps)
  X_HIT = 375,     // cleats asset only: per leg 4 ground-point records x 8 floats (x3 fn0 kn ct ftx0 fty0)
  X_IA = 439,      // 5 chains (left leg, right leg, head, left arm, right arm) x (Sym6 21 + bias 6); block 2 ends up holding head + arms
  X_BCN = 574,     // per leg: contact rows of the foot (B 9, C 6, F0 3), parked here across pass 2 (register pressure)
  X_CANDH = 610,   // per leg: the same 14 floats for the upper boxes (hip / thigh links), from role 2
  X_SLOTS = 638,
  X_RESETF = X_PSUM + 3,  // 1.0 where this step resets the env: its contact rows leave the kernel as zeros (written by role 7, read by the copy-out)
  X_SELFSUM = X_PSUM     // inside the substep loop (the pose-error slots are written after it): per leg the two joint sums of the leg<->leg contact scale
};
#ifndef BEZ_W8_CAND_SPLIT
#define BEZ_W8_CAND_SPLIT 4
#endif
#ifndef BEZ_W8_LEG_BAR
#define BEZ_W8_LEG_BAR 2
#endif
#ifndef BEZ_W8_SELF_BAR
#define BEZ_W8_SELF_BAR 3
#endif
constexpr int CAND_SPLIT_CL = BEZ_W8_CAND_SPLIT;
// default asset: role 2 tests leg links 0..2 (hip_side, hip_front, thigh), roles 4 / 5 links 3..4 (calf, ankle), the leg role its
// own foot box; cleats asset (X_HIT in use): role 2 links 0..3, roles 4 / 5 links 4..5
template <bool CL> BEZ_DEV constexpr int cand_split() { return CL ? CAND_SPLIT_CL : 3; }
template <bool CL> BEZ_DEV constexpr int cand_lo_end() { return CL ? 6 : 5; }
// The default asset has no cleat records: its kernels use the X_HIT slots for a third partial ball candidate per leg (the foot
// box, tested by the leg role itself: 2 x 14 floats).
constexpr int X_CANDF = X_HIT;
// ... and, in the lane-group form, for the link packages the arm roles hand to the legs (bez_ws_quad.inc: 2 x 16 slots behind the candidates)
constexpr int X_PKG = X_HIT + 28;
static_assert(X_PKG + 32 <= X_HIT + 64, "packages fit the cleat records' slots");
constexpr int X_STAGE = X_IA;  // staging aliases the chain blocks (see the header comment)
constexpr int WS_ACT_STRIDE = 19;
constexpr int WS_LDS_FLOATS = X_SLOTS * WS_LANES;
static_assert(WS_LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
static_assert(5 * 27 >= 54 && 5 * 27 >= WS_ACT_STRIDE, "staging fits the aliased block");

// joints (dof index) of the chain-owning roles
BEZ_DEV constexpr int role_ndof(int role) { return role <= 1 ? 6 : 2; }
BEZ_DEV constexpr int role_dof(int role, int i) { return role == 0 ? 4 + i : (role == 1 ? 12 + i : (role == 2 ? i : (role == 4 ? 2 + i : 10 + i))); }
// leg<->leg pairs by their LEFT capsule: part 0 = hip / thigh / calf capsules (0,1,2: 13 pairs), part 1 = ankle / foot (3,4,5: 15 pairs)
BEZ_DEV constexpr bool self_part_owns(int part, int ia, int) { return (part == 0) == (ia <= 2); }
BEZ_DEV constexpr int self_fk_depth(int part, int side) { return (part == 0 && side == 0) ? 4 : 6; }

#include "bez_ws_common.inc"
#if BEZ_WS_SUB == 4
#include "bez_ws_quad.inc"
#endif

BEZ_DEV void xs_load_sym6(const float* lds, int lane, int slot, Sym6& I, SV& p) {
  float* f = (float*)&I;
#pragma unroll
  for (int i = 0; i < 21; ++i) f[i] = XS(slot + i);
  p = xs_load_sv(lds, lane, slot + 21);
}
BEZ_DEV void xs_store_body_contact(float* lds, int lane, int slot, const BodyContact& b) {
  const float* fb = (const float*)&b.B;
#pragma unroll
  for (int i = 0; i < 9; ++i) XS(slot + i) = fb[i];
  const float* fc = (const float*)&b.C;
#pragma unroll
  for (int i = 0; i < 6; ++i) XS(slot + 9 + i) = fc[i];
  xs_store_v3(lds, lane, slot + 15, b.F0);
}
BEZ_DEV BodyContact xs_load_body_contact(const float* lds, int lane, int slot) {
  BodyContact b;
  float* fb = (float*)&b.B;
#pragma unroll
  for (int i = 0; i < 9; ++i) fb[i] = XS(slot + i);
  float* fc = (float*)&b.C;
#pragma unroll
  for (int i = 0; i < 6; ++i) fc[i] = XS(slot + 9 + i);
  b.F0 = xs_load_v3(lds, lane, slot + 15);
  return b;
}

// ---- the ball candidates.  Each leg has two partial candidates (upper boxes from role 2, lower boxes from roles 4 / 5); the
// boxes are ordered hip -> foot and an earlier box keeps a tie, so the lower one wins only if strictly deeper.  Then: left leg,
// right leg, torso -- the right leg needs strictly more depth than the left, the torso strictly more than the better leg.
struct CandDepths { float dl, dr, dt; int bl, br; };  // bl / br: slot base of the leg's winning partial candidate
template <bool CL>
BEZ_DEV CandDepths load_cand_depths(const float* lds, int lane) {
  CandDepths c;
  const float lh = XS(X_CANDH), ll = XS(X_CAND), rh = XS(X_CANDH + 14), rl = XS(X_CAND + 14);
  c.bl = (ll > lh) ? X_CAND : X_CANDH; c.dl = (ll > lh) ? ll : lh;
  c.br = (rl > rh) ? X_CAND + 14 : X_CANDH + 14; c.dr = (rl > rh) ? rl : rh;
  if (!CL) {  // the foot boxes, last in the box order
    const float lf = XS(X_CANDF), rf = XS(X_CANDF + 14);
    if (lf > c.dl) { c.bl = X_CANDF; c.dl = lf; }
    if (rf > c.dr) { c.br = X_CANDF + 14; c.dr = rf; }
  }
  c.dt = XS(X_TORSO);
  return c;
}
// 0 left, 1 right, 2 torso
BEZ_DEV int cand_winner(const CandDepths& c) { const int sw = (c.dr > c.dl) ? 1 : 0; return (c.dt > (sw ? c.dr : c.dl)) ? 2 : sw; }
BEZ_DEV void publish_cand(float* lds, int lane, int c0, const BallSel& sel, SV Vsel) {
  XS(c0) = sel.depth; XS(c0 + 1) = (float)sel.link;
  xs_store_v3(lds, lane, c0 + 2, sel.n); xs_store_v3(lds, lane, c0 + 5, sel.P); xs_store_sv(lds, lane, c0 + 8, Vsel);
}
// Forward kinematics of leg links [0, NFK) from the published joint state, keeping the frames of the box links [B0, NFK); the
// box tests follow separately because the ball's new state is published later than the root's (deferred ball update).
template <int B0, int NFK>
struct LegFrames { M3 E[NFK - B0]; V3 r[NFK - B0]; SV V[NFK - B0]; };
template <int LEG_FIRST, int B0, int NFK>
BEZ_DEV void leg_frames(const float* lds, int lane, int side, const M3& E0, SV V0, LegFrames<B0, NFK>& F, float quirk_z = BEZ_QUIRK_RZ) {
  M3 E = E0; V3 r = mk(0, 0, 0); SV V = V0, Sj, cbj;
  static_for<NFK>([&](auto I) {
    constexpr int i = decltype(I)::value;
    link_kinematics<LEG_FIRST + i>(XS(X_LEGQ + side * 12 + i), XS(X_LEGQ + side * 12 + 6 + i), E, r, V, Sj, cbj, quirk_z);
    if constexpr (i >= B0) { F.E[i - B0] = E; F.r[i - B0] = r; F.V[i - B0] = V; }
  });
}
template <int LEG_FIRST, int B0, int NFK>
BEZ_DEV void leg_box_tests(const LegFrames<B0, NFK>& F, V3 bc, float* lds, int lane, int c0) {
  BallSel sel;
  sel.link = -1; sel.depth = 0.f; sel.n = sel.P = sel.f0p = sel.x = sel.xb = mk(0, 0, 0); sel.A = sym3zero();
  SV Vsel = svzero();
  static_for<NFK - B0>([&](auto I) {
    constexpr int k = decltype(I)::value;
    constexpr int L = LEG_FIRST + B0 + k;
    if constexpr (link_has_box(L)) {
      test_box<link_box(L)>(F.E[k], F.r[k], bc, sel);
      if (sel.link == L) Vsel = F.V[k];  // this box just became the deepest candidate: keep its link velocity
    }
  });
  publish_cand(lds, lane, c0, sel, Vsel);
}
struct RootOnly { M3 E0; SV V0; V3 pos; };
BEZ_DEV RootOnly load_root_only(const float* lds, int lane) {
  RootOnly R;
  R.pos = xs_load_v3(lds, lane, X_ROOT);
  R.E0 = quat_to_mat(XS(X_ROOT + 3), XS(X_ROOT + 4), XS(X_ROOT + 5), XS(X_ROOT + 6));
  R.V0 = mksv(xs_load_v3(lds, lane, X_ROOT + 10), xs_load_v3(lds, lane, X_ROOT + 7));
  return R;
}

// per-env DR scalars every role may need
template <bool DR>
BEZ_DEV ChainDyn load_chain_dyn(const Params& P, int e) {
  ChainDyn D; D.mu = P.mu; D.g = mk(P.g[0], P.g[1], P.g[2]);
  if (DR) {
    if (P.dr_friction) D.mu = P.dr_friction[e];
    if (P.dr_gravity) {
      if (P.dr_gravity_uniform) D.g = mk(P.dr_gravity[0], P.dr_gravity[1], P.dr_gravity[2]);   // uniform address: scalar loads, no VGPR per lane
      else D.g = mk(P.dr_gravity[(size_t)e * 3], P.dr_gravity[(size_t)e * 3 + 1], P.dr_gravity[(size_t)e * 3 + 2]);
    }
  }
  return D;
}

// Inside the substep loop: the same scalars behind an opaque copy, so that products of loop-invariant UNIFORM values (mass x
// gravity per link, ...) are recomputed per substep instead of being hoisted out of the loop into long-lived VGPRs.
BEZ_DEV ChainDyn in_loop(const ChainDyn& D) { ChainDyn d = D; pin(d.g); pin(d.mu); return d; }

// joint state / gains of NJ consecutive joints starting at link FIRST (dof FIRST-1)
template <int FIRST, int NJ, bool DR>
BEZ_DEV void load_joints(const Params& P, int e, float* q, float* qd, float* kps, float* kds, float* ms, float* lo, float* hi) {
  const int n = P.n;
  const float* st = P.state;
#pragma unroll
  for (int i = 0; i < NJ; ++i) {
    constexpr int d0 = FIRST - 1;
    q[i] = st[(size_t)(F_Q + d0 + i) * n + e]; qd[i] = st[(size_t)(F_QD + d0 + i) * n + e];
    kps[i] = 1.f; kds[i] = 1.f; ms[i] = 1.f; lo[i] = (float)BEZ_DOF_LOWER[d0 + i]; hi[i] = (float)BEZ_DOF_UPPER[d0 + i];
    if (DR) {
      if (P.dr_pack) {   // one 16-byte load per joint instead of four 4-byte ones (the same numbers: bez_sim.hip keeps the pack current)
        const float4 v = P.dr_pack[(size_t)e * BEZ_ND + d0 + i];
        kps[i] = v.x; kds[i] = v.y; lo[i] = v.z; hi[i] = v.w;
      } else {
        if (P.dr_kp) kps[i] = P.dr_kp[(size_t)e * BEZ_ND + d0 + i];
        if (P.dr_kd) kds[i] = P.dr_kd[(size_t)e * BEZ_ND + d0 + i];
        if (P.dr_lower) lo[i] = P.dr_lower[(size_t)e * BEZ_ND + d0 + i];
        if (P.dr_upper) hi[i] = P.dr_upper[(size_t)e * BEZ_ND + d0 + i];
      }
      if (P.dr_mass) ms[i] = P.dr_mass[(size_t)e * BEZ_NL + FIRST + i];
    }
  }
}
// the per-env (domain-randomisation) part alone.  The leg roles call this INSIDE the substep loop with an opaque env index: 30
// values per leg that are only read once per joint would otherwise live in registers through the whole physics (the
// 256-VGPR roles spill); re-fetched per substep the loads sit next to their uses and hit in L2.
template <int FIRST, int NJ, bool DR>
BEZ_DEV void load_joint_params(const Params& P, int e, float* kps, float* kds, float* ms, float* lo, float* hi) {
#pragma unroll
  for (int i = 0; i < NJ; ++i) {
    constexpr int d0 = FIRST - 1;
    kps[i] = 1.f; kds[i] = 1.f; ms[i] = 1.f; lo[i] = (float)BEZ_DOF_LOWER[d0 + i]; hi[i] = (float)BEZ_DOF_UPPER[d0 + i];
    if (DR) {
      if (P.dr_pack) {   // one 16-byte load per joint instead of four 4-byte ones (the same numbers: bez_sim.hip keeps the pack current)
        const float4 v = P.dr_pack[(size_t)e * BEZ_ND + d0 + i];
        kps[i] = v.x; kds[i] = v.y; lo[i] = v.z; hi[i] = v.w;
      } else {
        if (P.dr_kp) kps[i] = P.dr_kp[(size_t)e * BEZ_ND + d0 + i];
        if (P.dr_kd) kds[i] = P.dr_kd[(size_t)e * BEZ_ND + d0 + i];
        if (P.dr_lower) lo[i] = P.dr_lower[(size_t)e * BEZ_ND + d0 + i];
        if (P.dr_upper) hi[i] = P.dr_upper[(size_t)e * BEZ_ND + d0 + i];
      }
      if (P.dr_mass) ms[i] = P.dr_mass[(size_t)e * BEZ_NL + FIRST + i];
    }
  }
}
// position targets of NJ joints from the staged action row (kick_env.py:410-419) or from the state (physics-only entry point)
template <int FIRST, int NJ, bool PRE, bool HEAD>
BEZ_DEV void load_targets(const Params& P, const float* lds, int lane, int e, float* target) {
  if (PRE) {
    const float* act = lds + X_STAGE * WS_LANES + ws_env_of(lane) * WS_ACT_STRIDE;
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
      constexpr int d0 = FIRST - 1;
      float a = fminf(fmaxf(act[d0 + i], -P.clip), P.clip);
      if (HEAD) a = 0.f;  // head frozen (kick_env.py:414)
      float t = a + (float)BEZ_DOF_DEFAULT[d0 + i];
      target[i] = fmaxf(fminf(t, (float)BEZ_DOF_UPPER[d0 + i]), (float)BEZ_DOF_LOWER[d0 + i]);
    }
  } else {
#pragma unroll
    for (int i = 0; i < NJ; ++i) target[i] = P.state[(size_t)(F_TARGET + FIRST - 1 + i) * P.n + e];
  }
}

// ------------------------------------------------------------------------------------------------ roles
template <int FIRST, bool PRE, bool POST, bool DR, bool CL>
BEZ_DEV void leg_role(const Params& P, float* lds, int lane, int e, bool active, int side) {
  constexpr int LEN = 6;
  float q[LEN], qd[LEN], target[LEN];
  {
    float kps[LEN], kds[LEN], ms[LEN], lo[LEN], hi[LEN];  // (re-fetched inside the substep loop)
    load_joints<FIRST, LEN, false>(P, e, q, qd, kps, kds, ms, lo, hi);
  }
#pragma unroll
  for (int i = 0; i < LEN; ++i) { XS(X_LEGQ + side * 12 + i) = q[i]; XS(X_LEGQ + side * 12 + 6 + i) = qd[i]; }
  const ChainDyn D0 = load_chain_dyn<DR>(P, e);
  bool do_reset = false;  // reset_buf of the previous step (kick_env.py:433-435) and the episode counter: fetched during the
  uint32_t episode = 0u;  // last substep (their latency hides behind pass 3 without holding registers through the physics)
  const bool last_only = (P.flags & BEZ_FLAG_CF_LAST_SUBSTEP) != 0;
  WS_STAMP(side, 0);
  ws_barrier();  // B0: actions staged, root/ball and the leg joint state published
  WS_STAMP(side, 1);
  load_targets<FIRST, LEN, PRE, false>(P, lds, lane, e, target);
  for (int s = 0; s < P.substeps; ++s) {
    const bool keep = last_only ? (s == P.substeps - 1) : true;
    const bool first = last_only ? true : (s == 0);
    const ChainDyn D = in_loop(D0);

    float kps[LEN], kds[LEN], ms[LEN], lo[LEN], hi[LEN];
    {
      int ei = e;
      if (DR) asm volatile("" : "+v"(ei));  // not hoisted out of the loop (see load_joint_params)
      load_joint_params<FIRST, LEN, DR>(P, ei, kps, kds, ms, lo, hi);
    }
    RootView R = load_root_view(lds, lane);
    LinkInertia LI[LEN]; SV pAl[LEN], Sl[LEN], cbl[LEN];
    BallSel sel;
    sel.link = -1; sel.depth = 0.f; sel.n = sel.P = sel.f0p = sel.x = sel.xb = mk(0, 0, 0); sel.A = sym3zero();
    M3 Eend; V3 rend; SV Vend, Vsel = svzero();
    ws_chain_pass1<FIRST, LEN, true, CL, false, BEZ_W8_LEG_BAR>(P, D, ms, R, q, qd, LI, pAl, Sl, cbl, Eend, rend, Vend, sel, Vsel, s > 0);  // B5 of the previous substep inside
    if (!CL) {  // this leg's foot box as a ball candidate (the ball's new state was published at the barrier inside pass 1)
      BallSel fs;
      fs.link = -1; fs.depth = 0.f; fs.n = fs.P = fs.f0p = fs.x = fs.xb = mk(0, 0, 0); fs.A = sym3zero();
      test_box<link_box(FIRST + LEN - 1)>(Eend, rend, xs_load_v3(lds, lane, X_BALL) - xs_load_v3(lds, lane, X_ROOT), fs);
      publish_cand(lds, lane, X_CANDF + side * 14, fs, Vend);
    }
    Sym6 Kc = sym6zero(); SV pc = svzero();
    ws_ground_points<FIRST + LEN - 1, CL>(P, D.mu, R.root_z, Eend, rend, Vend, Kc, pc, lds, lane, X_HIT + side * 32);
    xs_store_body_contact(lds, lane, X_BCN + side * 18, body_contact_of(Kc, pc));
    WS_STAMP(side, 2 + 8 * s);
    ws_barrier();  // B1: the three ball candidates (roles 4, 5, 3) are evaluated and published
    WS_STAMP(side, 3 + 8 * s);
    // this leg's candidate wins: evaluate the contact now (skipped by the whole wave when no env of the workgroup has one)
    const CandDepths cd = load_cand_depths<CL>(lds, lane);
    const int c0 = side ? cd.br : cd.bl;
    bool mine = (cand_winner(cd) == side) && (XS(c0 + 1) >= 1.f);
    if (mine) {
      const RootView Rb = load_root_view(lds, lane);
      sel.depth = XS(c0); sel.link = (int)XS(c0 + 1);
      sel.n = xs_load_v3(lds, lane, c0 + 2); sel.P = xs_load_v3(lds, lane, c0 + 5);
      const SV Vl = xs_load_sv(lds, lane, c0 + 8);
      const BallBody ball = ball_setup(P, D.mu, D.g, Rb.ball_z, Rb.ball_ang, Rb.ball_lin);
      ball_link_contact(P, D.mu, Rb.ball_ang, Rb.ball_lin, ball, Rb.bc, Vl, sel);  // may reject the candidate (link = -1)
      mine = sel.link >= 1;
    } else {
      sel.link = -1;
    }
#if BEZ_WS_SUB == 4
    // lane-group form (bez_ws_quad.inc): the quad's lanes hold one 3 x 3 block of the articulated inertia each
    const Quad Q = quad_of(lane);
    P3q p3[LEN];
    V3 pA;
    {
      Blk M;
      wq_chain_pass2<FIRST, LEN, (!CL && !DR)>(P, Q, lds, lane, side, s + 1, kps, kds, lo, hi, q, qd, target, LI, pAl, Sl, cbl, Kc, pc, mine, sel, p3, M, pA);
#pragma unroll
      for (int i = 0; i < 9; ++i) XS(X_IA + side * 27 + i) = M.m[i];   // the block, in the quad's layout (wq_add_leg_block)
    }
    WS_STAMP(side, 24 + s);
    ws_barrier();  // B1c: both helper parts' leg<->leg contact wrenches are in LDS
    const float sc = wq_chain_self_correction<LEN>(P, Q, lds, lane, side, s + 1, p3, pA, X_IA + side * 27 + 9);
#else
    P3 p3[LEN];
    Sym6 IA = sym6zero(); SV pA = svzero();
    ws_chain_pass2<FIRST, LEN, true>(P, D, kps, kds, lo, hi, q, qd, target, LI, pAl, Sl, cbl, Kc, pc, mine, sel, p3, IA, pA);
    {  // the chain's articulated inertia goes to its X_IA block NOW (free since B1: the staged actions are consumed): 21 registers
       // less across the barrier and the correction below; the bias follows once the leg<->leg share is in it
      const float* f = (const float*)&IA;
#pragma unroll
      for (int i = 0; i < 21; ++i) XS(X_IA + side * 27 + i) = f[i];
    }
    WS_STAMP(side, 24 + s);
    ws_barrier();  // B1c: both helper parts' leg<->leg contact wrenches are in LDS
    const float sc = ws_chain_self_correction<LEN>(P, lds, lane, side, s + 1, p3, pA);   // (stores the bias part of the chain's X_IA block)
#endif
    WS_STAMP(side, 4 + 8 * s);
    ws_barrier();  // B2
    WS_STAMP(side, 5 + 8 * s);
    ws_barrier();  // B3: torso acceleration published
    WS_STAMP(side, 6 + 8 * s);
    // the leg's joint state is re-read from its X_LEGQ slots (published there for the helper roles anyway): the twelve registers are free
    // across pass 2's tail, the correction and the root solve -- the leg role then compiles without spilled VGPRs (tools/role_resources.sh)
#pragma unroll
    for (int i = 0; i < LEN; ++i) { q[i] = XS(X_LEGQ + side * 12 + i); qd[i] = XS(X_LEGQ + side * 12 + 6 + i); }
    V3 fl = mk(0, 0, 0), fend = mk(0, 0, 0);
#if BEZ_WS_SUB == 4
    SV aend = wq_chain_pass3<FIRST, LEN, CL>(P, Q, p3, q, qd, mine, sel, fl, fend, lds, lane, keep, first, sc);
#else
    SV a0 = xs_load_sv(lds, lane, X_A0);
    SV aend = ws_chain_pass3<FIRST, LEN, true, CL>(P, a0, p3, q, qd, mine, sel, fl, fend, lds, lane, keep, first, sc);
#endif
    if (mine && sel.link >= 0) { xs_store_v3(lds, lane, X_FL, fl); xs_store_v3(lds, lane, X_FL + 3, sel.xb); }
    if (keep) {
      if constexpr (CL) {  // the foot plate only feels the ball / the other leg; the ground acts on the four cleats
        ws_cf_acc(lds, lane, link_body<CL>(FIRST + LEN - 1), fend, P.cf_w, first);
        ws_cleat_forces<FIRST + LEN - 1>(P, lds, lane, X_HIT + side * 32, aend, first);
      } else {
        const BodyContact bcn = xs_load_body_contact(lds, lane, X_BCN + side * 18);
        ws_cf_acc(lds, lane, link_body<CL>(FIRST + LEN - 1), fend + cf_ground(P, body_contact_force(bcn, aend)), P.cf_w, first);
      }
    }
#pragma unroll
    for (int i = 0; i < LEN; ++i) { XS(X_LEGQ + side * 12 + i) = q[i]; XS(X_LEGQ + side * 12 + 6 + i) = qd[i]; }
    if (POST && s == P.substeps - 1) { do_reset = P.reset[e] != 0; episode = P.episode[e]; }
    WS_STAMP(side, 7 + 8 * s);
    ws_barrier();  // B4
    WS_STAMP(side, 8 + 8 * s);
  }
  // joint-side post-physics in the window of the root's last ball update (needs nothing the ball publishes)
  ws_chain_epilogue<(FIRST == 5 ? 0 : 1), POST>(P, lds, lane, e, active, do_reset, episode, q, qd, target);
  WS_STAMP(side, 21);
  ws_barrier();  // B5 of the last substep = the last barrier: contact rows, staged observation rows and pose-error sums complete
  WS_STAMP(side, 9 + 8 * (P.substeps - 1));
}

// A 2-link chain (head / one arm) owned by a role: its state and the three passes, split at the barriers by the caller.
template <int FIRST, int BLOCK_IA, bool CL>
struct Chain2 {
  float q[2], qd[2], target[2], kps[2], kds[2], ms[2], lo[2], hi[2];
  P3 p3[2];
  BodyContact bcn;
  Sym6 IAc; SV pAc;  // the chain's contribution as seen by the torso (also stored to its X_IA block)
  BEZ_DEV void up(const Params& P, const ChainDyn& D, float* lds, int lane) {  // passes 1 + 2, contribution -> X_IA block
    RootView R = load_root_view(lds, lane);
    BallSel nosel; nosel.link = -1; nosel.depth = 0.f; nosel.n = nosel.P = nosel.f0p = nosel.x = nosel.xb = mk(0, 0, 0); nosel.A = sym3zero();
    LinkInertia LI[2]; SV pAl[2], Sl[2], cbl[2]; M3 Ee; V3 re; SV Ve, Vs = svzero();
    ws_chain_pass1<FIRST, 2, false, CL>(P, D, ms, R, q, qd, LI, pAl, Sl, cbl, Ee, re, Ve, nosel, Vs);
    Sym6 Kc = sym6zero(); SV pc = svzero();
    ws_ground_points<FIRST + 1, CL>(P, D.mu, R.root_z, Ee, re, Ve, Kc, pc);
    Sym6 IA = sym6zero(); SV pA = svzero();
    ws_chain_pass2<FIRST, 2, false>(P, D, kps, kds, lo, hi, q, qd, target, LI, pAl, Sl, cbl, Kc, pc, false, nosel, p3, IA, pA);
    bcn = body_contact_of(Kc, pc);
    IAc = IA; pAc = pA;
    xs_store_sym6(lds, lane, X_IA + BLOCK_IA * 27, IA, pA);
  }
  BEZ_DEV void down(const Params& P, float* lds, int lane, bool keep, bool first) {  // pass 3 + the chain-end contact row
    BallSel nosel; nosel.link = -1; nosel.depth = 0.f; nosel.n = nosel.P = nosel.f0p = nosel.x = nosel.xb = mk(0, 0, 0); nosel.A = sym3zero();
    SV a0 = xs_load_sv(lds, lane, X_A0);
    V3 fl = mk(0, 0, 0), fend = mk(0, 0, 0);
    SV ae = ws_chain_pass3<FIRST, 2, false, CL>(P, a0, p3, q, qd, false, nosel, fl, fend, lds, lane, keep, first);
    if (keep) ws_cf_acc(lds, lane, link_body<CL>(FIRST + 1), cf_ground(P, body_contact_force(bcn, ae)), P.cf_w, first);
  }
};

template <bool PRE, bool POST, bool DR, bool CL>
BEZ_DEV void head_role(const Params& P, float* lds, int lane, int e, bool active) {
  Chain2<1, 2, CL> C;
  load_joints<1, 2, DR>(P, e, C.q, C.qd, C.kps, C.kds, C.ms, C.lo, C.hi);
  const ChainDyn D = load_chain_dyn<DR>(P, e);
  const bool do_reset = POST && P.reset[e] != 0;
  const uint32_t episode = POST ? P.episode[e] : 0u;
  const bool last_only = (P.flags & BEZ_FLAG_CF_LAST_SUBSTEP) != 0;
  ws_barrier();  // B0
  load_targets<1, 2, PRE, true>(P, lds, lane, e, C.target);
  for (int s = 0; s < P.substeps; ++s) {
    const bool keep = last_only ? (s == P.substeps - 1) : true;
    const bool first = last_only ? true : (s == 0);
    {  // ball candidates among the hip / thigh boxes of both legs: kinematics first, the tests once the ball's new state is published
      const RootOnly R = load_root_only(lds, lane);
      constexpr int HI = cand_split<CL>();
      LegFrames<1, HI> FL, FR;
      leg_frames<5, 1, HI>(lds, lane, 0, R.E0, R.V0, FL);
      leg_frames<13, 1, HI>(lds, lane, 1, R.E0, R.V0, FR, quirk_rz<CL>(P.flags));
      if (s > 0) ws_barrier();  // B5 of the previous substep
      const V3 bc = xs_load_v3(lds, lane, X_BALL) - R.pos;
      leg_box_tests<5, 1, HI>(FL, bc, lds, lane, X_CANDH);
      leg_box_tests<13, 1, HI>(FR, bc, lds, lane, X_CANDH + 14);
    }
    WS_STAMP(2, 2 + 8 * s);
    ws_barrier();  // B1  (X_IA is free from here on: the staged actions have been consumed)
    C.up(P, in_loop(D), lds, lane);
    WS_STAMP(2, 4 + 8 * s);
    ws_barrier();  // B1c: head and arm blocks are in LDS (role 4 sums them into block 2 before B2)
    ws_barrier();  // B2
    ws_barrier();  // B3
    C.down(P, lds, lane, keep, first);
    ws_barrier();  // B4
  }
  ws_chain_epilogue<2, POST>(P, lds, lane, e, active, do_reset, episode, C.q, C.qd, C.target);
  WS_STAMP(2, 21);
  ws_barrier();  // B5 of the last substep
}

// roles 4 / 5: the deepest ball<->leg-box candidate of one leg from this wave's own forward kinematics of that leg (window of
// the leg's pass 1), then the arm of the same side as a chain of its own (window of the leg's pass 2); role 4 finally adds the head and
// right-arm blocks to its own and leaves the sum in block 2, so that the root role reads three blocks.
template <int LEG_FIRST, int ARM_FIRST, bool PRE, bool POST, bool DR, bool CL>
BEZ_DEV void cand_arm_role(const Params& P, float* lds, int lane, int e, bool active, int side) {
  constexpr int ROLE = LEG_FIRST == 5 ? 4 : 5;
  Chain2<ARM_FIRST, 3 + (LEG_FIRST == 5 ? 0 : 1), CL> C;
  load_joints<ARM_FIRST, 2, DR>(P, e, C.q, C.qd, C.kps, C.kds, C.ms, C.lo, C.hi);
  const ChainDyn D = load_chain_dyn<DR>(P, e);
  const bool do_reset = POST && P.reset[e] != 0;
  const uint32_t episode = POST ? P.episode[e] : 0u;
  const bool last_only = (P.flags & BEZ_FLAG_CF_LAST_SUBSTEP) != 0;
  ws_barrier();  // B0
  load_targets<ARM_FIRST, 2, PRE, false>(P, lds, lane, e, C.target);
  for (int s = 0; s < P.substeps; ++s) {
    const bool keep = last_only ? (s == P.substeps - 1) : true;
    const bool first = last_only ? true : (s == 0);
    {  // ball candidate among the calf / ankle / foot boxes: kinematics first, the tests once the ball's new state is published
      const RootOnly R = load_root_only(lds, lane);
      constexpr int LO0 = cand_split<CL>(), LO1 = cand_lo_end<CL>();
      LegFrames<LO0, LO1> F;
      leg_frames<LEG_FIRST, LO0, LO1>(lds, lane, side, R.E0, R.V0, F, quirk_rz<CL>(P.flags));
      if (s > 0) ws_barrier();  // B5 of the previous substep
      leg_box_tests<LEG_FIRST, LO0, LO1>(F, xs_load_v3(lds, lane, X_BALL) - R.pos, lds, lane, X_CAND + side * 14);
    }
    WS_STAMP(ROLE, 2 + 8 * s);
    ws_barrier();  // B1
#if BEZ_WS_SUB == 4
    if constexpr (!CL && !DR) wq_produce_packages<LEG_FIRST>(P, in_loop(D).g, lds, lane, side, s + 1);   // the hip links' packages of this side's leg
#endif
    C.up(P, in_loop(D), lds, lane);
    WS_STAMP(ROLE, 4 + 8 * s);
    ws_barrier();  // B1c: head and arm blocks are in LDS
    if (ROLE == 4) {
#if BEZ_WS_SUB == 4
      // (lane-group form: the root role reads the three inertia parts itself, behind B1c; only the biases are summed here)
      const SV p2 = (xs_load_sv(lds, lane, X_IA + 2 * 27 + 21) + xs_load_sv(lds, lane, X_IA + 4 * 27 + 21)) + C.pAc;
      xs_store_sv(lds, lane, X_IA + 2 * 27 + 21, p2);
#else
      Sym6 I2; SV p2;
      xs_load_sym6(lds, lane, X_IA + 2 * 27, I2, p2);
      xs_add_sym6(lds, lane, X_IA + 4 * 27, I2, p2);
      add_to(I2, C.IAc); p2 = p2 + C.pAc;
      xs_store_sym6(lds, lane, X_IA + 2 * 27, I2, p2);
#endif
    }
    ws_barrier();  // B2
    ws_barrier();  // B3
    C.down(P, lds, lane, keep, first);
    ws_barrier();  // B4
  }
  ws_chain_epilogue<ROLE, POST>(P, lds, lane, e, active, do_reset, episode, C.q, C.qd, C.target);
  WS_STAMP(ROLE, 21);
  ws_barrier();  // B5 of the last substep
}

// ---- post-physics shares of the two pair roles (they are idle after B1c): the observation slots that do not depend on
// the joints.  Both see the state the root role published before the last B5 and apply the pending reset themselves
// (kick_env.py:433-435 resets BEFORE compute_observations).
struct PostIn { bool reset; float prev[3], goal_x, goal_y; };  // fetched at kernel entry: the latency hides behind the physics
template <bool POST>
BEZ_DEV PostIn load_post_in(const Params& P, int e, bool want_imu) {
  PostIn in; in.reset = false; in.prev[0] = in.prev[1] = in.prev[2] = 0.f; in.goal_x = P.goal[0]; in.goal_y = P.goal[1];
  if (POST) {
    const int n = P.n;
    in.reset = P.reset[e] != 0;
    if (want_imu) {
#pragma unroll
      for (int i = 0; i < 3; ++i) in.prev[i] = P.state[(size_t)(F_PREV + i) * n + e];
      if (P.task != BEZ_TASK_KICK) { in.goal_x = P.state[(size_t)F_GOAL * n + e]; in.goal_y = P.state[(size_t)(F_GOAL + 1) * n + e]; }
    }
  }
  return in;
}
BEZ_DEV void post_imu_orn(const Params& P, float* lds, int lane, int e, bool active, const PostIn& in) {
  const int n = P.n;
  float* st = P.state;
  const bool reset = in.reset;
  V3 root_pos = xs_load_v3(lds, lane, X_ROOT);
  float rq[4] = {XS(X_ROOT + 3), XS(X_ROOT + 4), XS(X_ROOT + 5), XS(X_ROOT + 6)};
  V3 lin = xs_load_v3(lds, lane, X_ROOT + 7), ang = xs_load_v3(lds, lane, X_ROOT + 10);
  float prev[3] = {in.prev[0], in.prev[1], in.prev[2]};
  float goal_x = in.goal_x, goal_y = in.goal_y;
  if (reset) {
    root_pos = mk(P.bez_init[0], P.bez_init[1], P.bez_init[2]);
#pragma unroll
    for (int i = 0; i < 4; ++i) rq[i] = P.bez_init[3 + i];
    lin = ang = mk(0, 0, 0);
    if (P.task != BEZ_TASK_KICK) { goal_x = reset_goal(P, 0); goal_y = reset_goal(P, 1); }
  }
  float tail[8];
  obs_imu_orn(P, root_pos, rq, lin, ang, prev, goal_x, goal_y, tail);
  float* obs_row = lds + X_STAGE * WS_LANES + ws_env_of(lane) * P.nobs;
#pragma unroll
  for (int i = 0; i < 8; ++i) obs_row[36 + i] = tail[i];
  if (active && !P.lean) {
#pragma unroll
    for (int i = 0; i < 3; ++i) st[(size_t)(F_PREV + i) * n + e] = prev[i];
  }
}
template <bool CL>
BEZ_DEV void post_feet(const Params& P, float* lds, int lane, int e, bool active, const PostIn& in) {
  const int n = P.n;
  float* st = P.state;
  // a reset env reports no contact forces (kick_env.py:433-438: reset before the observation): its rows are zeroed on their way
  // out (X_RESETF, the copy-out at the end of the kernel), and the feet logic below sees zeros
  XS(X_RESETF) = in.reset ? 1.f : 0.f;
  CfOut co;
  co.base = nullptr; co.n = n;
  co.lf = xs_load_v3(lds, lane, X_CF + lfoot_body<CL>() * 3); co.rf = xs_load_v3(lds, lane, X_CF + rfoot_body<CL>() * 3);
  if (in.reset) co.lf = co.rf = mk(0, 0, 0);
  float cleats[24];
  if (CL) {
#pragma unroll
    for (int k = 0; k < 12; ++k) { cleats[k] = XS(X_CF + BEZ_LCLEAT_BODY_CL * 3 + k); cleats[12 + k] = XS(X_CF + BEZ_RCLEAT_BODY_CL * 3 + k); }
    if (in.reset) {
#pragma unroll
      for (int k = 0; k < 24; ++k) cleats[k] = 0.f;
    }
  }
  float feet[8], tail[18];
  obs_feet(P, co, CL ? cleats : nullptr, feet, tail);
  float* obs_row = lds + X_STAGE * WS_LANES + ws_env_of(lane) * P.nobs;
#pragma unroll
  for (int i = 8; i < 18; ++i) if (36 + i < P.nobs) obs_row[36 + i] = tail[i];
  if (!CL) {  // the no-cleats feet logic filters the two foot rows in place (kick_env.py:987-990)
    xs_store_v3(lds, lane, X_CF + BEZ_LFOOT_BODY * 3, co.lf); xs_store_v3(lds, lane, X_CF + BEZ_RFOOT_BODY * 3, co.rf);
  }
  if (active && !P.lean) {
#pragma unroll
    for (int i = 0; i < 8; ++i) st[(size_t)(F_FEET + i) * n + e] = feet[i];
  }
}

// roles 6 / 7: leg<->leg capsule pairs.  Forward kinematics of both legs from X_LEGQ, then this part's pairs; the wrenches are
// complete at B1c (the legs apply them after pass 2).  After the last substep: role 6 the imu / orientation observation slots,
// role 7 the feet slots.
template <int PART, bool POST, bool DR, bool CL>
BEZ_DEV void self_role(const Params& P, float* lds, int lane, int e, bool active) {
  const ChainDyn D = load_chain_dyn<DR>(P, e);
  const PostIn pin_ = load_post_in<POST>(P, e, PART == 0);
  ws_barrier();  // B0
  for (int s = 0; s < P.substeps; ++s) {
    SelfCaps K;
    {  // kinematics of both legs: one third in, meet the other roles at B5 of the previous substep (this role does not read the ball)
      const M3 E0 = quat_to_mat(XS(X_ROOT + 3), XS(X_ROOT + 4), XS(X_ROOT + 5), XS(X_ROOT + 6));
      const SV V0 = mksv(xs_load_v3(lds, lane, X_ROOT + 10), xs_load_v3(lds, lane, X_ROOT + 7));
#if BEZ_WS_SUB == 4
      ws_self_fk<PART, BEZ_W8_SELF_BAR, ((!CL && !DR) ? 1 - PART : -1)>(lds, lane, E0, V0, K, s > 0, quirk_rz<CL>(P.flags), in_loop(D).g);   // + the other leg's ankle / foot packages
#else
      ws_self_fk<PART, BEZ_W8_SELF_BAR>(lds, lane, E0, V0, K, s > 0, quirk_rz<CL>(P.flags));
#endif
      ws_self_pin<PART>(K);
    }
    WS_STAMP(6 + PART, 2 + 8 * s);
    ws_barrier();  // B1
    ws_self_pairs<PART>(P, D.mu, lds, lane, K);
    WS_STAMP(6 + PART, 4 + 8 * s);
    ws_barrier();  // B1c
    ws_barrier();  // B2
    ws_barrier();  // B3
    ws_barrier();  // B4
  }
  if (POST) {  // after B4 of the last substep: the root state (published before B4) and the robot's contact rows are final
    if (PART == 0) post_imu_orn(P, lds, lane, e, active, pin_);
    else post_feet<CL>(P, lds, lane, e, active, pin_);
  }
  WS_STAMP(6 + PART, 21);
  ws_barrier();  // B5 of the last substep
}

template <bool PRE, bool POST, bool DR, bool CL>
BEZ_DEV void root_role(const Params& P, float* lds, int lane, int e, bool active) {
  const int n = P.n;
  float* st = P.state;
  auto ld = [&](int f) { return st[(size_t)f * n + e]; };
  V3 root_pos = mk(ld(F_ROOT_POS), ld(F_ROOT_POS + 1), ld(F_ROOT_POS + 2));
  float rq[4] = {ld(F_ROOT_QUAT), ld(F_ROOT_QUAT + 1), ld(F_ROOT_QUAT + 2), ld(F_ROOT_QUAT + 3)};
  V3 root_lin = mk(ld(F_ROOT_LIN), ld(F_ROOT_LIN + 1), ld(F_ROOT_LIN + 2));
  V3 root_ang = mk(ld(F_ROOT_ANG), ld(F_ROOT_ANG + 1), ld(F_ROOT_ANG + 2));
  V3 ball_pos = mk(ld(F_BALL_POS), ld(F_BALL_POS + 1), ld(F_BALL_POS + 2));
  float bq[4] = {ld(F_BALL_QUAT), ld(F_BALL_QUAT + 1), ld(F_BALL_QUAT + 2), ld(F_BALL_QUAT + 3)};
  V3 ball_lin = mk(ld(F_BALL_LIN), ld(F_BALL_LIN + 1), ld(F_BALL_LIN + 2));
  V3 ball_ang = mk(ld(F_BALL_ANG), ld(F_BALL_ANG + 1), ld(F_BALL_ANG + 2));
  // bookkeeping inputs of the post-physics, fetched now so that their latency hides behind the physics
  int64_t progress = 0, reset = 0;
#if BEZ_WS_SUB == 4
  // (lane-group form: fetched during the last substep instead -- four registers less across the physics, the kernel's last spilled value)
#else
  if (POST) { progress = P.progress[e]; reset = P.reset[e]; }
#endif
  const ChainDyn D = load_chain_dyn<DR>(P, e);
  float ms0 = 1.f;
  if (DR) { if (P.dr_mass) ms0 = P.dr_mass[(size_t)e * BEZ_NL]; }
  const bool last_only = (P.flags & BEZ_FLAG_CF_LAST_SUBSTEP) != 0;
  auto publish_root = [&]() __attribute__((always_inline)) {
    xs_store_v3(lds, lane, X_ROOT, root_pos);
    XS(X_ROOT + 3) = rq[0]; XS(X_ROOT + 4) = rq[1]; XS(X_ROOT + 5) = rq[2]; XS(X_ROOT + 6) = rq[3];
    xs_store_v3(lds, lane, X_ROOT + 7, root_lin); xs_store_v3(lds, lane, X_ROOT + 10, root_ang);
  };
  auto publish_ball = [&]() __attribute__((always_inline)) {
    xs_store_v3(lds, lane, X_BALL, ball_pos); xs_store_v3(lds, lane, X_BALL + 3, ball_lin); xs_store_v3(lds, lane, X_BALL + 6, ball_ang);
  };
  publish_root(); publish_ball();
  WS_STAMP_ROOT_N(0, 3, 0);
  ws_barrier();  // B0
  WS_STAMP_ROOT_N(1, 3, 1);
  // The ball is integrated one barrier late: its update needs the ball<->link force of pass 3 (published at B4), but only the
  // box tests of the NEXT substep need its result -- so the update runs at the top of the next iteration (and once after the
  // loop), beside the other roles' forward kinematics, and B5 sits in the middle of their pass-1 window.
  BallBody ball; BallSel sel; bool torso_hit = false; V3 fl_t = mk(0, 0, 0);
  auto ball_update = [&](bool keep, bool first) __attribute__((always_inline)) {
    V3 fl = xs_load_v3(lds, lane, X_FL), xb = xs_load_v3(lds, lane, X_FL + 3);
    pin(fl); pin(xb);  // loaded values, not a select between an LDS and a private address (that would put fl_t / sel in scratch)
    if (torso_hit) { fl = fl_t; xb = sel.xb; }
    SV ab = ball_minv(ball, svzero() - ball.pb - wrench_at(xb, fl));
    if (keep) {
      V3 fb = -cf_along(P, fl, sel.n);
      if (ball.ground) fb = fb + cf_ground(P, hit_force(P, ball.ghit, ab));
      ws_cf_acc(lds, lane, nb_of<CL>(), fb, P.cf_w, first);
    }
    float damp = fmaxf(1.0f - P.h * P.ball_damp, 0.f);
    ball_lin = fma3(ab.l, P.h, ball_lin);
    ball_ang = fma3(ab.a, P.h, ball_ang) * damp;
    ball_pos = fma3(ball_lin, P.h, ball_pos);
    quat_integrate(bq, ball_ang, P.h);
    publish_ball();
  };
  for (int s = 0; s < P.substeps; ++s) {
    const bool keep = last_only ? (s == P.substeps - 1) : true;
    const bool first = last_only ? true : (s == 0);
    if (s > 0) {
      ball_update(last_only ? false : true, last_only ? true : (s == 1));  // the previous substep's flags
      ws_barrier();  // B5 of the previous substep
      WS_STAMP_ROOT_N(2, 3, 9 + 8 * (s - 1));
    }
    const M3 E0 = quat_to_mat(rq[0], rq[1], rq[2], rq[3]);
    const SV V0 = mksv(root_ang, root_lin);
    const V3 bc = ball_pos - root_pos;
    xs_store_v3(lds, lane, X_FL, mk(0, 0, 0)); xs_store_v3(lds, lane, X_FL + 3, mk(0, 0, 0));
    ball = ball_setup(P, D.mu, D.g, ball_pos.z, ball_ang, ball_lin);
    // the torso box as a ball candidate, evaluated now; the winner is decided from the published depths after B1
    sel.link = -1; sel.depth = 0.f; sel.n = sel.P = sel.f0p = sel.x = sel.xb = mk(0, 0, 0); sel.A = sym3zero();
    test_torso_box(P, E0, mk(0, 0, 0), bc, sel);
    XS(X_TORSO) = sel.depth;
    if (sel.link == 0) ball_link_contact(P, D.mu, ball_ang, ball_lin, ball, bc, V0, sel);
    WS_STAMP_ROOT_N(3, 3, 2 + 8 * s);
    ws_barrier();  // B1: all candidates are in LDS
    WS_STAMP_ROOT_N(4, 3, 3 + 8 * s);
    const CandDepths cd = load_cand_depths<CL>(lds, lane);
    const int winner = cand_winner(cd);
    torso_hit = (winner == 2) && (sel.link == 0);
    {  // contact normal of the winning leg box (for the ball's contact row)
      V3 nw = xs_load_v3(lds, lane, (winner == 1 ? cd.br : cd.bl) + 2);
      pin(nw);
      if (winner != 2) sel.n = nw;
    }
    // the torso's own inertia and guard points: nobody needs them before B2, so they run in this role's idle window beside the
    // legs' pass 2 instead of ahead of B1
    Sym6 IA0 = sym6zero(); SV pA0;
    LinkInertia I0;
    link_inertia<0, CL>(ms0, D.g, E0, mk(0, 0, 0), V0, I0, pA0);
    Sym6 Kc = sym6zero(); SV pc = svzero();
    ws_ground_points<0, CL>(P, D.mu, root_pos.z, E0, mk(0, 0, 0), V0, Kc, pc);
#if BEZ_WS_SUB == 4
    {  // the torso's contact rows are only needed behind B3: parked in the spare slots 72..76 (word k in column k % 4 of the quad) so that they
       // do not hold 18 registers across the factorisation below
      const BodyContact b = body_contact_of(Kc, pc);
      const float* f = (const float*)&b;
      float* col = lds + (lane & ~3);
#pragma unroll
      for (int k = 0; k < 18; ++k) col[(X_SPARE + k / 4) * WS_LANES + (k & 3)] = f[k];
    }
#else
    BodyContact bc0 = body_contact_of(Kc, pc);
#endif
    add_link_inertia(IA0, I0);
    add_to(IA0, Kc); pA0 = pA0 + pc;
    if (torso_hit) { add_point_stiffness(IA0, sel.x, sel.A); pA0 = pA0 - wrench_at(sel.x, sel.f0p); }
    WS_STAMP_ROOT_N(5, 3, 24 + s);
    ws_barrier();  // B1c
#if BEZ_WS_SUB == 4
    // every chain's articulated inertia is in LDS (the legs store theirs right behind pass 2; the leg<->leg correction that runs now
    // changes the biases only): sum and FACTORISE while the legs are busy, so that only the two substitutions sit between B2 and B3
    wq_add_leg_inertia(lds, lane, X_IA, IA0); wq_add_leg_inertia(lds, lane, X_IA + 27, IA0);   // gathered from the quad's columns
#pragma unroll
    for (int k = 2; k < 5; ++k) xs_add_inertia(lds, lane, X_IA + k * 27, IA0);                 // head, left arm, right arm (role 4 sums their biases only)
    const Ldl6 F = ldl6_factor(IA0);
    ws_barrier();  // B2: the biases are published
    WS_STAMP_ROOT_N(6, 3, 5 + 8 * s);
    wq_add_leg_bias(lds, lane, X_IA, pA0); wq_add_leg_bias(lds, lane, X_IA + 27, pA0);
    pA0 = pA0 + xs_load_sv(lds, lane, X_IA + 2 * 27 + 21);
    SV a0 = ldl6_solve(F, svzero() - pA0);
#else
    ws_barrier();  // B2: chain contributions published
    WS_STAMP_ROOT_N(7, 3, 5 + 8 * s);
#pragma unroll
    for (int k = 0; k < 3; ++k) xs_add_sym6(lds, lane, X_IA + k * 27, IA0, pA0);  // legs + (head + arms, summed by role 4)
    SV a0 = solve_spd6(IA0, svzero() - pA0);
#endif
    xs_store_sv(lds, lane, X_A0, a0);
    WS_STAMP_ROOT_N(8, 3, 4 + 8 * s);
    ws_barrier();  // B3
    WS_STAMP_ROOT_N(9, 3, 6 + 8 * s);
    fl_t = mk(0, 0, 0);
#if BEZ_WS_SUB == 4
    BodyContact bc0;
    {
      float* f = (float*)&bc0;
      const float* col = lds + (lane & ~3);
#pragma unroll
      for (int k = 0; k < 18; ++k) f[k] = col[(X_SPARE + k / 4) * WS_LANES + (k & 3)];
    }
#endif
    if (torso_hit) fl_t = sel.f0p - mul(sel.A, point_of(a0, sel.x));
    if (keep) ws_cf_acc(lds, lane, 0, cf_along(P, fl_t, sel.n) + cf_ground(P, body_contact_force(bc0, a0)), P.cf_w, first);
    V3 vdot = a0.l + cross(root_ang, root_lin);
    root_ang = fma3(a0.a, P.h, root_ang);
    root_lin = fma3(vdot, P.h, root_lin);
    root_pos = fma3(root_lin, P.h, root_pos);
    quat_integrate(rq, root_ang, P.h);
    publish_root();  // the chains' next pass 1 starts right after B4
#if BEZ_WS_SUB == 4
    if (POST && s == P.substeps - 1) { progress = P.progress[e]; reset = P.reset[e]; }
#endif
    WS_STAMP_ROOT_N(10, 3, 7 + 8 * s);
    ws_barrier();  // B4: ball<->link force published
    WS_STAMP_ROOT_N(11, 3, 8 + 8 * s);
  }
  {
    const bool keep_l = true, first_l = last_only ? true : (P.substeps == 1);
    ball_update(keep_l, first_l);
  }
  // post-physics, root's share.  Before the last B5 (the chain roles are busy with their joints' post-physics): bookkeeping, the
  // pending reset of the root / ball state, the state stores.  After it (pose-error sums of the chain roles are in LDS): the
  // reward and the reset flag of the next step, beside the other waves' copy-out.
  float goal_x = P.goal[0], goal_y = P.goal[1];
  int64_t timeout = 0;
  bool new_episode = false;
  if (POST) {
    if (P.task != BEZ_TASK_KICK) { goal_x = ld(F_GOAL); goal_y = ld(F_GOAL + 1); }
    timeout = (progress >= (int64_t)(P.max_len - 1)) ? 1 : 0;  // vec_task.py:331-332
    progress += 1;                                              // kick_env.py:429
    if (reset != 0) {                                           // kick_env.py:433-435, 831-850 (root / ball part)
      root_pos = mk(P.bez_init[0], P.bez_init[1], P.bez_init[2]);
      ball_pos = mk(P.ball_init[0], P.ball_init[1], P.ball_init[2]);
#pragma unroll
      for (int i = 0; i < 4; ++i) { rq[i] = P.bez_init[3 + i]; bq[i] = P.ball_init[3 + i]; }
      root_lin = root_ang = ball_lin = ball_ang = mk(0, 0, 0);
      new_episode = true;  // the counter itself is bumped behind the last barrier (below)
      if (P.task != BEZ_TASK_KICK) {  // walk_env.py:570-575: every env reset by this call receives the same fresh goal
        goal_x = reset_goal(P, 0); goal_y = reset_goal(P, 1);
        if (active) { st[(size_t)F_GOAL * n + e] = goal_x; st[(size_t)(F_GOAL + 1) * n + e] = goal_y; }
      }
      progress = 0; reset = 0;
    }
  }
  if (active) {
#if BEZ_WS_SUB == 4
    int es = e;
    asm volatile("" : "+v"(es));   // the store addresses are formed HERE: hoisted to the kernel's head (where the same fields are loaded) they were 22 spilled VGPRs
    auto sv = [&](int f, float v) { st[(size_t)f * n + es] = v; };
#else
    auto sv = [&](int f, float v) { st[(size_t)f * n + e] = v; };
#endif
    sv(F_ROOT_POS, root_pos.x); sv(F_ROOT_POS + 1, root_pos.y); sv(F_ROOT_POS + 2, root_pos.z);
#pragma unroll
    for (int i = 0; i < 4; ++i) { sv(F_ROOT_QUAT + i, rq[i]); sv(F_BALL_QUAT + i, bq[i]); }
    sv(F_ROOT_LIN, root_lin.x); sv(F_ROOT_LIN + 1, root_lin.y); sv(F_ROOT_LIN + 2, root_lin.z);
    sv(F_ROOT_ANG, root_ang.x); sv(F_ROOT_ANG + 1, root_ang.y); sv(F_ROOT_ANG + 2, root_ang.z);
    sv(F_BALL_POS, ball_pos.x); sv(F_BALL_POS + 1, ball_pos.y); sv(F_BALL_POS + 2, ball_pos.z);
    sv(F_BALL_LIN, ball_lin.x); sv(F_BALL_LIN + 1, ball_lin.y); sv(F_BALL_LIN + 2, ball_lin.z);
    sv(F_BALL_ANG, ball_ang.x); sv(F_BALL_ANG + 1, ball_ang.y); sv(F_BALL_ANG + 2, ball_ang.z);
  }
  WS_STAMP_ROOT_N(12, 3, 19);
  ws_barrier();  // B5 of the last substep = the last barrier
  WS_STAMP_ROOT_N(13, 3, 9 + 8 * (P.substeps - 1));
  // The chain roles key their reset draw with P.episode[e] / P.reset[e], loaded before B4 and CONSUMED (hence waited for) before
  // this barrier; ws_barrier() itself does not wait for outstanding global loads (vmcnt), so the two words are only rewritten here,
  // behind the last barrier -- no reliance on the memory pipeline serving another wave's earlier load before this store.
  if (POST && active && new_episode) P.episode[e] = P.episode[e] + 1;
  if (POST) {
    const float pn = (((XS(X_PSUM + 2) + XS(X_PSUM + 4)) + XS(X_PSUM + 5)) + XS(X_PSUM + 0)) + XS(X_PSUM + 1);
    OrnOut orn; orn.ux = orn.uy = orn.gn = orn.ang_goal = 0.f;
    if (P.task != BEZ_TASK_KICK) {  // the walk / orient rewards use the goal direction / heading error (role 6 computes the same for the observation)
      float t6, t7;
      orn = obs_off_orn(P, root_pos, rq, goal_x, goal_y, t6, t7);
    }
    float rew;
    reward_of(P, root_pos, rq, root_lin, root_ang, ball_pos, ball_lin, pn, orn, rew, reset, progress, goal_x, goal_y);
    if (active) { P.rew[e] = rew; P.reset[e] = reset; P.progress[e] = progress; P.timeout[e] = timeout; }
  }
  WS_STAMP_ROOT_N(14, 3, 21);
}

// ---- the kernel.  grid = ceil(N / 64) workgroups of 512 threads.
template <bool PRE, bool POST, bool DR, bool CL>
__global__ __launch_bounds__(WS_BLOCK) void step_kernel_ws8(Params P) {
  __shared__ __attribute__((aligned(16))) float lds[WS_LDS_FLOATS];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int role = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int env0 = blockIdx.x * WS_ENVS;
  const int nloc = min(WS_ENVS, P.n - env0);
  const bool valid = ws_env_of(lane) < nloc;
  const int e = env0 + (valid ? ws_env_of(lane) : 0);  // inactive lanes shadow env0 (loads only; every global store is guarded)
  const bool active = valid && ((lane & (WS_SUB - 1)) == 0);   // the lane that stores its env's results (a quad's lanes hold the same ones)
  WS_STAMP(role, 22);
  if (PRE) {
    // coalesced stage of this workgroup's contiguous (nloc,18) action block, transposed to [lane][19]
    float* act = lds + X_STAGE * WS_LANES;
    const float* src = P.actions + (size_t)env0 * BEZ_ND;
    for (int i = tid; i < nloc * BEZ_ND; i += WS_BLOCK) act[(i / BEZ_ND) * WS_ACT_STRIDE + (i % BEZ_ND)] = src[i];
  }
  if (tid < (WS_SUB == 4 ? 4 : 2)) reinterpret_cast<int*>(lds + X_PAIRSEQ * WS_LANES)[tid] = 0;   // (published before B0; words 2 / 3: the package hand-over of the lane-group form)
  // contact-force rows start from zero: bodies nothing touches are never accumulated into
  constexpr int NROW = (nb_of<CL>() + 1) * 3;  // contact-force rows of this asset (robot bodies + ball)
  for (int i = tid; i < NROW * WS_LANES; i += WS_BLOCK) lds[X_CF * WS_LANES + i] = 0.f;
  WS_STAMP(role, 18);
#ifdef BEZ_AB_ONLY_ROLE   // offline diagnostics (tools/role_resources.sh): the register / spill figures of ONE role's code; never launched
  if (BEZ_AB_ONLY_ROLE == 0) leg_role<5, PRE, POST, DR, CL>(P, lds, lane, e, active, 0);
  if (BEZ_AB_ONLY_ROLE == 1) leg_role<13, PRE, POST, DR, CL>(P, lds, lane, e, active, 1);
  if (BEZ_AB_ONLY_ROLE == 2) head_role<PRE, POST, DR, CL>(P, lds, lane, e, active);
  if (BEZ_AB_ONLY_ROLE == 3) root_role<PRE, POST, DR, CL>(P, lds, lane, e, active);
  if (BEZ_AB_ONLY_ROLE == 4) cand_arm_role<5, 3, PRE, POST, DR, CL>(P, lds, lane, e, active, 0);
  if (BEZ_AB_ONLY_ROLE == 5) cand_arm_role<13, 11, PRE, POST, DR, CL>(P, lds, lane, e, active, 1);
  if (BEZ_AB_ONLY_ROLE == 6) self_role<0, POST, DR, CL>(P, lds, lane, e, active);
  if (BEZ_AB_ONLY_ROLE == 7) self_role<1, POST, DR, CL>(P, lds, lane, e, active);
#else
  if (role == 0) leg_role<5, PRE, POST, DR, CL>(P, lds, lane, e, active, 0);
  else if (role == 1) leg_role<13, PRE, POST, DR, CL>(P, lds, lane, e, active, 1);
  else if (role == 2) head_role<PRE, POST, DR, CL>(P, lds, lane, e, active);
  else if (role == 3) root_role<PRE, POST, DR, CL>(P, lds, lane, e, active);
  else if (role == 4) cand_arm_role<5, 3, PRE, POST, DR, CL>(P, lds, lane, e, active, 0);
  else if (role == 5) cand_arm_role<13, 11, PRE, POST, DR, CL>(P, lds, lane, e, active, 1);
  else if (role == 6) self_role<0, POST, DR, CL>(P, lds, lane, e, active);
  else self_role<1, POST, DR, CL>(P, lds, lane, e, active);
#endif
  // The last substep's B5 was the last barrier: the contact-force rows and (with POST) the staged observation rows are complete in
  // LDS.  The seven other waves copy them out while the root role is still busy with the reward.
  if (role != 3) {
    constexpr int NT = WS_BLOCK - 64;
#if BEZ_WS_SUB == 4
    // the thread index is formed again here (role is a scalar; the lane from mbcnt): carried from the kernel's head it is live across the
    // root role's 256 registers and was the kernel's one spilled value -- with it gone the kernel needs no scratch memory at all
    const int ctid = (role - (role > 3 ? 1 : 0)) * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
#else
    const int ctid = tid - (role > 3 ? 64 : 0);
#endif
    if (!P.lean) {
      // net contact force: SoA rows of 64 consecutive envs each -> coalesced
      float* dst = P.state + (size_t)F_CF * P.n + env0;
      static_assert(NT % WS_ENVS == 0, "a copy-out thread stays on one env lane");
      const bool zero = POST && lds[X_RESETF * WS_LANES + ((ctid & (WS_ENVS - 1)) << WS_SUB_SHIFT)] != 0.f;  // env reset by this step: no contact forces
      for (int i = ctid; i < NROW * WS_ENVS; i += NT) {
        const int k = i >> WS_ENV_SHIFT, l = i & (WS_ENVS - 1);
        if (l < nloc) dst[(size_t)k * P.n + l] = zero ? 0.f : lds[(X_CF + k) * WS_LANES + (l << WS_SUB_SHIFT)];
      }
    }
    if (DR && POST && P.dr_snap && blockIdx.x == 0 && ctid == 0) {   // the next step's action-noise parameters: a copy (bez_kernels.h DrSnap)
      const unsigned long long f = P.dr_state->frame;
      *P.dr_snap = DrSnap{P.dr_state->noise[2], P.dr_state->noise[3], (unsigned int)f, (unsigned int)(f >> 32)};
    }
    if (POST) {
      // the staged rows are the contiguous (nloc,nobs) image of this workgroup's slice of obs_buf: 16-byte copy-out
      const float4* rows = reinterpret_cast<const float4*>(lds + X_STAGE * WS_LANES);
      float4* dst = reinterpret_cast<float4*>(P.obs + (size_t)env0 * P.nobs);  // 64 * nobs * 4 B per workgroup: 16-B aligned for 54 and 52
      const int nvec = (nloc * P.nobs) >> 2;
      if (DR && P.obs_noise) {
        // the observation noise of the domain randomisation rides on the copy-out (one Philox block per 16-byte vector, spread over
        // the seven copying waves) instead of a launch of its own behind the step: the same bits as bez_sim_add_dr_noise would add
        const long long q0 = ((long long)env0 * P.nobs) >> 2;   // env0 is a multiple of 16: the workgroup's block starts on a vector
        const float mean = P.dr_state->noise[0], sd = P.dr_state->noise[1];
        const unsigned long long frame = P.dr_state->frame;
        for (int i = ctid; i < nvec; i += NT) {
          float z[4];
          dr_noise_quad(P.seed, P.env_off, frame, 0, q0 + i, z);
          float4 v = rows[i];
          v.x += fmaf(z[0], sd, mean); v.y += fmaf(z[1], sd, mean); v.z += fmaf(z[2], sd, mean); v.w += fmaf(z[3], sd, mean);
          dst[i] = v;
        }
        for (int i = (nvec << 2) + ctid; i < nloc * P.nobs; i += NT)
          P.obs[(size_t)env0 * P.nobs + i] = obs_with_noise(P, (long long)env0 * P.nobs + i, lds[X_STAGE * WS_LANES + i]);
      } else {
        for (int i = ctid; i < nvec; i += NT) dst[i] = rows[i];
        for (int i = (nvec << 2) + ctid; i < nloc * P.nobs; i += NT) P.obs[(size_t)env0 * P.nobs + i] = lds[X_STAGE * WS_LANES + i];
      }
    }
  }
  WS_STAMP(role, 23);
}

#undef XS
}  // namespace w8 / w8q
}  // namespace bez

// bez_kernel_ws.h -- wave-specialised fused step kernel (the production hot path at num_envs = 4096).
//
// One workgroup = 4 waves = 64 environments.  Lane l of EVERY wave works on the same env (block*64 + l); the
// wave index selects a ROLE, so the four SIMDs of a CU advance the four independent parts of one articulated
// body at the same time instead of one wave walking all 19 links serially:
//     role 0  left leg   (links 5..10)      passes 1-3, foot ground contact, ball<->leg-box contact
//     role 1  right leg  (links 13..18)     same
//     role 2  leg<->leg self-collision (capsule pairs, while the legs run pass 1), then head + arms
//             (links 1,2 / 3,4 / 11,12) and their guard points
//     role 3  torso, ball, ball<->box winner (legs / torso box), 6x6 root solve, root/ball integration, then the
//             whole post-physics (bookkeeping, reset, observations, reward)
// Chains only meet at the torso: per substep the roles exchange 27 floats (articulated inertia + bias of a chain)
// up and 6 floats (torso acceleration) down through LDS, lds[slot * 64 + lane] (bank = lane, conflict-free),
// with six workgroup barriers.  Role dispatch is a scalar branch on readfirstlane(wave id), every role executes
// the same number of barriers.  Because a wave only ever holds ONE chain, all per-link data of passes 1-3 stays
// in VGPRs (no scratch, no LDS staging of the pass-3 operands).
// Actions (N,18) and observations (N,54) are row-major in HBM: each workgroup's 64 rows are one contiguous block,
// moved with coalesced accesses by all 256 threads and transposed through LDS.  The 22 net-contact-force rows are
// accumulated (mean over the substeps) in LDS by the role that owns the body and written out once, coalesced.
#pragma once
#include "bez_kernels.h"

namespace bez {

constexpr int WS_BLOCK = 256;
constexpr int WS_ENVS = 64;  // envs per workgroup = lanes per wave

// LDS exchange slots (floats per env lane)
enum : int {
  X_ROOT = 0,      // pos3 quat4 lin3 ang3
  X_BALL = 13,     // pos3 lin3 ang3
  X_IA = 22,       // 3 roles x (Sym6 21 + bias 6)
  X_A0 = 103,      // torso spatial acceleration
  X_FL = 109,      // ball<->link force on the link (3) + contact point rel. ball centre (3)
  X_PSUM = 115,    // per chain role: sum of (default - q)^2 over its joints
  X_CAND = 118,    // per leg: depth, link, n(3), P(3), V(6) of its deepest ball/box candidate = 14 floats x 2
  X_FOLD = 146,    // winner side, link, A(6), f0p(3), x(3), xb(3) = 17 floats (written by the root role)
  X_LEGQ = 163,    // per leg: q(6) qd(6) at the start of the substep (read by the self-collision role)
  X_SELF = 187,    // 2 helper parts x [per leg box link (left 5, right 5): bias wrench (6) + reported force (3)] = 2 x 90
  X_CF = 367,      // net contact force rows, up to BEZ_NBE_MAX = 30 bodies (mean over substeps)
  X_HIT = 457,     // cleats asset only: per leg 4 ground-point records x 8 floats (x3 fn0 kn ct ftx0 fty0)
  X_SLOTS = 521
};
constexpr int WS_OBS_STRIDE = 54;  // staging sized for the widest row; rows are unpadded (stride = P.nobs): the staged block IS the contiguous HBM image
constexpr int WS_ACT_STRIDE = 19;
constexpr int WS_LDS_FLOATS = X_SLOTS * WS_ENVS + WS_ENVS * WS_OBS_STRIDE;


constexpr int X_STAGE = X_SLOTS;  // action / observation staging block behind the exchange slots
// joints of the three chain roles (dof index): left leg, right leg, upper (head 0,1; left arm 2,3; right arm 10,11)
BEZ_DEV constexpr int role_ndof(int) { return 6; }
BEZ_DEV constexpr int role_dof(int role, int i) { return role == 0 ? 4 + i : (role == 1 ? 12 + i : (i < 4 ? i : 6 + i)); }
// leg<->leg pairs: PART 0 (upper role, window of the legs' pass 1) owns the hip/thigh x hip/thigh pairs and only walks both legs
// down to the thigh; PART 1 (root role, window of the legs' pass 2) owns every other pair of BEZ_CPAIR
BEZ_DEV constexpr bool self_part_owns(int part, int ia, int ib) { return (part == 0) == (ia <= 1 && ib <= BEZ_NCAP / 2 + 1); }
BEZ_DEV constexpr int self_fk_depth(int part, int) { return part == 0 ? 3 : 6; }

#include "bez_ws_common.inc"


// ------------------------------------------------------------------------------------------------ roles
template <int FIRST, bool PRE, bool POST, bool DR, bool CL>
BEZ_DEV void ws_leg_role(const Params& P, float* lds, int lane, int e, bool active, int side) {
  constexpr int LEN = 6;
  const int n = P.n;
  float* st = P.state;
  float q[LEN], qd[LEN], target[LEN], kps[LEN], kds[LEN], ms[LEN], lo[LEN], hi[LEN];
  ChainDyn D; D.mu = P.mu; D.g = mk(P.g[0], P.g[1], P.g[2]);
#pragma unroll
  for (int i = 0; i < LEN; ++i) {
    constexpr int d0 = FIRST - 1;
    q[i] = st[(size_t)(F_Q + d0 + i) * n + e]; qd[i] = st[(size_t)(F_QD + d0 + i) * n + e];
    kps[i] = 1.f; kds[i] = 1.f; ms[i] = 1.f; lo[i] = (float)BEZ_DOF_LOWER[d0 + i]; hi[i] = (float)BEZ_DOF_UPPER[d0 + i];
    if (DR) {
      if (P.dr_kp) kps[i] = P.dr_kp[(size_t)e * BEZ_ND + d0 + i];
      if (P.dr_kd) kds[i] = P.dr_kd[(size_t)e * BEZ_ND + d0 + i];
      if (P.dr_mass) ms[i] = P.dr_mass[(size_t)e * BEZ_NL + FIRST + i];
      if (P.dr_lower) lo[i] = P.dr_lower[(size_t)e * BEZ_ND + d0 + i];
      if (P.dr_upper) hi[i] = P.dr_upper[(size_t)e * BEZ_ND + d0 + i];
    }
    XS(X_LEGQ + side * 12 + i) = q[i]; XS(X_LEGQ + side * 12 + 6 + i) = qd[i];
  }
  if (DR) {
    if (P.dr_friction) D.mu = P.dr_friction[e];
    if (P.dr_gravity) D.g = mk(P.dr_gravity[(size_t)e * 3], P.dr_gravity[(size_t)e * 3 + 1], P.dr_gravity[(size_t)e * 3 + 2]);
  }
  const bool do_reset = POST && P.reset[e] != 0;  // reset_buf of the previous step (kick_env.py:433-435)
  const uint32_t episode = POST ? P.episode[e] : 0u;
  const bool last_only = (P.flags & BEZ_FLAG_CF_LAST_SUBSTEP) != 0;
  WS_STAMP(side, 0);
  ws_barrier();  // B0: actions staged, root/ball and the leg joint state published
  WS_STAMP(side, 1);
  if (PRE) {
    const float* act = lds + X_SLOTS * WS_ENVS + lane * WS_ACT_STRIDE;
#pragma unroll
    for (int i = 0; i < LEN; ++i) {
      constexpr int d0 = FIRST - 1;
      float a = fminf(fmaxf(act[d0 + i], -P.clip), P.clip);
      float t = a + (float)BEZ_DOF_DEFAULT[d0 + i];
      target[i] = fmaxf(fminf(t, (float)BEZ_DOF_UPPER[d0 + i]), (float)BEZ_DOF_LOWER[d0 + i]);
    }
  } else {
#pragma unroll
    for (int i = 0; i < LEN; ++i) target[i] = st[(size_t)(F_TARGET + FIRST - 1 + i) * n + e];
  }
  for (int s = 0; s < P.substeps; ++s) {
    const bool keep = last_only ? (s == P.substeps - 1) : true;
    const bool first = last_only ? true : (s == 0);
    RootView R = load_root_view(lds, lane);
    LinkInertia LI[LEN]; SV pAl[LEN], Sl[LEN], cbl[LEN];
    BallSel sel;
    sel.link = -1; sel.depth = 0.f; sel.n = sel.P = sel.f0p = sel.x = sel.xb = mk(0, 0, 0); sel.A = sym3zero();
    M3 Eend; V3 rend; SV Vend, Vsel = svzero();
    ws_chain_pass1<FIRST, LEN, true, CL>(P, D, ms, R, q, qd, LI, pAl, Sl, cbl, Eend, rend, Vend, sel, Vsel);
    {  // publish this leg's deepest ball/box candidate; the root role picks the winner and prepares the contact
      const int c0 = X_CAND + side * 14;
      XS(c0) = sel.depth; XS(c0 + 1) = (float)sel.link;
      xs_store_v3(lds, lane, c0 + 2, sel.n); xs_store_v3(lds, lane, c0 + 5, sel.P); xs_store_sv(lds, lane, c0 + 8, Vsel);
    }
    WS_STAMP(side, 2 + 8 * s);
    ws_barrier();  // B1: candidates and the leg<->leg contact wrenches are in LDS
    WS_STAMP(side, 3 + 8 * s);
    // foot ground contact while the root role evaluates the ball<->link contact operands
    Sym6 Kc = sym6zero(); SV pc = svzero();
    ws_ground_points<FIRST + LEN - 1, CL>(P, D.mu, R.root_z, Eend, rend, Vend, Kc, pc, lds, lane, X_HIT + side * 32);
    ws_barrier();  // B1b: fold operands published
    bool mine = (XS(X_FOLD) == (float)side) && (XS(X_FOLD + 1) >= 1.f);
    if (mine) {
      sel.link = (int)XS(X_FOLD + 1);
      sel.A.xx = XS(X_FOLD + 2); sel.A.yy = XS(X_FOLD + 3); sel.A.zz = XS(X_FOLD + 4);
      sel.A.xy = XS(X_FOLD + 5); sel.A.xz = XS(X_FOLD + 6); sel.A.yz = XS(X_FOLD + 7);
      sel.f0p = xs_load_v3(lds, lane, X_FOLD + 8); sel.x = xs_load_v3(lds, lane, X_FOLD + 11); sel.xb = xs_load_v3(lds, lane, X_FOLD + 14);
    } else {
      sel.link = -1;
    }
    P3 p3[LEN];
    Sym6 IA = sym6zero(); SV pA = svzero();
    ws_chain_pass2<FIRST, LEN, true>(P, D, kps, kds, lo, hi, q, qd, target, LI, pAl, Sl, cbl, Kc, pc, mine, sel, p3, IA, pA);
    WS_STAMP(side, 24 + s);
    ws_barrier();  // B1c: both helper parts' leg<->leg contact wrenches are in LDS
    ws_chain_self_correction<LEN>(lds, lane, side, p3, pA);
    xs_store_sym6(lds, lane, X_IA + side * 27, IA, pA);
    BodyContact bcn = body_contact_of(Kc, pc);
    WS_STAMP(side, 4 + 8 * s);
    ws_barrier();  // B2
    WS_STAMP(side, 5 + 8 * s);
    ws_barrier();  // B3: torso acceleration published
    WS_STAMP(side, 6 + 8 * s);
    SV a0 = xs_load_sv(lds, lane, X_A0);
    V3 fl = mk(0, 0, 0), fend = mk(0, 0, 0);
    SV aend = ws_chain_pass3<FIRST, LEN, true, CL>(P, a0, p3, q, qd, mine, sel, fl, fend, lds, lane, keep, first);
    if (mine && sel.link >= 0) { xs_store_v3(lds, lane, X_FL, fl); xs_store_v3(lds, lane, X_FL + 3, sel.xb); }
    if (keep) {
      if constexpr (CL) {  // the foot plate only feels the ball / the other leg; the ground acts on the four cleats
        ws_cf_acc(lds, lane, link_body<CL>(FIRST + LEN - 1), fend, P.cf_w, first);
        ws_cleat_forces<FIRST + LEN - 1>(P, lds, lane, X_HIT + side * 32, aend, first);
      } else {
        ws_cf_acc(lds, lane, link_body<CL>(FIRST + LEN - 1), fend + cf_ground(P, body_contact_force(bcn, aend)), P.cf_w, first);
      }
    }
#pragma unroll
    for (int i = 0; i < LEN; ++i) { XS(X_LEGQ + side * 12 + i) = q[i]; XS(X_LEGQ + side * 12 + 6 + i) = qd[i]; }
    WS_STAMP(side, 7 + 8 * s);
    ws_barrier();  // B4
    WS_STAMP(side, 8 + 8 * s);
    ws_barrier();  // B5: new root/ball state published
    WS_STAMP(side, 9 + 8 * s);
  }
  ws_chain_epilogue<(FIRST == 5 ? 0 : 1), POST>(P, lds, lane, e, active, do_reset, episode, q, qd, target);
  ws_barrier();  // B6
}

template <bool PRE, bool POST, bool DR, bool CL>
BEZ_DEV void ws_upper_role(const Params& P, float* lds, int lane, int e, bool active) {
  // joints (dof index): head 0,1 (links 1,2); left arm 2,3 (links 3,4); right arm 10,11 (links 11,12)
  const int n = P.n;
  float* st = P.state;
  constexpr int DOF[6] = {0, 1, 2, 3, 10, 11};
  float q[6], qd[6], target[6], kps[6], kds[6], ms[6], lo[6], hi[6];
  ChainDyn D; D.mu = P.mu; D.g = mk(P.g[0], P.g[1], P.g[2]);
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    q[i] = st[(size_t)(F_Q + DOF[i]) * n + e]; qd[i] = st[(size_t)(F_QD + DOF[i]) * n + e];
    kps[i] = 1.f; kds[i] = 1.f; ms[i] = 1.f; lo[i] = (float)BEZ_DOF_LOWER[DOF[i]]; hi[i] = (float)BEZ_DOF_UPPER[DOF[i]];
    if (DR) {
      if (P.dr_kp) kps[i] = P.dr_kp[(size_t)e * BEZ_ND + DOF[i]];
      if (P.dr_kd) kds[i] = P.dr_kd[(size_t)e * BEZ_ND + DOF[i]];
      if (P.dr_mass) ms[i] = P.dr_mass[(size_t)e * BEZ_NL + DOF[i] + 1];
      if (P.dr_lower) lo[i] = P.dr_lower[(size_t)e * BEZ_ND + DOF[i]];
      if (P.dr_upper) hi[i] = P.dr_upper[(size_t)e * BEZ_ND + DOF[i]];
    }
  }
  if (DR) {
    if (P.dr_friction) D.mu = P.dr_friction[e];
    if (P.dr_gravity) D.g = mk(P.dr_gravity[(size_t)e * 3], P.dr_gravity[(size_t)e * 3 + 1], P.dr_gravity[(size_t)e * 3 + 2]);
  }
  const bool do_reset = POST && P.reset[e] != 0;
  const uint32_t episode = POST ? P.episode[e] : 0u;
  const bool last_only = (P.flags & BEZ_FLAG_CF_LAST_SUBSTEP) != 0;
  ws_barrier();  // B0
  if (PRE) {
    const float* act = lds + X_SLOTS * WS_ENVS + lane * WS_ACT_STRIDE;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      float a = fminf(fmaxf(act[DOF[i]], -P.clip), P.clip);
      if (DOF[i] < 2) a = 0.f;  // head frozen (kick_env.py:414)
      float t = a + (float)BEZ_DOF_DEFAULT[DOF[i]];
      target[i] = fmaxf(fminf(t, (float)BEZ_DOF_UPPER[DOF[i]]), (float)BEZ_DOF_LOWER[DOF[i]]);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 6; ++i) target[i] = st[(size_t)(F_TARGET + DOF[i]) * n + e];
  }
  BallSel nosel; nosel.link = -1; nosel.depth = 0.f; nosel.n = nosel.P = nosel.f0p = nosel.x = nosel.xb = mk(0, 0, 0); nosel.A = sym3zero();
  for (int s = 0; s < P.substeps; ++s) {
    const bool keep = last_only ? (s == P.substeps - 1) : true;
    const bool first = last_only ? true : (s == 0);
    RootView R = load_root_view(lds, lane);
    {  // leg<->leg self-collision, part 0: overlaps the legs' pass 1
      SelfCaps K;
      ws_self_fk<0>(lds, lane, R.E0, R.V0, K, false, quirk_rz<CL>(P.flags));
      ws_self_pairs<0>(P, D.mu, lds, lane, K);
    }
    P3 p3[6];
    BodyContact bcn[3];
    Sym6 IA = sym6zero(); SV pA = svzero();
    WS_STAMP(2, 2 + 8 * s);
    ws_barrier();  // B1
    WS_STAMP(2, 3 + 8 * s);
    ws_barrier();  // B1b  (only the legs / root exchange data here: the chains below overlap the legs' pass 2)
    {  // three 2-link chains, one after the other
      LinkInertia LI[2]; SV pAl[2], Sl[2], cbl[2]; Sym6 Kc; SV pc; M3 Ee; V3 re; SV Ve, Vs = svzero();
      ws_chain_pass1<1, 2, false, CL>(P, D, ms + 0, R, q + 0, qd + 0, LI, pAl, Sl, cbl, Ee, re, Ve, nosel, Vs);
      Kc = sym6zero(); pc = svzero(); ws_ground_points<2, CL>(P, D.mu, R.root_z, Ee, re, Ve, Kc, pc);
      ws_chain_pass2<1, 2, false>(P, D, kps + 0, kds + 0, lo + 0, hi + 0, q + 0, qd + 0, target + 0, LI, pAl, Sl, cbl, Kc, pc, false, nosel, p3 + 0, IA, pA);
      bcn[0] = body_contact_of(Kc, pc);
      ws_chain_pass1<3, 2, false, CL>(P, D, ms + 2, R, q + 2, qd + 2, LI, pAl, Sl, cbl, Ee, re, Ve, nosel, Vs);
      Kc = sym6zero(); pc = svzero(); ws_ground_points<4, CL>(P, D.mu, R.root_z, Ee, re, Ve, Kc, pc);
      ws_chain_pass2<3, 2, false>(P, D, kps + 2, kds + 2, lo + 2, hi + 2, q + 2, qd + 2, target + 2, LI, pAl, Sl, cbl, Kc, pc, false, nosel, p3 + 2, IA, pA);
      bcn[1] = body_contact_of(Kc, pc);
      ws_chain_pass1<11, 2, false, CL>(P, D, ms + 4, R, q + 4, qd + 4, LI, pAl, Sl, cbl, Ee, re, Ve, nosel, Vs);
      Kc = sym6zero(); pc = svzero(); ws_ground_points<12, CL>(P, D.mu, R.root_z, Ee, re, Ve, Kc, pc);
      ws_chain_pass2<11, 2, false>(P, D, kps + 4, kds + 4, lo + 4, hi + 4, q + 4, qd + 4, target + 4, LI, pAl, Sl, cbl, Kc, pc, false, nosel, p3 + 4, IA, pA);
      bcn[2] = body_contact_of(Kc, pc);
    }
    xs_store_sym6(lds, lane, X_IA + 2 * 27, IA, pA);
    WS_STAMP(2, 4 + 8 * s);
    ws_barrier();  // B1c
    ws_barrier();  // B2
    WS_STAMP(2, 5 + 8 * s);
    ws_barrier();  // B3
    WS_STAMP(2, 6 + 8 * s);
    SV a0 = xs_load_sv(lds, lane, X_A0);
    V3 fl = mk(0, 0, 0), fend = mk(0, 0, 0);
    SV ae0 = ws_chain_pass3<1, 2, false, CL>(P, a0, p3 + 0, q + 0, qd + 0, false, nosel, fl, fend, lds, lane, keep, first);
    SV ae1 = ws_chain_pass3<3, 2, false, CL>(P, a0, p3 + 2, q + 2, qd + 2, false, nosel, fl, fend, lds, lane, keep, first);
    SV ae2 = ws_chain_pass3<11, 2, false, CL>(P, a0, p3 + 4, q + 4, qd + 4, false, nosel, fl, fend, lds, lane, keep, first);
    if (keep) {
      ws_cf_acc(lds, lane, link_body<CL>(2), cf_ground(P, body_contact_force(bcn[0], ae0)), P.cf_w, first);
      ws_cf_acc(lds, lane, link_body<CL>(4), cf_ground(P, body_contact_force(bcn[1], ae1)), P.cf_w, first);
      ws_cf_acc(lds, lane, link_body<CL>(12), cf_ground(P, body_contact_force(bcn[2], ae2)), P.cf_w, first);
    }
    ws_barrier();  // B4
    ws_barrier();  // B5
  }
  ws_chain_epilogue<2, POST>(P, lds, lane, e, active, do_reset, episode, q, qd, target);
  ws_barrier();  // B6
}

template <bool PRE, bool POST, bool DR, bool CL>
BEZ_DEV void ws_root_role(const Params& P, float* lds, int lane, int e, bool active) {
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
  float prev[3] = {0.f, 0.f, 0.f};
  if (POST) {
    progress = P.progress[e]; reset = P.reset[e];
#pragma unroll
    for (int i = 0; i < 3; ++i) prev[i] = ld(F_PREV + i);
  }
  ChainDyn D; D.mu = P.mu; D.g = mk(P.g[0], P.g[1], P.g[2]);
  float ms0 = 1.f;
  if (DR) {
    if (P.dr_friction) D.mu = P.dr_friction[e];
    if (P.dr_gravity) D.g = mk(P.dr_gravity[(size_t)e * 3], P.dr_gravity[(size_t)e * 3 + 1], P.dr_gravity[(size_t)e * 3 + 2]);
    if (P.dr_mass) ms0 = P.dr_mass[(size_t)e * BEZ_NL];
  }
  const bool last_only = (P.flags & BEZ_FLAG_CF_LAST_SUBSTEP) != 0;
  auto publish = [&]() {
    xs_store_v3(lds, lane, X_ROOT, root_pos);
    XS(X_ROOT + 3) = rq[0]; XS(X_ROOT + 4) = rq[1]; XS(X_ROOT + 5) = rq[2]; XS(X_ROOT + 6) = rq[3];
    xs_store_v3(lds, lane, X_ROOT + 7, root_lin); xs_store_v3(lds, lane, X_ROOT + 10, root_ang);
    xs_store_v3(lds, lane, X_BALL, ball_pos); xs_store_v3(lds, lane, X_BALL + 3, ball_lin); xs_store_v3(lds, lane, X_BALL + 6, ball_ang);
  };
  publish();
  WS_STAMP(3, 0);
  ws_barrier();  // B0
  WS_STAMP(3, 1);
  for (int s = 0; s < P.substeps; ++s) {
    const bool keep = last_only ? (s == P.substeps - 1) : true;
    const bool first = last_only ? true : (s == 0);
    const M3 E0 = quat_to_mat(rq[0], rq[1], rq[2], rq[3]);
    const SV V0 = mksv(root_ang, root_lin);
    const V3 bc = ball_pos - root_pos;
    xs_store_v3(lds, lane, X_FL, mk(0, 0, 0)); xs_store_v3(lds, lane, X_FL + 3, mk(0, 0, 0));
    Sym6 IA0 = sym6zero(); SV pA0;
    LinkInertia I0;
    link_inertia<0, CL>(ms0, D.g, E0, mk(0, 0, 0), V0, I0, pA0);
    Sym6 Kc = sym6zero(); SV pc = svzero();
    ws_ground_points<0, CL>(P, D.mu, root_pos.z, E0, mk(0, 0, 0), V0, Kc, pc);
    BodyContact bc0 = body_contact_of(Kc, pc);
    add_link_inertia(IA0, I0);
    add_to(IA0, Kc); pA0 = pA0 + pc;
    BallBody ball = ball_setup(P, D.mu, D.g, ball_pos.z, ball_ang, ball_lin);
    SelfCaps K;  // this wave's share of the leg<->leg self-collision: kinematics now (the legs run pass 1), pairs after B1b
    ws_self_fk<1>(lds, lane, E0, V0, K, false, quirk_rz<CL>(P.flags));
    ws_self_pin<1>(K);
    WS_STAMP(3, 2 + 8 * s);
    ws_barrier();  // B1: both legs' ball/box candidates are in LDS
    WS_STAMP(3, 3 + 8 * s);
    // deepest candidate wins: left leg, right leg, torso box -- in that order on ties, as the box order of the oracle;
    // the contact is evaluated once, here
    BallSel sel;
    {
      const float dl = XS(X_CAND), dr = XS(X_CAND + 14);
      const int side_w = (dr > dl) ? 1 : 0;
      const int c0 = X_CAND + side_w * 14;
      sel.depth = XS(c0); sel.link = (int)XS(c0 + 1);
      sel.n = xs_load_v3(lds, lane, c0 + 2); sel.P = xs_load_v3(lds, lane, c0 + 5);
      sel.A = sym3zero(); sel.f0p = sel.x = sel.xb = mk(0, 0, 0);
      SV Vl = xs_load_sv(lds, lane, c0 + 8);
      test_torso_box(P, E0, mk(0, 0, 0), bc, sel);
      if (sel.link == 0) Vl = V0;
      if (sel.link >= 0) ball_link_contact(P, D.mu, ball_ang, ball_lin, ball, bc, Vl, sel);
      XS(X_FOLD) = (float)side_w; XS(X_FOLD + 1) = (float)sel.link;  // link 0 / -1: neither leg folds anything
      XS(X_FOLD + 2) = sel.A.xx; XS(X_FOLD + 3) = sel.A.yy; XS(X_FOLD + 4) = sel.A.zz;
      XS(X_FOLD + 5) = sel.A.xy; XS(X_FOLD + 6) = sel.A.xz; XS(X_FOLD + 7) = sel.A.yz;
      xs_store_v3(lds, lane, X_FOLD + 8, sel.f0p); xs_store_v3(lds, lane, X_FOLD + 11, sel.x); xs_store_v3(lds, lane, X_FOLD + 14, sel.xb);
    }
    const bool torso_hit = (sel.link == 0);
    if (torso_hit) { add_point_stiffness(IA0, sel.x, sel.A); pA0 = pA0 - wrench_at(sel.x, sel.f0p); }
    ws_barrier();  // B1b
    ws_self_pairs<1>(P, D.mu, lds, lane, K);  // in the window of the legs' pass 2
    WS_STAMP(3, 24 + s);
    ws_barrier();  // B1c
    ws_barrier();  // B2: chain contributions published
    WS_STAMP(3, 5 + 8 * s);
    xs_add_sym6(lds, lane, X_IA + 0 * 27, IA0, pA0);
    xs_add_sym6(lds, lane, X_IA + 1 * 27, IA0, pA0);
    xs_add_sym6(lds, lane, X_IA + 2 * 27, IA0, pA0);
    SV a0 = solve_spd6(IA0, svzero() - pA0);
    xs_store_sv(lds, lane, X_A0, a0);
    WS_STAMP(3, 4 + 8 * s);
    ws_barrier();  // B3
    WS_STAMP(3, 6 + 8 * s);
    V3 fl_t = mk(0, 0, 0);
    if (torso_hit) fl_t = sel.f0p - mul(sel.A, point_of(a0, sel.x));
    if (keep) ws_cf_acc(lds, lane, 0, cf_along(P, fl_t, sel.n) + cf_ground(P, body_contact_force(bc0, a0)), P.cf_w, first);
    V3 vdot = a0.l + cross(root_ang, root_lin);
    root_ang = fma3(a0.a, P.h, root_ang);
    root_lin = fma3(vdot, P.h, root_lin);
    root_pos = fma3(root_lin, P.h, root_pos);
    quat_integrate(rq, root_ang, P.h);
    WS_STAMP(3, 7 + 8 * s);
    ws_barrier();  // B4: ball<->link force published
    WS_STAMP(3, 8 + 8 * s);
    V3 fl = xs_load_v3(lds, lane, X_FL), xb = xs_load_v3(lds, lane, X_FL + 3);
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
    publish();
    ws_barrier();  // B5
    WS_STAMP(3, 9 + 8 * s);
  }
  ws_barrier();  // B6: joint obs slots / pose-error sums / contact-force rows of the chain roles are in LDS
  WS_STAMP(3, 20);
  if (POST) {
    CfOut co;
    co.base = nullptr; co.n = n;
    co.lf = xs_load_v3(lds, lane, X_CF + lfoot_body<CL>() * 3); co.rf = xs_load_v3(lds, lane, X_CF + rfoot_body<CL>() * 3);
    float goal_x = P.goal[0], goal_y = P.goal[1];
    if (P.task != BEZ_TASK_KICK) { goal_x = ld(F_GOAL); goal_y = ld(F_GOAL + 1); }
    int64_t timeout = (progress >= (int64_t)(P.max_len - 1)) ? 1 : 0;  // vec_task.py:331-332
    progress += 1;                                                    // kick_env.py:429
    if (reset != 0) {                                                 // kick_env.py:433-435, 831-850 (root / ball part)
      root_pos = mk(P.bez_init[0], P.bez_init[1], P.bez_init[2]);
      ball_pos = mk(P.ball_init[0], P.ball_init[1], P.ball_init[2]);
#pragma unroll
      for (int i = 0; i < 4; ++i) { rq[i] = P.bez_init[3 + i]; bq[i] = P.ball_init[3 + i]; }
      root_lin = root_ang = ball_lin = ball_ang = mk(0, 0, 0);
      co.lf = co.rf = mk(0, 0, 0);
#pragma unroll
      for (int k = 0; k < (nb_of<CL>() + 1) * 3; ++k) XS(X_CF + k) = 0.f;
      if (active) P.episode[e] = P.episode[e] + 1;
      if (P.task != BEZ_TASK_KICK) {  // walk_env.py:570-575: every env reset by this call receives the same fresh goal
        goal_x = P.goal_draw[0]; goal_y = P.goal_draw[1];
        if (active) { st[(size_t)F_GOAL * n + e] = goal_x; st[(size_t)(F_GOAL + 1) * n + e] = goal_y; }
      }
      progress = 0; reset = 0;
    }
    float pn = (XS(X_PSUM + 2) + XS(X_PSUM + 0)) + XS(X_PSUM + 1);
    float feet[8], rew;
    float* obs_row = lds + X_SLOTS * WS_ENVS + lane * P.nobs;
    float tail[18];
    float cleats[24];
    if (CL) {
#pragma unroll
      for (int k = 0; k < 12; ++k) { cleats[k] = XS(X_CF + BEZ_LCLEAT_BODY_CL * 3 + k); cleats[12 + k] = XS(X_CF + BEZ_RCLEAT_BODY_CL * 3 + k); }
    }
    env_observe_core(P, root_pos, rq, root_lin, root_ang, ball_pos, ball_lin, co, prev, feet, tail, pn, rew, reset, progress, goal_x, goal_y,
                     CL ? cleats : nullptr);
#pragma unroll
    for (int i = 0; i < 18; ++i) if (36 + i < P.nobs) obs_row[36 + i] = tail[i];
    if (!CL) {  // the no-cleats feet logic filters the two foot rows in place (kick_env.py:987-990)
      xs_store_v3(lds, lane, X_CF + BEZ_LFOOT_BODY * 3, co.lf); xs_store_v3(lds, lane, X_CF + BEZ_RFOOT_BODY * 3, co.rf);
    }
    if (active) {
#pragma unroll
      for (int i = 0; i < 3; ++i) st[(size_t)(F_PREV + i) * n + e] = prev[i];
#pragma unroll
      for (int i = 0; i < 8; ++i) st[(size_t)(F_FEET + i) * n + e] = feet[i];
      P.rew[e] = rew; P.reset[e] = reset; P.progress[e] = progress; P.timeout[e] = timeout;
    }
  }
  WS_STAMP(3, 21);
  if (active) {
    auto sv = [&](int f, float v) { st[(size_t)f * n + e] = v; };
    sv(F_ROOT_POS, root_pos.x); sv(F_ROOT_POS + 1, root_pos.y); sv(F_ROOT_POS + 2, root_pos.z);
#pragma unroll
    for (int i = 0; i < 4; ++i) { sv(F_ROOT_QUAT + i, rq[i]); sv(F_BALL_QUAT + i, bq[i]); }
    sv(F_ROOT_LIN, root_lin.x); sv(F_ROOT_LIN + 1, root_lin.y); sv(F_ROOT_LIN + 2, root_lin.z);
    sv(F_ROOT_ANG, root_ang.x); sv(F_ROOT_ANG + 1, root_ang.y); sv(F_ROOT_ANG + 2, root_ang.z);
    sv(F_BALL_POS, ball_pos.x); sv(F_BALL_POS + 1, ball_pos.y); sv(F_BALL_POS + 2, ball_pos.z);
    sv(F_BALL_LIN, ball_lin.x); sv(F_BALL_LIN + 1, ball_lin.y); sv(F_BALL_LIN + 2, ball_lin.z);
    sv(F_BALL_ANG, ball_ang.x); sv(F_BALL_ANG + 1, ball_ang.y); sv(F_BALL_ANG + 2, ball_ang.z);
  }
}

// ---- the kernel.  grid = ceil(N / 64) workgroups of 256 threads.
template <bool PRE, bool POST, bool DR, bool CL>
__global__ __launch_bounds__(WS_BLOCK) void step_kernel_ws(Params P) {
  __shared__ __attribute__((aligned(16))) float lds[WS_LDS_FLOATS];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int role = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int env0 = blockIdx.x * WS_ENVS;
  const int nloc = min(WS_ENVS, P.n - env0);
  const bool active = lane < nloc;
  const int e = env0 + (active ? lane : 0);  // inactive lanes shadow env0 (loads only; every global store is guarded)
  WS_STAMP(role, 22);
  if (PRE) {
    // coalesced stage of this workgroup's contiguous (nloc,18) action block, transposed to [lane][19]
    float* act = lds + X_SLOTS * WS_ENVS;
    const float* src = P.actions + (size_t)env0 * BEZ_ND;
    for (int i = tid; i < nloc * BEZ_ND; i += WS_BLOCK) act[(i / BEZ_ND) * WS_ACT_STRIDE + (i % BEZ_ND)] = src[i];
  }
  // contact-force rows start from zero: bodies nothing touches are never accumulated into
  constexpr int NROW = (nb_of<CL>() + 1) * 3;  // contact-force rows of this asset (robot bodies + ball)
  for (int i = tid; i < NROW * WS_ENVS; i += WS_BLOCK) lds[X_CF * WS_ENVS + i] = 0.f;
  if (role == 0) ws_leg_role<5, PRE, POST, DR, CL>(P, lds, lane, e, active, 0);
  else if (role == 1) ws_leg_role<13, PRE, POST, DR, CL>(P, lds, lane, e, active, 1);
  else if (role == 2) ws_upper_role<PRE, POST, DR, CL>(P, lds, lane, e, active);
  else ws_root_role<PRE, POST, DR, CL>(P, lds, lane, e, active);
  ws_barrier();  // contact-force rows (and, with POST, the observation rows staged by role 3) are complete in LDS
  {
    // net contact force: 66 SoA rows of 64 consecutive envs each -> coalesced
    float* dst = P.state + (size_t)F_CF * P.n + env0;
    for (int i = tid; i < NROW * WS_ENVS; i += WS_BLOCK) {
      const int k = i >> 6, l = i & 63;
      if (l < nloc) dst[(size_t)k * P.n + l] = lds[(X_CF + k) * WS_ENVS + l];
    }
  }
  if (POST) {
    // the staged rows are the contiguous (nloc,54) image of this workgroup's slice of obs_buf: 16-byte copy-out
    const float4* rows = reinterpret_cast<const float4*>(lds + X_SLOTS * WS_ENVS);
    float4* dst = reinterpret_cast<float4*>(P.obs + (size_t)env0 * P.nobs);  // 64 * nobs * 4 B per workgroup: 16-B aligned for 54 and 52
    const int nvec = (nloc * P.nobs) >> 2;
    for (int i = tid; i < nvec; i += WS_BLOCK) dst[i] = rows[i];
    for (int i = (nvec << 2) + tid; i < nloc * P.nobs; i += WS_BLOCK) P.obs[(size_t)env0 * P.nobs + i] = lds[X_SLOTS * WS_ENVS + i];
  }
  WS_STAMP(role, 23);
}

#undef XS
}  // namespace bez

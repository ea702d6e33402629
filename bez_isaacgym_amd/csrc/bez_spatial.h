// bez_spatial.h -- fp32 spatial-algebra primitives for the gfx950 kernels (device only).
//
// Everything is expressed in WORLD-ALIGNED axes about one common reference point O (the torso origin
// at the start of the substep), so spatial transforms between links are the identity and articulated
// inertias accumulate by plain addition.  A spatial vector is [angular; linear].
// A symmetric 6x6 is kept as three 3x3 blocks  [ A  B ; B^T  C ]  with A, C symmetric (21 floats).
#pragma once
#include <hip/hip_runtime.h>

#define BEZ_DEV __device__ __forceinline__

// single-instruction approximations (v_rcp_f32 / v_sqrt_f32 / v_rsq_f32: 1 ulp; v_sin_f32 / v_cos_f32 take revolutions)
BEZ_DEV float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
BEZ_DEV float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
BEZ_DEV float frsq(float x) { return __builtin_amdgcn_rsqf(x); }
BEZ_DEV void fsincos(float x, float* s, float* c) {
  float r = x * 0.15915494309189535f;
  *s = __builtin_amdgcn_sinf(r);
  *c = __builtin_amdgcn_cosf(r);
}

struct V3 { float x, y, z; };
struct Sym3 { float xx, yy, zz, xy, xz, yz; };
struct M3 { float m00, m01, m02, m10, m11, m12, m20, m21, m22; };  // row-major general 3x3
struct SV { V3 a, l; };                                           // spatial motion or force vector
struct Sym6 { Sym3 A; M3 B; Sym3 C; };

BEZ_DEV V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
BEZ_DEV V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
BEZ_DEV V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
BEZ_DEV V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
BEZ_DEV V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
BEZ_DEV V3 operator*(float s, V3 a) { return mk(a.x * s, a.y * s, a.z * s); }
BEZ_DEV float dot(V3 a, V3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
BEZ_DEV V3 cross(V3 a, V3 b) {
  return mk(fmaf(a.y, b.z, -a.z * b.y), fmaf(a.z, b.x, -a.x * b.z), fmaf(a.x, b.y, -a.y * b.x));
}
BEZ_DEV V3 fma3(V3 a, float s, V3 c) { return mk(fmaf(a.x, s, c.x), fmaf(a.y, s, c.y), fmaf(a.z, s, c.z)); }  // a*s + c

BEZ_DEV SV mksv(V3 a, V3 l) { SV r; r.a = a; r.l = l; return r; }
BEZ_DEV SV operator+(SV p, SV q) { return mksv(p.a + q.a, p.l + q.l); }
BEZ_DEV SV operator-(SV p, SV q) { return mksv(p.a - q.a, p.l - q.l); }
BEZ_DEV SV operator*(SV p, float s) { return mksv(p.a * s, p.l * s); }
BEZ_DEV float dot(SV p, SV q) { return dot(p.a, q.a) + dot(p.l, q.l); }
BEZ_DEV SV svzero() { return mksv(mk(0, 0, 0), mk(0, 0, 0)); }
// motion cross product V x S  and force cross product V x* F
BEZ_DEV SV crm(SV V, SV S) { return mksv(cross(V.a, S.a), cross(V.a, S.l) + cross(V.l, S.a)); }
BEZ_DEV SV crf(SV V, SV F) { return mksv(cross(V.a, F.a) + cross(V.l, F.l), cross(V.a, F.l)); }
// wrench (about O) of a force f applied at x
BEZ_DEV SV wrench_at(V3 x, V3 f) { return mksv(cross(x, f), f); }
// velocity / acceleration of the body point at x given the body's spatial vector about O
BEZ_DEV V3 point_of(SV V, V3 x) { return V.l + cross(V.a, x); }

BEZ_DEV V3 mul(const Sym3& S, V3 v) {
  return mk(fmaf(S.xx, v.x, fmaf(S.xy, v.y, S.xz * v.z)), fmaf(S.xy, v.x, fmaf(S.yy, v.y, S.yz * v.z)),
            fmaf(S.xz, v.x, fmaf(S.yz, v.y, S.zz * v.z)));
}
BEZ_DEV V3 mul(const M3& M, V3 v) {
  return mk(fmaf(M.m00, v.x, fmaf(M.m01, v.y, M.m02 * v.z)), fmaf(M.m10, v.x, fmaf(M.m11, v.y, M.m12 * v.z)),
            fmaf(M.m20, v.x, fmaf(M.m21, v.y, M.m22 * v.z)));
}
BEZ_DEV V3 mulT(const M3& M, V3 v) {
  return mk(fmaf(M.m00, v.x, fmaf(M.m10, v.y, M.m20 * v.z)), fmaf(M.m01, v.x, fmaf(M.m11, v.y, M.m21 * v.z)),
            fmaf(M.m02, v.x, fmaf(M.m12, v.y, M.m22 * v.z)));
}
BEZ_DEV SV mul(const Sym6& I, SV v) { return mksv(mul(I.A, v.a) + mul(I.B, v.l), mulT(I.B, v.a) + mul(I.C, v.l)); }

BEZ_DEV Sym3 sym3zero() { Sym3 s; s.xx = s.yy = s.zz = s.xy = s.xz = s.yz = 0.f; return s; }
BEZ_DEV M3 m3zero() { M3 m; m.m00 = m.m01 = m.m02 = m.m10 = m.m11 = m.m12 = m.m20 = m.m21 = m.m22 = 0.f; return m; }
BEZ_DEV Sym6 sym6zero() { Sym6 I; I.A = sym3zero(); I.B = m3zero(); I.C = sym3zero(); return I; }
BEZ_DEV void add_to(Sym3& d, const Sym3& s) { d.xx += s.xx; d.yy += s.yy; d.zz += s.zz; d.xy += s.xy; d.xz += s.xz; d.yz += s.yz; }
BEZ_DEV void add_to(M3& d, const M3& s) {
  d.m00 += s.m00; d.m01 += s.m01; d.m02 += s.m02; d.m10 += s.m10; d.m11 += s.m11; d.m12 += s.m12; d.m20 += s.m20; d.m21 += s.m21; d.m22 += s.m22;
}
BEZ_DEV void add_to(Sym6& d, const Sym6& s) { add_to(d.A, s.A); add_to(d.B, s.B); add_to(d.C, s.C); }
// d += k * u u^T (symmetric part kept)
BEZ_DEV void add_outer(Sym3& d, V3 u, float k) {
  float kx = k * u.x, ky = k * u.y, kz = k * u.z;
  d.xx = fmaf(kx, u.x, d.xx); d.yy = fmaf(ky, u.y, d.yy); d.zz = fmaf(kz, u.z, d.zz);
  d.xy = fmaf(kx, u.y, d.xy); d.xz = fmaf(kx, u.z, d.xz); d.yz = fmaf(ky, u.z, d.yz);
}
// d += k * u v^T
BEZ_DEV void add_outer(M3& d, V3 u, V3 v, float k) {
  float kx = k * u.x, ky = k * u.y, kz = k * u.z;
  d.m00 = fmaf(kx, v.x, d.m00); d.m01 = fmaf(kx, v.y, d.m01); d.m02 = fmaf(kx, v.z, d.m02);
  d.m10 = fmaf(ky, v.x, d.m10); d.m11 = fmaf(ky, v.y, d.m11); d.m12 = fmaf(ky, v.z, d.m12);
  d.m20 = fmaf(kz, v.x, d.m20); d.m21 = fmaf(kz, v.y, d.m21); d.m22 = fmaf(kz, v.z, d.m22);
}
// I += k * w w^T for a spatial vector w
BEZ_DEV void add_outer(Sym6& I, SV w, float k) { add_outer(I.A, w.a, k); add_outer(I.B, w.a, w.l, k); add_outer(I.C, w.l, k); }

// I += J^T K J, J = [-[x]x  1]: a symmetric 3x3 point "stiffness" K acting at x (about O)
BEZ_DEV void add_point_stiffness(Sym6& I, V3 x, const Sym3& K) {
  // M = [x]x K  (rows of [x]x: (0,-z,y), (z,0,-x), (-y,x,0))
  M3 M;
  M.m00 = fmaf(-x.z, K.xy, x.y * K.xz); M.m01 = fmaf(-x.z, K.yy, x.y * K.yz); M.m02 = fmaf(-x.z, K.yz, x.y * K.zz);
  M.m10 = fmaf(x.z, K.xx, -x.x * K.xz); M.m11 = fmaf(x.z, K.xy, -x.x * K.yz); M.m12 = fmaf(x.z, K.xz, -x.x * K.zz);
  M.m20 = fmaf(-x.y, K.xx, x.x * K.xy); M.m21 = fmaf(-x.y, K.xy, x.x * K.yy); M.m22 = fmaf(-x.y, K.xz, x.x * K.yz);
  add_to(I.B, M);
  add_to(I.C, K);
  // A += M [x]x^T ; column j of [x]x^T is row j of [x]x
  I.A.xx += fmaf(-x.z, M.m01, x.y * M.m02);
  I.A.xy += fmaf(x.z, M.m00, -x.x * M.m02);
  I.A.xz += fmaf(-x.y, M.m00, x.x * M.m01);
  I.A.yy += fmaf(x.z, M.m10, -x.x * M.m12);
  I.A.yz += fmaf(-x.y, M.m10, x.x * M.m11);
  I.A.zz += fmaf(-x.y, M.m20, x.x * M.m21);
}

// Same for K = diag(kt, kt, kn) (ground contact: tangential / normal point stiffness): the zero terms are dropped.
//   M = [x]x K = [[0, -z kt, y kn], [z kt, 0, -x kn], [-y kt, x kt, 0]],  A += M [x]x^T,  B += M,  C += K
BEZ_DEV void add_point_stiffness_diag(Sym6& I, V3 x, float kt, float kn) {
  const float xt = x.x * kt, yt = x.y * kt, zt = x.z * kt, xn = x.x * kn, yn = x.y * kn;
  I.B.m01 -= zt; I.B.m02 += yn; I.B.m10 += zt; I.B.m12 -= xn; I.B.m20 -= yt; I.B.m21 += xt;
  I.C.xx += kt; I.C.yy += kt; I.C.zz += kn;
  I.A.xx += fmaf(x.z, zt, x.y * yn);
  I.A.yy += fmaf(x.z, zt, x.x * xn);
  I.A.zz += fmaf(x.y, yt, x.x * xt);
  I.A.xy -= x.x * yn;
  I.A.xz -= x.x * zt;
  I.A.yz -= x.y * zt;
}

// rotation matrix (body->world) of an xyzw unit quaternion
BEZ_DEV M3 quat_to_mat(float x, float y, float z, float w) {
  M3 R;
  R.m00 = 1.f - 2.f * (y * y + z * z); R.m01 = 2.f * (x * y - z * w); R.m02 = 2.f * (x * z + y * w);
  R.m10 = 2.f * (x * y + z * w); R.m11 = 1.f - 2.f * (x * x + z * z); R.m12 = 2.f * (y * z - x * w);
  R.m20 = 2.f * (x * z - y * w); R.m21 = 2.f * (y * z + x * w); R.m22 = 1.f - 2.f * (x * x + y * y);
  return R;
}
BEZ_DEV V3 col(const M3& E, int k) {
  return k == 0 ? mk(E.m00, E.m10, E.m20) : (k == 1 ? mk(E.m01, E.m11, E.m21) : mk(E.m02, E.m12, E.m22));
}
BEZ_DEV void set_col(M3& E, int k, V3 v) {
  if (k == 0) { E.m00 = v.x; E.m10 = v.y; E.m20 = v.z; }
  else if (k == 1) { E.m01 = v.x; E.m11 = v.y; E.m21 = v.z; }
  else { E.m02 = v.x; E.m12 = v.y; E.m22 = v.z; }
}
// E * R_axis(angle): rotate the two columns orthogonal to `axis` (0/1/2); s, c = sin/cos of the signed angle
BEZ_DEV M3 rotate_about(const M3& E, int axis, float s, float c) {
  M3 R = E;
  int i = (axis + 1) % 3, j = (axis + 2) % 3;
  V3 ci = col(E, i), cj = col(E, j);
  set_col(R, i, ci * c + cj * s);
  set_col(R, j, cj * c - ci * s);
  return R;
}
// world-frame rotational inertia  E * diag/sym(Il) * E^T
BEZ_DEV Sym3 rotate_inertia(const M3& E, const Sym3& Il) {
  // T = E * Il
  M3 T;
  T.m00 = fmaf(E.m00, Il.xx, fmaf(E.m01, Il.xy, E.m02 * Il.xz)); T.m01 = fmaf(E.m00, Il.xy, fmaf(E.m01, Il.yy, E.m02 * Il.yz)); T.m02 = fmaf(E.m00, Il.xz, fmaf(E.m01, Il.yz, E.m02 * Il.zz));
  T.m10 = fmaf(E.m10, Il.xx, fmaf(E.m11, Il.xy, E.m12 * Il.xz)); T.m11 = fmaf(E.m10, Il.xy, fmaf(E.m11, Il.yy, E.m12 * Il.yz)); T.m12 = fmaf(E.m10, Il.xz, fmaf(E.m11, Il.yz, E.m12 * Il.zz));
  T.m20 = fmaf(E.m20, Il.xx, fmaf(E.m21, Il.xy, E.m22 * Il.xz)); T.m21 = fmaf(E.m20, Il.xy, fmaf(E.m21, Il.yy, E.m22 * Il.yz)); T.m22 = fmaf(E.m20, Il.xz, fmaf(E.m21, Il.yz, E.m22 * Il.zz));
  Sym3 R;
  R.xx = fmaf(T.m00, E.m00, fmaf(T.m01, E.m01, T.m02 * E.m02));
  R.xy = fmaf(T.m00, E.m10, fmaf(T.m01, E.m11, T.m02 * E.m12));
  R.xz = fmaf(T.m00, E.m20, fmaf(T.m01, E.m21, T.m02 * E.m22));
  R.yy = fmaf(T.m10, E.m10, fmaf(T.m11, E.m11, T.m12 * E.m12));
  R.yz = fmaf(T.m10, E.m20, fmaf(T.m11, E.m21, T.m12 * E.m22));
  R.zz = fmaf(T.m20, E.m20, fmaf(T.m21, E.m21, T.m22 * E.m22));
  return R;
}
// the same for Il = diag(a, b, c)
BEZ_DEV Sym3 rotate_inertia_diag(const M3& E, float a, float b, float c) {
  const float t00 = E.m00 * a, t01 = E.m01 * b, t02 = E.m02 * c, t10 = E.m10 * a, t11 = E.m11 * b, t12 = E.m12 * c;
  const float t20 = E.m20 * a, t21 = E.m21 * b, t22 = E.m22 * c;
  Sym3 R;
  R.xx = fmaf(t00, E.m00, fmaf(t01, E.m01, t02 * E.m02));
  R.xy = fmaf(t00, E.m10, fmaf(t01, E.m11, t02 * E.m12));
  R.xz = fmaf(t00, E.m20, fmaf(t01, E.m21, t02 * E.m22));
  R.yy = fmaf(t10, E.m10, fmaf(t11, E.m11, t12 * E.m12));
  R.yz = fmaf(t10, E.m20, fmaf(t11, E.m21, t12 * E.m22));
  R.zz = fmaf(t20, E.m20, fmaf(t21, E.m21, t22 * E.m22));
  return R;
}
// general 3x3 inverse by cofactors
BEZ_DEV M3 inverse(const M3& a) {
  float c00 = fmaf(a.m11, a.m22, -a.m12 * a.m21), c01 = fmaf(a.m12, a.m20, -a.m10 * a.m22), c02 = fmaf(a.m10, a.m21, -a.m11 * a.m20);
  float det = fmaf(a.m00, c00, fmaf(a.m01, c01, a.m02 * c02));
  float id = frcp(det);
  M3 r;
  r.m00 = c00 * id; r.m01 = fmaf(a.m02, a.m21, -a.m01 * a.m22) * id; r.m02 = fmaf(a.m01, a.m12, -a.m02 * a.m11) * id;
  r.m10 = c01 * id; r.m11 = fmaf(a.m00, a.m22, -a.m02 * a.m20) * id; r.m12 = fmaf(a.m02, a.m10, -a.m00 * a.m12) * id;
  r.m20 = c02 * id; r.m21 = fmaf(a.m01, a.m20, -a.m00 * a.m21) * id; r.m22 = fmaf(a.m00, a.m11, -a.m01 * a.m10) * id;
  return r;
}
BEZ_DEV M3 matmul(const M3& a, const M3& b) {
  M3 r;
  r.m00 = fmaf(a.m00, b.m00, fmaf(a.m01, b.m10, a.m02 * b.m20)); r.m01 = fmaf(a.m00, b.m01, fmaf(a.m01, b.m11, a.m02 * b.m21)); r.m02 = fmaf(a.m00, b.m02, fmaf(a.m01, b.m12, a.m02 * b.m22));
  r.m10 = fmaf(a.m10, b.m00, fmaf(a.m11, b.m10, a.m12 * b.m20)); r.m11 = fmaf(a.m10, b.m01, fmaf(a.m11, b.m11, a.m12 * b.m21)); r.m12 = fmaf(a.m10, b.m02, fmaf(a.m11, b.m12, a.m12 * b.m22));
  r.m20 = fmaf(a.m20, b.m00, fmaf(a.m21, b.m10, a.m22 * b.m20)); r.m21 = fmaf(a.m20, b.m01, fmaf(a.m21, b.m11, a.m22 * b.m21)); r.m22 = fmaf(a.m20, b.m02, fmaf(a.m21, b.m12, a.m22 * b.m22));
  return r;
}
BEZ_DEV M3 to_m3(const Sym3& s) { M3 m; m.m00 = s.xx; m.m01 = s.xy; m.m02 = s.xz; m.m10 = s.xy; m.m11 = s.yy; m.m12 = s.yz; m.m20 = s.xz; m.m21 = s.yz; m.m22 = s.zz; return m; }

// Solve I a = b for SPD Sym6 by LDL^T on the packed lower triangle (compile-time indices only).
BEZ_DEV SV solve_spd6(const Sym6& I, SV b) {
  float L[6][6];
  // unpack (lower triangle)
  L[0][0] = I.A.xx; L[1][0] = I.A.xy; L[1][1] = I.A.yy; L[2][0] = I.A.xz; L[2][1] = I.A.yz; L[2][2] = I.A.zz;
  L[3][0] = I.B.m00; L[3][1] = I.B.m10; L[3][2] = I.B.m20;   // B^T rows
  L[4][0] = I.B.m01; L[4][1] = I.B.m11; L[4][2] = I.B.m21;
  L[5][0] = I.B.m02; L[5][1] = I.B.m12; L[5][2] = I.B.m22;
  L[3][3] = I.C.xx; L[4][3] = I.C.xy; L[4][4] = I.C.yy; L[5][3] = I.C.xz; L[5][4] = I.C.yz; L[5][5] = I.C.zz;
  float d[6], x[6] = {b.a.x, b.a.y, b.a.z, b.l.x, b.l.y, b.l.z};
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    float dj = L[j][j];
#pragma unroll
    for (int k = 0; k < j; ++k) dj = fmaf(-L[j][k] * L[j][k], d[k], dj);
    d[j] = dj;
    float inv = frcp(dj);
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      float s = L[i][j];
#pragma unroll
      for (int k = 0; k < j; ++k) s = fmaf(-L[i][k] * L[j][k], d[k], s);
      L[i][j] = s * inv;
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
#pragma unroll
    for (int k = 0; k < i; ++k) x[i] = fmaf(-L[i][k], x[k], x[i]);
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) x[i] = x[i] * frcp(d[i]);
#pragma unroll
  for (int i = 5; i >= 0; --i) {
#pragma unroll
    for (int k = i + 1; k < 6; ++k) x[i] = fmaf(-L[k][i], x[k], x[i]);
  }
  return mksv(mk(x[0], x[1], x[2]), mk(x[3], x[4], x[5]));
}

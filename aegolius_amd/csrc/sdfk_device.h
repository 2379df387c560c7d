// sdfk_device.h — per-point SDF math for gfx950, shared verbatim by
//   * the interpreter kernel (sdfk_interp.hip, compiled by hipcc) and
//   * every topology-specialised kernel (text of this file is embedded in libsdfk.so and
//     handed to hiprtc in front of the generated straight-line body),
// so both paths execute the same fp32 instruction sequences (both are built with
// -ffp-contract=off; every fused multiply-add below is spelled out).
//
// Conventions
//   V3        : one point (x, y, z) held in VGPRs.
//   P         : this instruction's slice of the parameter table. It is addressed with
//               wave-uniform offsets only, so the compiler fetches it with s_load_* through the
//               scalar data cache into SGPRs — shape parameters never occupy VGPRs or LDS bandwidth.
//   T         : base of the variable-length tables (poly-lines, point sets, convex pieces).
//   Host side : every constant that does not depend on the point (reciprocals, sin/cos of fixed
//               angles, R^T t, half sizes …) is computed in float64 by aegolius_amd/_lower.py
//               and rounded once to fp32. Divisions by constants therefore appear here as
//               multiplications by a pre-rounded reciprocal.
//   Citations : "C/" = Code/spomso/spomso/cores/ of the reference (read-only study copy).
//
// No MFMA: this is scalar per-point math; the bound is HBM (16 B/point) for shallow trees and
// VALU issue for deep ones.
#ifndef SDFK_DEVICE_H
#define SDFK_DEVICE_H

#define SDFK_DEV static __device__ __forceinline__

// Launch indices straight from the hardware registers. The HIP spellings (blockIdx.x, threadIdx.x ...) go through
// device-library functions (__ockl_get_group_id ...) that are NOT inlined into kernels built with -mno-amdgpu-ieee
// (the attribute differs from the library's, so the inliner refuses): they stay real calls whose result comes back in
// a VGPR, and everything derived from the workgroup index then counts as divergent — vector address arithmetic and
// vector loads where scalar ones would do. The builtins below are SGPR reads.
SDFK_DEV unsigned sdfk_bx() { return __builtin_amdgcn_workgroup_id_x(); }
SDFK_DEV unsigned sdfk_by() { return __builtin_amdgcn_workgroup_id_y(); }
SDFK_DEV unsigned sdfk_bz() { return __builtin_amdgcn_workgroup_id_z(); }
SDFK_DEV unsigned sdfk_tx() { return __builtin_amdgcn_workitem_id_x(); }
SDFK_DEV unsigned sdfk_gx() { return __builtin_amdgcn_grid_size_x() / __builtin_amdgcn_workgroup_size_x(); }   // gridDim.x

// Lane value types. `float` = one point per lane; `f2` = TWO points per lane, which lets the
// compiler use the packed-fp32 VALU forms (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32): measured on
// MI355X a packed instruction costs ~1.2x a plain one and does 2x the work (tools/valu_peak.hip),
// and this path is VALU-issue-bound for deep trees. min/max/sqrt/abs/compare have no packed form
// and are issued per element. The hot operators below are templates over the lane type so that the
// interpreter (float) and the specialised kernels (f2) run the SAME operation sequence per element
// (packed and plain fma/mul/add round identically) — results are bit-identical.
typedef float f2 __attribute__((ext_vector_type(2)));

template <typename T> struct V3T { T x, y, z; };
typedef V3T<float> V3;
typedef V3T<f2> V3P;

#define SDFK_TWO_PI 6.283185307179586f

// ---------------------------------------------------------------------------------------------
// small helpers (float and f2 overloads)
// ---------------------------------------------------------------------------------------------
template <typename T> SDFK_DEV T sp(float v);                      // broadcast a parameter to the lane type
template <> SDFK_DEV float sp<float>(float v) { return v; }
template <> SDFK_DEV f2 sp<f2>(float v) { f2 r = {v, v}; return r; }

SDFK_DEV float sd_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
SDFK_DEV f2 sd_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
// v_sqrt_f32: 1 ulp, half rate — instead of the ~10-instruction correctly-rounded expansion.
SDFK_DEV float sd_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
SDFK_DEV f2 sd_sqrt(f2 x) { f2 r = {__builtin_amdgcn_sqrtf(x.x), __builtin_amdgcn_sqrtf(x.y)}; return r; }
SDFK_DEV float sd_min(float a, float b) { return __builtin_fminf(a, b); }
SDFK_DEV f2 sd_min(f2 a, f2 b) { f2 r = {__builtin_fminf(a.x, b.x), __builtin_fminf(a.y, b.y)}; return r; }
SDFK_DEV float sd_max(float a, float b) { return __builtin_fmaxf(a, b); }
SDFK_DEV f2 sd_max(f2 a, f2 b) { f2 r = {__builtin_fmaxf(a.x, b.x), __builtin_fmaxf(a.y, b.y)}; return r; }
// one v_max_f32, no canonicalising pre-op (inputs are plain loaded / computed numbers)
SDFK_DEV float sd_rawmax(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
SDFK_DEV float sd_abs(float a) { return __builtin_fabsf(a); }
SDFK_DEV f2 sd_abs(f2 a) { f2 r = {__builtin_fabsf(a.x), __builtin_fabsf(a.y)}; return r; }
// np.sign: -1, 0, +1
SDFK_DEV float sd_sign(float v) { return (v > 0.0f) ? 1.0f : ((v < 0.0f) ? -1.0f : 0.0f); }
SDFK_DEV f2 sd_sign(f2 v) { f2 r = {sd_sign(v.x), sd_sign(v.y)}; return r; }
// np.clip(v, lo, hi) == minimum(maximum(v, lo), hi)
template <typename T> SDFK_DEV T sd_clip(T v, float lo, float hi) { return sd_min(sd_max(v, sp<T>(lo)), sp<T>(hi)); }
template <typename T> SDFK_DEV T sd_clip01(T v) { return sd_min(sd_max(v, sp<T>(0.0f)), sp<T>(1.0f)); }
template <typename T> SDFK_DEV T sd_max0(T v) { return sd_max(v, sp<T>(0.0f)); }
template <typename T> SDFK_DEV T sd_min0(T v) { return sd_min(v, sp<T>(0.0f)); }
template <typename T> SDFK_DEV T sd_len2(T x, T y) { return sd_sqrt(sd_fma(x, x, y * y)); }
template <typename T> SDFK_DEV T sd_len3(T x, T y, T z) { return sd_sqrt(sd_fma(x, x, sd_fma(y, y, z * z))); }
// dot products of lane values with PARAMETERS (b*)
template <typename T> SDFK_DEV T sd_dot2(T ax, T ay, float bx, float by) {
    return sd_fma(ax, sp<T>(bx), ay * by);
}
template <typename T> SDFK_DEV T sd_dot3(T ax, T ay, T az, float bx, float by, float bz) {
    return sd_fma(ax, sp<T>(bx), sd_fma(ay, sp<T>(by), az * bz));
}
// pair plumbing for the operators that exist in scalar form only
SDFK_DEV V3 sd_lo(V3P p) { V3 r = {p.x.x, p.y.x, p.z.x}; return r; }
SDFK_DEV V3 sd_hi(V3P p) { V3 r = {p.x.y, p.y.y, p.z.y}; return r; }
SDFK_DEV V3P sd_join(V3 a, V3 b) { V3P r = {{a.x, b.x}, {a.y, b.y}, {a.z, b.z}}; return r; }
// np.mod(a, d): floored modulo, result carries the sign of the divisor. inv_d = 1/d (host, f64->f32).
// One fma recovers a - q*d exactly once q is right; the two fix-ups repair an off-by-one q.
// (Round 3, measured on the modification-chain config at 1025^3, four variants in one process: a branch-free form of
//  the fix-ups — sign folded in with xor, no scalar branch on d — has fewer instructions in the listing but executes
//  more of them, 3.64 against 3.50 ms; sincos / atan2 on packed pairs (v_pk_fma for reduction and polynomials) 3.73 ms:
//  the packed forms need their constants in SGPR pairs and cost 1.3 plain issues each. Both were removed again. A native
//  pair form of op_infrep — packed add / mul / fma, fix-ups as additions of d, -d or 0 — is bit-identical and exactly as
//  fast as the pair wrapper around this scalar function, 3.45 vs 3.45 ms in alternation: the compiler pairs the two
//  scalar calls by itself.)
// all ones where the sign bit of x is set. Inline asm on purpose: written as a shift the optimiser turns it back into
// compare + select, and the select's constants into moves (16 VALU instructions per sd_mod in round 3's listing, 11 now).
SDFK_DEV int sd_signmask(float x) {
    int m;
    asm("v_ashrrev_i32_e32 %0, 31, %1" : "=v"(m) : "v"(x));
    return m;
}
SDFK_DEV float sd_mod(float a, float d, float inv_d) {
    float q = __builtin_floorf(a * inv_d);
    float r = sd_fma(-q, d, a);
    if (d > 0.0f) {            // wave-uniform: d is a parameter
        // r < 0 ? r + d : r   (r is never -0 here: an exact cancellation gives +0)
        r += __builtin_bit_cast(float, sd_signmask(r) & __builtin_bit_cast(int, d));
        // r >= d ? r - d : r   (t = r - d carries the exact sign of the comparison)
        const float t = r - d;
        const int m = sd_signmask(t);
        r = __builtin_bit_cast(float, (m & __builtin_bit_cast(int, r)) | (~m & __builtin_bit_cast(int, t)));
    } else {
        r = (r > 0.0f) ? r + d : r;
        r = (r <= d) ? r - d : r;
    }
    return r;
}
// distance from p to segment a + t*ba, t in [0,1]; inv = 1/dot(ba,ba)
SDFK_DEV float sd_seg3_sq(float px, float py, float pz, const float* __restrict__ S) {
    float pax = px - S[0], pay = py - S[1], paz = pz - S[2];
    float h = sd_clip01(sd_dot3(pax, pay, paz, S[3], S[4], S[5]) * S[6]);
    float dx = sd_fma(-S[3], h, pax), dy = sd_fma(-S[4], h, pay), dz = sd_fma(-S[5], h, paz);
    return sd_fma(dx, dx, sd_fma(dy, dy, dz * dz));
}
SDFK_DEV float sd_seg2_sq(float px, float py, const float* __restrict__ S) {
    float pax = px - S[0], pay = py - S[1];
    float h = sd_clip01(sd_dot2(pax, pay, S[2], S[3]) * S[4]);
    float dx = sd_fma(-S[2], h, pax), dy = sd_fma(-S[3], h, pay);
    return sd_fma(dx, dx, dy * dy);
}

// =============================================================================================
// coordinate -> coordinate
// signature: V3 f(V3 p, const float* P, const float* T, int imm)
// =============================================================================================
template <typename T> SDFK_DEV V3T<T> op_movc(V3T<T> p, const float* __restrict__, const float* __restrict__, int) { return p; }

// C/transformations.py:232-242  co' = (R^T co)/s - R^T t.  P = M(9, row major, R^T/s) , c(3) = R^T t
// The sum is associated x, y first and z last, so that for a run of points sharing x and y (a grid row)
// the first two thirds (op_xform_base) are one value per run and only op_xform_z depends on the point:
// the culling kernel computes the base once per brick and evaluates 3 instead of 9 fmas per point, with
// bit-identical results.
template <typename T> SDFK_DEV V3T<T> op_xform_base(T x, T y, const float* __restrict__ P) {
    V3T<T> b;
    b.x = sd_fma(sp<T>(P[1]), y, sd_fma(sp<T>(P[0]), x, sp<T>(-P[9])));
    b.y = sd_fma(sp<T>(P[4]), y, sd_fma(sp<T>(P[3]), x, sp<T>(-P[10])));
    b.z = sd_fma(sp<T>(P[7]), y, sd_fma(sp<T>(P[6]), x, sp<T>(-P[11])));
    return b;
}
template <typename T> SDFK_DEV V3T<T> op_xform_z(V3T<T> base, T z, const float* __restrict__ P) {
    V3T<T> q = {sd_fma(sp<T>(P[2]), z, base.x), sd_fma(sp<T>(P[5]), z, base.y), sd_fma(sp<T>(P[8]), z, base.z)};
    return q;
}
template <typename T> SDFK_DEV V3T<T> op_xform(V3T<T> p, const float* __restrict__ P, const float* __restrict__, int) {
    return op_xform_z(op_xform_base(p.x, p.y, P), p.z, P);
}
// Flat grids (z = 0 exactly, rows along y): M2 * 0 = 0 adds to the sum without a rounding, so
// op_xform(x, y, 0) = fma(M1, y, fma(M0, x, -c)) bit for bit (up to the sign of a zero) — the x part is one value
// per row (op_xform_base_x) and only op_xform_y depends on the point.
template <typename T> SDFK_DEV V3T<T> op_xform_base_x(T x, const float* __restrict__ P) {
    V3T<T> b = {sd_fma(sp<T>(P[0]), x, sp<T>(-P[9])), sd_fma(sp<T>(P[3]), x, sp<T>(-P[10])), sd_fma(sp<T>(P[6]), x, sp<T>(-P[11]))};
    return b;
}
template <typename T> SDFK_DEV V3T<T> op_xform_y(V3T<T> base, T y, const float* __restrict__ P) {
    V3T<T> q = {sd_fma(sp<T>(P[1]), y, base.x), sd_fma(sp<T>(P[4]), y, base.y), sd_fma(sp<T>(P[7]), y, base.z)};
    return q;
}
// R = I, s = 1 (I·co and co/1.0 are exact in the reference): q = p - t.  Also move_sdf C/modifications.py:1283
template <typename T> SDFK_DEV V3T<T> op_xlate(V3T<T> p, const float* __restrict__ P, const float* __restrict__, int) {
    V3T<T> q = {p.x - P[0], p.y - P[1], p.z - P[2]};
    return q;
}
// q = O p : the six named shears + shear(), rotate_sdf.  C/modifications.py:579-774, :1323-1324
template <typename T> SDFK_DEV V3T<T> op_lin3(V3T<T> p, const float* __restrict__ P, const float* __restrict__, int) {
    V3T<T> q;
    q.x = sd_fma(sp<T>(P[0]), p.x, sd_fma(sp<T>(P[1]), p.y, P[2] * p.z));
    q.y = sd_fma(sp<T>(P[3]), p.x, sd_fma(sp<T>(P[4]), p.y, P[5] * p.z));
    q.z = sd_fma(sp<T>(P[6]), p.x, sd_fma(sp<T>(P[7]), p.y, P[8] * p.z));
    return q;
}
// q = p / k with P[0] = 1/k
template <typename T> SDFK_DEV V3T<T> op_cscale(V3T<T> p, const float* __restrict__ P, const float* __restrict__, int) {
    V3T<T> q = {p.x * P[0], p.y * P[0], p.z * P[0]};
    return q;
}
// elongation C/modifications.py:88-93 ; P = ev/2
template <typename T> SDFK_DEV V3T<T> op_elongate(V3T<T> p, const float* __restrict__ P, const float* __restrict__, int) {
    V3T<T> q = {p.x - sd_clip(p.x, -P[0], P[0]), p.y - sd_clip(p.y, -P[1], P[1]), p.z - sd_clip(p.z, -P[2], P[2])};
    return q;
}
// revolution C/modifications.py:426-431 ; P = radius
template <typename T> SDFK_DEV V3T<T> op_revolve(V3T<T> p, const float* __restrict__ P, const float* __restrict__, int) {
    V3T<T> q = {sd_len2(p.x, p.z) - P[0], p.y, sp<T>(0.0f)};
    return q;
}
// xy <- [[c, s], [-s, c]] xy ; P = (c, s). z untouched.
template <typename T> SDFK_DEV V3T<T> op_rot2d(V3T<T> p, const float* __restrict__ P, const float* __restrict__, int) {
    V3T<T> q = {sd_fma(sp<T>(P[0]), p.x, P[1] * p.y), sd_fma(sp<T>(-P[1]), p.x, P[0] * p.y), p.z};
    return q;
}
// axis_revolution tail C/modifications.py:460-466 (input already rotated in place); P = (c, s, radius)
template <typename T> SDFK_DEV V3T<T> op_axrev(V3T<T> p, const float* __restrict__ P, const float* __restrict__, int) {
    T m = sd_len2(p.x, p.z);
    // qo[:2] = rot.T · (m, y)   with rot.T = [[c, -s], [s, c]]
    V3T<T> q = {sd_fma(sp<T>(P[0]), m, -P[1] * p.y) - P[2], sd_fma(sp<T>(P[1]), m, P[0] * p.y), sp<T>(0.0f)};
    return q;
}
template <typename T> SDFK_DEV V3T<T> op_zeroz(V3T<T> p, const float* __restrict__, const float* __restrict__, int) {
    V3T<T> q = {p.x, p.y, sp<T>(0.0f)};
    return q;
}
// sin and cos of one angle. |x| <= 8192: Cody-Waite reduction by pi/2 in three parts (exact under fma for these k)
// and the classic minimax polynomials on [-pi/4, pi/4] (Cephes sinf / cosf coefficients, < 1.5 ulp), inlined: 26
// VALU instructions against ~60 for the call into ocml's sincosf with its Payne-Hanek tail, which still serves
// larger arguments and NaN. (The fast intrinsic __sincosf is NOT accurate enough: 1e-6 parity needs ~1 ulp.)
SDFK_DEV void sd_sincos(float x, float* s, float* c) {
    if (!(sd_abs(x) <= 8192.0f)) {
        sincosf(x, s, c);
        return;
    }
    const float k = __builtin_rintf(x * 0.636619772367581343f);
    float r = sd_fma(k, -1.5703125f, x);
    r = sd_fma(k, -4.837512969970703125e-4f, r);
    r = sd_fma(k, -7.54978995489188216e-8f, r);
    const float z = r * r;
    const float ps = sd_fma(r * z, sd_fma(z, sd_fma(z, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), r);
    const float pc = sd_fma(z * z, sd_fma(z, sd_fma(z, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f),
                            sd_fma(z, -0.5f, 1.0f));
    const int q = (int)k;
    const float ss = (q & 1) ? pc : ps, cc = (q & 1) ? ps : pc;
    // sign flips as bit 1 of q (sine) and of q + 1 (cosine) moved onto the sign bit: three two-operand instructions each
    // instead of and + compare + select
    *s = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, ss) ^ (((unsigned)q << 30) & 0x80000000u));
    *c = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, cc) ^ (((unsigned)(q + 1) << 30) & 0x80000000u));
}

// atan2 with numpy's conventions for finite arguments (atan2(0, 0) = 0, signed zeros of y kept): one division
// t = min/max (v_rcp_f32 + one Newton step on the quotient), atan(t) = t * P(t^2) on [0, 1] with a degree-10
// Chebyshev-fitted P (1.4 ulp in fp32), octant fix-ups with pi/2 and pi split in two parts. ~30 VALU instructions,
// inlined, against the 42 + call of ocml's atan2f; infinities / NaN go to ocml.
SDFK_DEV float sd_atan2(float y, float x) {
    const float ax = sd_abs(x), ay = sd_abs(y);
    const float mx = sd_rawmax(ax, ay), mn = (ax < ay) ? ax : ay;
    if (!(mx < 3.0e38f)) return atan2f(y, x);
    const float rc = __builtin_amdgcn_rcpf(mx);
    float t = mn * rc;
    t = sd_fma(sd_fma(-mx, t, mn), rc, t);
    t = (mx == 0.0f) ? 0.0f : t;
    const float z = t * t;
    const float c[11] = {1.0f, -0.333333224f, 0.199995577f, -0.142785758f, 0.110507712f, -0.0878504366f, 0.0668528154f, -0.0439284109f, 0.0219129473f, -0.0070306696f, 0.00105760747f};
    float p = c[10];
#pragma unroll
    for (int i = 9; i >= 1; --i) p = sd_fma(p, z, c[i]);
    float a = sd_fma(t * z, p, t);                      // t * (1 + z Q(z)): the leading term carries no rounding
    a = (ay > ax) ? (1.57079637050628662109375f - a) + -4.37113882867379e-8f : a;      // pi/2 = hi + lo
    a = __builtin_signbit(x) ? (3.1415927410125732421875f - a) + -8.74227765734758e-8f : a;   // pi = hi + lo
    return __builtin_copysignf(a, y);
}

// twist C/modifications.py:517-522 ; P = pitch
SDFK_DEV V3 op_twist(V3 p, const float* __restrict__ P, const float* __restrict__, int) {
    float s, c;
    sd_sincos(P[0] * p.z, &s, &c);
    V3 q = {sd_fma(c, p.x, -s * p.y), sd_fma(s, p.x, c * p.y), p.z};
    return q;
}
// bend C/modifications.py:546-571 ; P = (R, cos(a/2), sin(a/2), R*a/2, R*sin(a/2), R*(1-cos(a/2)))
SDFK_DEV V3 op_bend(V3 p, const float* __restrict__ P, const float* __restrict__, int) {
    float R = P[0], c = P[1], s = P[2];
    float yr = p.y - R;
    float phi = sd_atan2(p.x, -yr);
    float qx = R * phi;
    float qy = -R + sd_len2(p.x, yr);
    if (P[3] <= sd_abs(qx)) {                     // rigid continuation past the bent arc
        float sg = sd_sign(p.x);
        float wx = p.x - P[4] * sg;
        float wy = p.y - P[5];
        const float ss = (p.x >= 0.0f) ? s : -s;           // one select on the sine instead of one per output: same bits
        const float rx = sd_fma(c, wx, ss * wy), ry = sd_fma(-ss, wx, c * wy);
        qx = rx + P[3] * sg;
        qy = ry;
    }
    V3 q = {qx, qy, p.z};
    return q;
}
// infinite_repetition C/modifications.py:819-821 ; P = half(3), d(3), 1/d(3)
SDFK_DEV V3 op_infrep(V3 p, const float* __restrict__ P, const float* __restrict__, int) {
    V3 q = {sd_mod(p.x + P[0], P[3], P[6]) - P[0], sd_mod(p.y + P[1], P[4], P[7]) - P[1],
            sd_mod(p.z + P[2], P[5], P[8]) - P[2]};
    return q;
}
// finite_repetition C/modifications.py:846-868 ; P = c(3), d(3), s(3), s/2(3), 1/s(3)
SDFK_DEV float sd_finrep1(float x, float c, float d, float s, float hs, float inv_s) {
    float v = sd_abs(x) - c;
    v = (x < 0.0f) ? -v : v;                      // v -= 2 v (x < 0)
    float u = sd_mod(x - d, s, inv_s) - hs;
    return (x >= -d && x <= d) ? u : v;
}
SDFK_DEV V3 op_finrep(V3 p, const float* __restrict__ P, const float* __restrict__, int) {
    V3 q = {sd_finrep1(p.x, P[0], P[3], P[6], P[9], P[12]), sd_finrep1(p.y, P[1], P[4], P[7], P[10], P[13]),
            sd_finrep1(p.z, P[2], P[5], P[8], P[11], P[14])};
    return q;
}
// symmetry C/modifications.py:948-952 ; imm = axis
template <typename T> SDFK_DEV V3T<T> op_symmetry(V3T<T> p, const float* __restrict__, const float* __restrict__, int imm) {
    V3T<T> q = {imm == 0 ? sd_abs(p.x) : p.x, imm == 1 ? sd_abs(p.y) : p.y, imm == 2 ? sd_abs(p.z) : p.z};
    return q;
}
// mirror tail C/modifications.py:990-993 ; P = l/2
template <typename T> SDFK_DEV V3T<T> op_foldx(V3T<T> p, const float* __restrict__ P, const float* __restrict__, int) {
    V3T<T> q = {sd_abs(p.x) - P[0], p.y, p.z};
    return q;
}
// rotational_symmetry tail C/modifications.py:1023-1029 ; P = (angle, angle/2, 1/angle, radius)
SDFK_DEV V3 op_rotsym(V3 p, const float* __restrict__ P, const float* __restrict__, int) {
    float phi = sd_atan2(p.y, p.x);
    phi = (phi < 0.0f) ? SDFK_TWO_PI + phi : phi;
    phi = sd_mod(phi, P[0], P[2]) - P[1];
    float r = sd_len2(p.x, p.y);
    float s, c;
    sd_sincos(phi, &s, &c);
    V3 q = {sd_fma(r, c, -P[3]), r * s, p.z};
    return q;
}
// linear_instancing tail C/modifications.py:1070-1084 ; P = (l/2, lo, hi, l/2-d, s, d, 1/s, n>2)
SDFK_DEV V3 op_lininst(V3 p, const float* __restrict__ P, const float* __restrict__, int) {
    float v = sd_abs(p.x) - P[0];
    v = (p.x < 0.0f) ? -v : v;
    if (P[7] != 0.0f) {                           // wave-uniform
        float u = sd_mod(p.x - P[3], P[4], P[6]) - P[5];
        v = (p.x >= P[1] && p.x <= P[2]) ? u : v;
    }
    V3 q = {v, p.y, p.z};
    return q;
}
// curve_instancing family C/modifications.py:1108-1263 ; P = (count, table offset, has_frames)
// table: per instance centre(3) [frame rows dx,dy,dz (9)]
SDFK_DEV V3 op_curveinst(V3 p, const float* __restrict__ P, const float* __restrict__ T, int) {
    int n = (int)P[0];
    int stride = (P[2] != 0.0f) ? 12 : 3;
    const float* __restrict__ tab = T + (int)P[1];
    float best = 3.0e38f;
    int bi = 0;
    for (int i = 0; i < n; ++i) {
        const float* __restrict__ c = tab + i * stride;
        float dx = p.x - c[0], dy = p.y - c[1], dz = p.z - c[2];
        float d2 = sd_fma(dx, dx, sd_fma(dy, dy, dz * dz));
        if (d2 < best) { best = d2; bi = i; }
    }
    const float* __restrict__ c = tab + bi * stride;   // per-lane gather (L1/L2 resident table)
    V3 v = {p.x - c[0], p.y - c[1], p.z - c[2]};
    if (P[2] != 0.0f) {
        V3 w = {sd_dot3(c[3], c[4], c[5], v.x, v.y, v.z), sd_dot3(c[6], c[7], c[8], v.x, v.y, v.z),
                sd_dot3(c[9], c[10], c[11], v.x, v.y, v.z)};
        return w;
    }
    return v;
}

// =============================================================================================
// primitives: coordinate -> value
// signature: float f(V3 p, const float* P, const float* T)
// =============================================================================================
// sdf_x/y/z C/sdf_3D.py:13-22 ; P = (offset, axis)
template <typename T> SDFK_DEV T prim_axis(V3T<T> p, const float* __restrict__ P, const float* __restrict__) {
    T v = (P[1] == 0.0f) ? p.x : ((P[1] == 1.0f) ? p.y : p.z);     // wave-uniform selection
    return v - P[0];
}
// sdf_sphere C/sdf_3D.py:25-27
template <typename T> SDFK_DEV T prim_sphere(V3T<T> p, const float* __restrict__ P, const float* __restrict__) {
    return sd_len3(p.x, p.y, p.z) - P[0];
}
// sdf_cylinder C/sdf_3D.py:30-37 ; P = (radius, height/2)
template <typename T> SDFK_DEV T prim_cylinder(V3T<T> p, const float* __restrict__ P, const float* __restrict__) {
    T d0 = sd_len2(p.x, p.y) - P[0];
    T d1 = sd_abs(p.z) - P[1];
    T t1 = sd_min0(sd_max(d0, d1));
    T t2 = sd_len2(sd_max0(d0), sd_max0(d1));
    return t1 + t2;
}
// sdf_box C/sdf_3D.py:40-47 ; P = size/2
template <typename T> SDFK_DEV T prim_box(V3T<T> p, const float* __restrict__ P, const float* __restrict__) {
    T qx = sd_abs(p.x) - P[0], qy = sd_abs(p.y) - P[1], qz = sd_abs(p.z) - P[2];
    T t1 = sd_len3(sd_max0(qx), sd_max0(qy), sd_max0(qz));
    T t2 = sd_min0(sd_max(qx, sd_max(qy, qz)));
    return t1 + t2;
}
// sdf_torus C/sdf_3D.py:50-53 ; P = (R, r)
template <typename T> SDFK_DEV T prim_torus(V3T<T> p, const float* __restrict__ P, const float* __restrict__) {
    T a = sd_len2(p.x, p.y) - P[0];
    return sd_len2(a, p.z) - P[1];
}
// sdf_chainlink C/sdf_3D.py:56-61 ; P = (R, r, length/2)  (length as passed to sdf_chainlink)
template <typename T> SDFK_DEV T prim_chainlink(V3T<T> p, const float* __restrict__ P, const float* __restrict__) {
    T x = p.x - sd_clip(p.x, -P[2], P[2]);
    T a = sd_len2(x, p.y) - P[0];
    return sd_len2(a, p.z) - P[1];
}
// sdf_braid C/sdf_3D.py:64-75 ; P = (length/2, R, r, pitch)
SDFK_DEV float prim_braid(V3 p, const float* __restrict__ P, const float* __restrict__) {
    float s, c;
    sd_sincos(P[3] * p.z, &s, &c);
    // co[:2] = rot[0]*x + rot[1]*y with rot = [[c, s], [-s, c]]  ->  x' = c x - s y ; y' = s x + c y
    float x = sd_fma(c, p.x, -s * p.y);
    float y = sd_fma(s, p.x, c * p.y);
    float z = p.z - sd_clip(p.z, -P[0], P[0]);
    float a = sd_len2(x, z) - P[1];
    return sd_len2(a, y) - P[2];
}
// shared by arcs / sectors: rotate xy by the mid angle with [[c, s], [-s, c]]
SDFK_DEV void sd_rotmid(float x, float y, float c, float s, float* ox, float* oy) {
    *ox = sd_fma(c, x, s * y);
    *oy = sd_fma(-s, x, c * y);
}
// sdf_arc_3d C/sdf_3D.py:78-96 ; P = (R, r, cos mid, sin mid, |end - mid|)
SDFK_DEV float prim_arc3d(V3 p, const float* __restrict__ P, const float* __restrict__) {
    float x, y;
    sd_rotmid(p.x, p.y, P[2], P[3], &x, &y);
    y = sd_abs(y);
    float psi = sd_clip(sd_atan2(y, x), 0.0f, P[4]);
    float s, c;
    sd_sincos(psi, &s, &c);
    return sd_len3(x - P[0] * c, y - P[0] * s, p.z) - P[1];
}
// sdf_plane C/sdf_3D.py:99-102 ; P = (n̂(3), offset)
template <typename T> SDFK_DEV T prim_plane(V3T<T> p, const float* __restrict__ P, const float* __restrict__) {
    return sd_dot3(p.x, p.y, p.z, P[0], P[1], P[2]) - P[3];
}
// sudf_plane C/sdf_3D.py:105-108 ; P = (n̂(3), thickness/2)
template <typename T> SDFK_DEV T prim_uplane(V3T<T> p, const float* __restrict__ P, const float* __restrict__) {
    return sd_abs(sd_dot3(p.x, p.y, p.z, P[0], P[1], P[2])) - P[3];
}
// sdf_segment_3d C/sdf_3D.py:111-118 ; P = a(3), ba(3), 1/dot(ba,ba)
SDFK_DEV float prim_segment3(V3 p, const float* __restrict__ P, const float* __restrict__) {
    return sd_sqrt(sd_seg3_sq(p.x, p.y, p.z, P));
}
// sdf_cone C/sdf_3D.py:121-136 ; P = (q0 = H tan a, q1 = -H, z offset, 1/dot(q,q), 1/q0)
template <typename T> SDFK_DEV T prim_cone(V3T<T> p, const float* __restrict__ P, const float* __restrict__) {
    float q0 = P[0], q1 = P[1];
    T w0 = sd_len2(p.x, p.y);
    T w1 = p.z - P[2];
    T t1 = sd_clip01(sd_dot2(w0, w1, q0, q1) * P[3]);
    T ax = sd_fma(sp<T>(-q0), t1, w0), ay = sd_fma(sp<T>(-q1), t1, w1);
    T t2 = sd_clip01(w0 * P[4]);
    T bx = sd_fma(sp<T>(-q0), t2, w0), by = w1 - q1;
    T d = sd_min(sd_fma(ax, ax, ay * ay), sd_fma(bx, bx, by * by));
    T s = sd_max(-sd_fma(w0, sp<T>(q1), -w1 * q0), -(w1 - q1));
    return sd_sqrt(d) * sd_sign(s);
}
// sdf_infinite_cone / sdf_oriented_infinite_cone C/sdf_3D.py:139-157 ; P = (sin a, cos a, oriented)
SDFK_DEV float prim_infcone(V3 p, const float* __restrict__ P, const float* __restrict__) {
    float q0 = sd_len2(p.x, p.y), q1 = -p.z;
    float t = sd_max(sd_dot2(q0, q1, P[0], P[1]), 0.0f);
    float d = sd_len2(sd_fma(-P[0], t, q0), sd_fma(-P[1], t, q1));
    if (P[2] != 0.0f) d = (sd_fma(q0, P[1], -q1 * P[0]) < 0.0f) ? -d : d;
    return d;
}
// common tail of sdf_solid_angle C/sdf_3D.py:168-183 and sdf_sector C/sdf_2D.py:114-129
// (x, y) already rotated / folded ; P = (radius, cos mid, sin mid, half width, cos hw, sin hw)
SDFK_DEV float sd_sector_tail(float x, float y, const float* __restrict__ P) {
    float phi = sd_atan2(y, x);
    float psi = sd_clip(phi, 0.0f, P[3]);
    float s, c;
    sd_sincos(psi, &s, &c);
    float length = sd_len2(x - P[0] * c, y - P[0] * s);
    float t = sd_clip(sd_dot2(x, y, P[4], P[5]), 0.0f, P[0]);
    float m = sd_len2(sd_fma(-P[4], t, x), sd_fma(-P[5], t, y));
    float out = sd_min(m, length);
    return (sd_len2(x, y) <= P[0] && phi <= P[3]) ? -out : out;
}
SDFK_DEV float prim_solidangle(V3 p, const float* __restrict__ P, const float* __restrict__) {
    float x, y;
    sd_rotmid(p.x, p.y, P[1], P[2], &x, &y);
    return sd_sector_tail(x, sd_len2(y, p.z), P);
}
// sdf_triangle_3d C/sdf_3D.py:186-214
// P = a b c (9) | s1 s2 s3 (9) | cross(s_i, normal) (9) | normal (3) | 1/dot(s_i,s_i) (3) | 1/dot(n,n)
SDFK_DEV float sd_edge_sq(float cx, float cy, float cz, const float* __restrict__ s, float inv) {
    float h = sd_clip01(sd_dot3(s[0], s[1], s[2], cx, cy, cz) * inv);
    float dx = sd_fma(s[0], h, -cx), dy = sd_fma(s[1], h, -cy), dz = sd_fma(s[2], h, -cz);
    return sd_fma(dx, dx, sd_fma(dy, dy, dz * dz));
}
SDFK_DEV float prim_triangle3(V3 p, const float* __restrict__ P, const float* __restrict__) {
    float ax = p.x - P[0], ay = p.y - P[1], az = p.z - P[2];
    float bx = p.x - P[3], by = p.y - P[4], bz = p.z - P[5];
    float cx = p.x - P[6], cy = p.y - P[7], cz = p.z - P[8];
    float m = sd_sign(sd_dot3(P[18], P[19], P[20], ax, ay, az)) + sd_sign(sd_dot3(P[21], P[22], P[23], bx, by, bz)) +
              sd_sign(sd_dot3(P[24], P[25], P[26], cx, cy, cz));
    float e;
    if (m < 2.0f) {
        e = sd_min(sd_min(sd_edge_sq(ax, ay, az, P + 9, P[30]), sd_edge_sq(bx, by, bz, P + 12, P[31])),
                   sd_edge_sq(cx, cy, cz, P + 15, P[32]));
    } else {
        float d = sd_dot3(P[27], P[28], P[29], ax, ay, az);
        e = d * d * P[33];
    }
    return sd_sqrt(e);
}
// sdf_quad_3d C/sdf_3D.py:217-250
// P = a b c d (12) | s1..s4 (12) | cross(s_i, normal) (12) | normal (3) | 1/dot(s_i,s_i) (4) | 1/dot(n,n)
SDFK_DEV float prim_quad3(V3 p, const float* __restrict__ P, const float* __restrict__) {
    float ax = p.x - P[0], ay = p.y - P[1], az = p.z - P[2];
    float bx = p.x - P[3], by = p.y - P[4], bz = p.z - P[5];
    float cx = p.x - P[6], cy = p.y - P[7], cz = p.z - P[8];
    float dx = p.x - P[9], dy = p.y - P[10], dz = p.z - P[11];
    float m = sd_sign(sd_dot3(P[24], P[25], P[26], ax, ay, az)) + sd_sign(sd_dot3(P[27], P[28], P[29], bx, by, bz)) +
              sd_sign(sd_dot3(P[30], P[31], P[32], cx, cy, cz)) + sd_sign(sd_dot3(P[33], P[34], P[35], dx, dy, dz));
    float e;
    if (m < 3.0f) {
        float e12 = sd_min(sd_edge_sq(ax, ay, az, P + 12, P[39]), sd_edge_sq(bx, by, bz, P + 15, P[40]));
        float e43 = sd_min(sd_edge_sq(dx, dy, dz, P + 21, P[42]), sd_edge_sq(cx, cy, cz, P + 18, P[41]));
        e = sd_min(e43, e12);
    } else {
        float d = sd_dot3(P[36], P[37], P[38], ax, ay, az);
        e = d * d * P[43];
    }
    return sd_sqrt(e);
}
// sdf_segmented_line_3d C/sdf_3D.py:264-271 ; P = (segment count, table offset) ; table rows a(3) ba(3) inv
SDFK_DEV float prim_segline3(V3 p, const float* __restrict__ P, const float* __restrict__ T) {
    int n = (int)P[0];
    const float* __restrict__ tab = T + (int)P[1];
    float best = 1.0e32f;                                   // (1e16)^2
    for (int i = 0; i < n; ++i) best = sd_min(best, sd_seg3_sq(p.x, p.y, p.z, tab + 7 * i));
    return sd_sqrt(best);
}
// KDTree nearest-point distance C/sdf_3D.py:253-261,274-286 ; P = (point count, table offset) ; rows xyz
SDFK_DEV float prim_nearest3(V3 p, const float* __restrict__ P, const float* __restrict__ T) {
    int n = (int)P[0];
    const float* __restrict__ tab = T + (int)P[1];
    float best = 3.0e38f;
    for (int i = 0; i < n; ++i) {
        float dx = p.x - tab[3 * i], dy = p.y - tab[3 * i + 1], dz = p.z - tab[3 * i + 2];
        best = sd_min(best, sd_fma(dx, dx, sd_fma(dy, dy, dz * dz)));
    }
    return sd_sqrt(best);
}

// The same distance for large tables: points sorted into leaves of <= 32 (k-d median splits, built on
// the host: sdfk_point_tree_build), <= 32 consecutive leaves under one middle box, <= 32 consecutive middle boxes under
// one root box (round 4: the third level — a cloud of 1.7 M points has 1,696 middle boxes, and a walk that looked at each
// of them twice per query was a serial chain of milliseconds). Table at T + P[1]:
//   root boxes   P[0] x 8 floats : lo(3), hi(3), index of the first middle box, middle boxes
//   middle boxes        8 floats : lo(3), hi(3), index of the first leaf box, leaf boxes
//   leaf boxes          8 floats : lo(3), hi(3), index of the first point, points      (indices relative to the table)
//   points              3 floats (2-D tables carry z = 0 and the query's z is ignored)
// A box is skipped only if a slightly deflated lower bound of its distance exceeds the best squared distance found
// so far, and every visited point goes through the arithmetic of prim_nearest3/2: the result is bit-identical to the
// brute-force scan (the min does not depend on the order).
SDFK_DEV float sd_boxdist2(V3 p, const float* __restrict__ b) {
    const float dx = sd_max0(sd_max(b[0] - p.x, p.x - b[3])), dy = sd_max0(sd_max(b[1] - p.y, p.y - b[4]));
    const float dz = sd_max0(sd_max(b[2] - p.z, p.z - b[5]));
    return 0.99999f * sd_fma(dx, dx, sd_fma(dy, dy, dz * dz));
}
SDFK_DEV float sd_scan_leaf(V3 p, const float* __restrict__ tab, const float* __restrict__ leaf, float best) {
    const float* __restrict__ pt = tab + (int)leaf[6];
    const int n = (int)leaf[7];
    for (int i = 0; i < n; ++i) {
        float dx = p.x - pt[3 * i], dy = p.y - pt[3 * i + 1], dz = p.z - pt[3 * i + 2];
        best = sd_min(best, sd_fma(dx, dx, sd_fma(dy, dy, dz * dz)));
    }
    return best;
}
// index of the child of `box` (8-float rows at tab + box[6], box[7] of them) nearest to p
SDFK_DEV int sd_nearest_child(V3 p, const float* __restrict__ tab, const float* __restrict__ box) {
    const float* __restrict__ kids = tab + (int)box[6];
    const int n = (int)box[7];
    int b = 0;
    float bd = 3.0e38f;
    for (int k = 0; k < n; ++k) {
        const float d = sd_boxdist2(p, kids + 8 * k);
        if (d < bd) { bd = d; b = k; }
    }
    return b;
}
SDFK_DEV float prim_neartree(V3 p, const float* __restrict__ P, const float* __restrict__ T) {
    const int n_root = (int)P[0];
    const float* __restrict__ tab = T + (int)P[1];
    if (P[2] == 2.0f) p.z = 0.0f;
    // a first bound: the nearest root box, its nearest middle box, its nearest leaf
    int br = 0;
    float bd = 3.0e38f;
    for (int r = 0; r < n_root; ++r) {
        const float d = sd_boxdist2(p, tab + 8 * r);
        if (d < bd) { bd = d; br = r; }
    }
    const float* __restrict__ mid0 = tab + (int)tab[8 * br + 6] + 8 * sd_nearest_child(p, tab, tab + 8 * br);
    const float* __restrict__ leaf0 = tab + (int)mid0[6] + 8 * sd_nearest_child(p, tab, mid0);
    float best = sd_scan_leaf(p, tab, leaf0, 3.0e38f);
    // pruned sweep over everything else
    for (int r = 0; r < n_root; ++r) {
        const float* __restrict__ rb = tab + 8 * r;
        if (sd_boxdist2(p, rb) > best) continue;
        const float* __restrict__ mids = tab + (int)rb[6];
        const int nm = (int)rb[7];
        for (int m = 0; m < nm; ++m) {
            const float* __restrict__ mb = mids + 8 * m;
            if (sd_boxdist2(p, mb) > best) continue;
            const float* __restrict__ leaves = tab + (int)mb[6];
            const int nl = (int)mb[7];
            for (int l = 0; l < nl; ++l) {
                const float* __restrict__ lf = leaves + 8 * l;
                if (lf == leaf0 || sd_boxdist2(p, lf) > best) continue;
                best = sd_scan_leaf(p, tab, lf, best);
            }
        }
    }
    return sd_sqrt(best);
}

// Nearest INSTANCE through the same tree (curve_instancing with more centres than a scan should visit): the tree's
// points are the centres in leaf order; P[5] points at the ORIGINAL index of each of them (same order), P[4] is the
// index of the tree's first point, P[3] the instance rows (centre(3) [+ frame rows (9)], original order). Among
// centres at exactly the same fp32 distance the lowest original index wins — what the scan's strict `<` does — so the
// chosen instance, and with it the field, is bit-identical to the scan's whatever the visiting order (with thousands
// of centres along a curve such ties are common: neighbours differ by less than an ulp of d^2 near the foot point).
SDFK_DEV void sd_scan_leaf_idx(V3 p, const float* __restrict__ tab, const float* __restrict__ leaf,
                               const float* __restrict__ orig, int point_base, float* best, int* who) {
    const int first = (int)leaf[6];
    const float* __restrict__ pt = tab + first;
    const float* __restrict__ id = orig + (first - point_base) / 3;
    const int n = (int)leaf[7];
    for (int i = 0; i < n; ++i) {
        float dx = p.x - pt[3 * i], dy = p.y - pt[3 * i + 1], dz = p.z - pt[3 * i + 2];
        const float d2 = sd_fma(dx, dx, sd_fma(dy, dy, dz * dz));
        const int o = (int)id[i];
        if (d2 < *best || (d2 == *best && o < *who)) {
            *best = d2;
            *who = o;
        }
    }
}
SDFK_DEV V3 op_curveinstt(V3 p, const float* __restrict__ P, const float* __restrict__ T, int) {
    const int n_root = (int)P[0];
    const float* __restrict__ tab = T + (int)P[1];
    const float* __restrict__ orig = T + (int)P[5];
    const int point_base = (int)P[4];
    int br = 0;
    float bd = 3.0e38f;
    for (int r = 0; r < n_root; ++r) {
        const float d = sd_boxdist2(p, tab + 8 * r);
        if (d < bd) { bd = d; br = r; }
    }
    const float* __restrict__ mid0 = tab + (int)tab[8 * br + 6] + 8 * sd_nearest_child(p, tab, tab + 8 * br);
    const float* __restrict__ leaf0 = tab + (int)mid0[6] + 8 * sd_nearest_child(p, tab, mid0);
    float best = 3.0e38f;
    int who = 0x7fffffff;
    sd_scan_leaf_idx(p, tab, leaf0, orig, point_base, &best, &who);
    for (int r = 0; r < n_root; ++r) {
        const float* __restrict__ rb = tab + 8 * r;
        if (sd_boxdist2(p, rb) > best) continue;              // a box AT the best distance is still visited (ties)
        const float* __restrict__ mids = tab + (int)rb[6];
        const int nm = (int)rb[7];
        for (int m = 0; m < nm; ++m) {
            const float* __restrict__ mb = mids + 8 * m;
            if (sd_boxdist2(p, mb) > best) continue;
            const float* __restrict__ leaves = tab + (int)mb[6];
            const int nl = (int)mb[7];
            for (int l = 0; l < nl; ++l) {
                const float* __restrict__ lf = leaves + 8 * l;
                if (lf == leaf0 || sd_boxdist2(p, lf) > best) continue;
                sd_scan_leaf_idx(p, tab, lf, orig, point_base, &best, &who);
            }
        }
    }
    const int stride = (P[2] != 0.0f) ? 12 : 3;
    const float* __restrict__ c = T + (int)P[3] + who * stride;
    V3 v = {p.x - c[0], p.y - c[1], p.z - c[2]};
    if (P[2] != 0.0f) {
        V3 w = {sd_dot3(c[3], c[4], c[5], v.x, v.y, v.z), sd_dot3(c[6], c[7], c[8], v.x, v.y, v.z),
                sd_dot3(c[9], c[10], c[11], v.x, v.y, v.z)};
        return w;
    }
    return v;
}

// ---- 2-D primitives (z ignored) -------------------------------------------------------------
// sdf_circle C/sdf_2D.py:12-14
template <typename T> SDFK_DEV T prim_circle(V3T<T> p, const float* __restrict__ P, const float* __restrict__) {
    return sd_len2(p.x, p.y) - P[0];
}
// sdf_neu_circle C/sdf_2D.py:17-19 ; P = (radius, ord, kind) kind: 0 general p-norm, 1 = +inf, 2 = -inf, 3 = ord 0
SDFK_DEV float prim_neucircle(V3 p, const float* __restrict__ P, const float* __restrict__) {
    float ax = sd_abs(p.x), ay = sd_abs(p.y);
    float l;
    if (P[2] == 1.0f) l = sd_max(ax, ay);
    else if (P[2] == 2.0f) l = sd_min(ax, ay);
    else if (P[2] == 3.0f) l = ((ax != 0.0f) ? 1.0f : 0.0f) + ((ay != 0.0f) ? 1.0f : 0.0f);
    else if (P[1] == 1.0f) l = ax + ay;
    else if (P[1] == 2.0f) l = sd_len2(ax, ay);
    else l = powf(powf(ax, P[1]) + powf(ay, P[1]), 1.0f / P[1]);
    return l - P[0];
}
// sdf_box_2d C/sdf_2D.py:22-28 ; P = size/2
template <typename T> SDFK_DEV T prim_box2(V3T<T> p, const float* __restrict__ P, const float* __restrict__) {
    T dx = sd_abs(p.x) - P[0], dy = sd_abs(p.y) - P[1];
    return sd_len2(sd_max0(dx), sd_max0(dy)) + sd_min0(sd_max(dx, dy));
}
// sdf_segment_2d C/sdf_2D.py:31-38 ; P = a(2), ba(2), 1/dot(ba,ba)
SDFK_DEV float prim_segment2(V3 p, const float* __restrict__ P, const float* __restrict__) {
    return sd_sqrt(sd_seg2_sq(p.x, p.y, P));
}
// sdf_rounded_box_2d C/sdf_2D.py:41-57 ; P = size/2 (2), rounding (4)
SDFK_DEV float prim_rbox2(V3 p, const float* __restrict__ P, const float* __restrict__) {
    float r = P[2];
    r = (p.x > 0.0f) ? P[3] : r;
    r = (p.y > 0.0f) ? P[4] : r;
    r = (p.x < 0.0f && p.y > 0.0f) ? P[5] : r;
    float dx = (sd_abs(p.x) - P[0]) + r, dy = (sd_abs(p.y) - P[1]) + r;
    float o = sd_len2(sd_max(dx, 0.0f), sd_max(dy, 0.0f));
    float u = sd_min(sd_max(dx, dy), 0.0f) - r;
    return o + u;
}
// sdf_triangle_2d C/sdf_2D.py:60-82 ; P = p0 p1 p2 (6) | e0 e1 e2 (6) | 1/dot(e_i,e_i) (3) | s
SDFK_DEV float prim_triangle2(V3 p, const float* __restrict__ P, const float* __restrict__) {
    float dmin = 3.0e38f, cmin = 3.0e38f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float vx = p.x - P[2 * i], vy = p.y - P[2 * i + 1];
        float ex = P[6 + 2 * i], ey = P[7 + 2 * i];
        float h = sd_clip01(sd_dot2(vx, vy, ex, ey) * P[12 + i]);
        float qx = sd_fma(-ex, h, vx), qy = sd_fma(-ey, h, vy);
        dmin = sd_min(dmin, sd_fma(qx, qx, qy * qy));
        cmin = sd_min(cmin, P[15] * sd_fma(vx, ey, -vy * ex));
    }
    return -sd_sqrt(dmin) * sd_sign(cmin);
}
// sdf_arc C/sdf_2D.py:85-103 ; P = (radius, cos mid, sin mid, |end - mid|)
SDFK_DEV float prim_arc2(V3 p, const float* __restrict__ P, const float* __restrict__) {
    float x, y;
    sd_rotmid(p.x, p.y, P[1], P[2], &x, &y);
    y = sd_abs(y);
    float psi = sd_clip(sd_atan2(y, x), 0.0f, P[3]);
    float s, c;
    sd_sincos(psi, &s, &c);
    return sd_len2(x - P[0] * c, y - P[0] * s);
}
// sdf_sector C/sdf_2D.py:105-129 ; P as sd_sector_tail
SDFK_DEV float prim_sector(V3 p, const float* __restrict__ P, const float* __restrict__) {
    float x, y;
    sd_rotmid(p.x, p.y, P[1], P[2], &x, &y);
    return sd_sector_tail(x, sd_abs(y), P);
}
// sdf_inf_sector C/sdf_2D.py:132-150 ; P = (cos mid, sin mid, half width, cos hw, sin hw)
SDFK_DEV float prim_infsector(V3 p, const float* __restrict__ P, const float* __restrict__) {
    float x, y;
    sd_rotmid(p.x, p.y, P[0], P[1], &x, &y);
    y = sd_abs(y);
    float phi = sd_atan2(y, x);
    float t = sd_max(sd_dot2(x, y, P[3], P[4]), 0.0f);
    float m = sd_len2(sd_fma(-P[3], t, x), sd_fma(-P[4], t, y));
    return sd_sign(phi - P[2]) * m;
}
// sdf_ngon C/sdf_2D.py:153-177 ; P = (radius, alpha, 1/alpha, t0 = -cos β, t1 = sin β, n0 = sin β, n1 = cos β, l)
SDFK_DEV float prim_ngon(V3 p, const float* __restrict__ P, const float* __restrict__) {
    float qx, qy;
    if (P[10] > 0.0f) {
        // integer n <= 16: the field is mirror-symmetric in y, so fold |y| into the first sector by at most n/2
        // rotations of -alpha (P[8], P[9] = cos, sin alpha) — no atan2, no sincos, same folded point up to rounding
        float x = p.x, y = sd_abs(p.y);
        const int nf = (int)P[10];
        for (int j = 0; j < nf; ++j) {
            const float yr = sd_fma(P[8], y, -P[9] * x), xr = sd_fma(P[8], x, P[9] * y);
            const bool over = yr >= 0.0f;                      // the angle is still >= alpha
            x = over ? xr : x;
            y = over ? yr : y;
        }
        qx = x - P[0];
        qy = y;
    } else {
        float phi = sd_atan2(p.y, p.x);
        phi = (phi < 0.0f) ? SDFK_TWO_PI + phi : phi;
        phi = sd_mod(phi, P[1], P[2]);
        float r = sd_len2(p.x, p.y);
        float s, c;
        sd_sincos(phi, &s, &c);
        qx = c * r - P[0];
        qy = s * r;
    }
    float h = sd_clip(sd_dot2(qx, qy, P[3], P[4]), 0.0f, P[7]);
    float len = sd_len2(sd_fma(-P[3], h, qx), sd_fma(-P[4], h, qy));
    return len * sd_sign(sd_dot2(qx, qy, P[5], P[6]));
}
// sdf_segmented_line_2d / edge loop of sdf_polygon_2d C/sdf_2D.py:191-208 ; table rows a(2) ba(2) inv
SDFK_DEV float prim_segline2(V3 p, const float* __restrict__ P, const float* __restrict__ T) {
    int n = (int)P[0];
    const float* __restrict__ tab = T + (int)P[1];
    float best = 1.0e32f;
    for (int i = 0; i < n; ++i) best = sd_min(best, sd_seg2_sq(p.x, p.y, tab + 5 * i));
    return sd_sqrt(best);
}
// KDTree nearest-point distance in the plane C/sdf_2D.py:180-188,214-224 ; table rows xy
SDFK_DEV float prim_nearest2(V3 p, const float* __restrict__ P, const float* __restrict__ T) {
    int n = (int)P[0];
    const float* __restrict__ tab = T + (int)P[1];
    float best = 3.0e38f;
    for (int i = 0; i < n; ++i) {
        float dx = p.x - tab[2 * i], dy = p.y - tab[2 * i + 1];
        best = sd_min(best, sd_fma(dx, dx, dy * dy));
    }
    return sd_sqrt(best);
}
// interior_polygon C/triangulation_functions.py:355-430 : -1 inside any convex piece, +1 outside.
// P = (piece count, table offset) ; table: per piece  K, then K rows (px, py, nx, ny).
// piece count -1: ONE piece, and the value is interior_convex's own (:355-387): max over the edges of
// sign(dot(p - p_k, n_k)), i.e. -1 inside, 0 on an edge line, +1 outside.
SDFK_DEV float prim_polysign(V3 p, const float* __restrict__ P, const float* __restrict__ T) {
    const bool raw = P[0] < 0.0f;
    int np_ = raw ? 1 : (int)P[0];
    const float* __restrict__ tab = T + (int)P[1];
    float interior = 1.0f;
    for (int j = 0; j < np_; ++j) {
        int k = (int)tab[0];
        float side = -1.0f;
        for (int i = 0; i < k; ++i) {
            const float* __restrict__ h = tab + 1 + 4 * i;
            side = sd_max(side, sd_sign(sd_dot2(p.x - h[0], p.y - h[1], h[2], h[3])));
        }
        if (raw) return side;
        interior = (side <= 0.0f) ? -1.0f : interior;
        tab += 1 + 4 * k;
    }
    return interior;
}
// ParametricCurve.shape() C/geom_2d.py:432-452: product over the outline's edges whose x-range
// [lx, ux) contains the point of sign(dot(p - p_i, n_i)). P = (edge count, table offset);
// table rows (px, py, nx, ny, lx, ux)
SDFK_DEV float prim_shapesign(V3 p, const float* __restrict__ P, const float* __restrict__ T) {
    int n = (int)P[0];
    const float* __restrict__ tab = T + (int)P[1];
    float interior = 1.0f;
    for (int i = 0; i < n; ++i) {
        const float* __restrict__ h = tab + 6 * i;
        float s = sd_sign(sd_dot2(p.x - h[0], p.y - h[1], h[2], h[3]));
        interior = (p.x >= h[4] && p.x < h[5]) ? interior * s : interior;
    }
    return interior;
}
// |z| - h/2 of extrusion C/modifications.py:493 ; P = h/2
template <typename T> SDFK_DEV T prim_zslab(V3T<T> p, const float* __restrict__ P, const float* __restrict__) { return sd_abs(p.z) - P[0]; }

// =============================================================================================
// value -> value     signature: float f(float v, const float* P)
// =============================================================================================
template <typename T> SDFK_DEV T val_scale(T v, const float* __restrict__ P) { return P[0] * v; }
template <typename T> SDFK_DEV T val_subc(T v, const float* __restrict__ P) { return v - P[0]; }
// rounding_cs C/modifications.py:141 : scale*f - r   (two roundings, as the reference)
template <typename T> SDFK_DEV T val_affine(T v, const float* __restrict__ P) { return P[0] * v - P[1]; }
template <typename T> SDFK_DEV T val_abs(T v, const float* __restrict__) { return sd_abs(v); }
template <typename T> SDFK_DEV T val_neg(T v, const float* __restrict__) { return -v; }
template <typename T> SDFK_DEV T val_sign(T v, const float* __restrict__) { return sd_sign(v); }
template <typename T> SDFK_DEV T val_onion(T v, const float* __restrict__ P) { return sd_abs(v) - P[0]; }
template <typename T> SDFK_DEV T val_concentric(T v, const float* __restrict__ P) { return sd_abs(v - P[0]); }
// sigmoid_falloff / positive_sigmoid_falloff C/post_processing.py:380-412 ; P = (A, 4/w, shift)
SDFK_DEV float val_sigmoid(float v, const float* __restrict__ P) {
    float e = expf((v - P[2]) * P[1]);
    return P[0] * (1.0f / (1.0f + e));
}
// capped_exponential :415-429 ; P = (A, -4/w)
SDFK_DEV float val_capexp(float v, const float* __restrict__ P) { return P[0] * sd_min(expf(v * P[1]), 1.0f); }
// hard_binarization :432-446
SDFK_DEV float val_hardbin(float v, const float* __restrict__ P) { return (v <= P[0]) ? 1.0f : 0.0f; }
// linear_falloff :449-463 ; P = (A, 1/w)
template <typename T> SDFK_DEV T val_linfall(T v, const float* __restrict__ P) { return sd_clip01(1.0f - v * P[1]) * P[0]; }
// relu :466-477 ; P = 1/w
template <typename T> SDFK_DEV T val_relu(T v, const float* __restrict__ P) { return sd_max0(v * P[0]); }
// smooth_relu :480-500 ; P = (1/w, b)
template <typename T> SDFK_DEV T val_smoothrelu(T v, const float* __restrict__ P) {
    T u = v * P[0];
    return (u + sd_sqrt(sd_fma(u, u, sp<T>(P[1])))) * 0.5f;
}
// slowstart :503-523 ; P = (1/w, b/w, sqrt(b/w)*ground)
template <typename T> SDFK_DEV T val_slowstart(T v, const float* __restrict__ P) {
    T u = sd_max0(v * P[0]);
    return sd_sqrt(sd_fma(u, u, sp<T>(P[1]))) - P[2];
}
// gaussian_boundary / gaussian_falloff :526-558 ; P = (A, 1/w, clamp_at_zero)
SDFK_DEV float val_gauss(float v, const float* __restrict__ P) {
    float u = (P[2] != 0.0f) ? sd_max(v, 0.0f) : v;
    u = u * P[1];
    return P[0] * expf(-4.0f * (u * u));
}

// sign() / hard_binarization(. , 0) of a value map that is strictly positive in exact arithmetic — A exp(-4 (v/w)^2),
// A min(exp(-4 v / w), 1), A / (1 + exp(4 (v - shift) / w)) with A > 0 (C/post_processing.py:380-429, 526-558). The
// reference computes the map in float64, where exp reaches zero at an exponent of -745.13 (and 1 + exp(x) infinity at
// x = 709.78); in fp32 the map is zero from -103 on, so sign(map) would be 0 where the reference still has 1 (|v / w|
// between 5.1 and 13.6 for the Gaussian). The lowering (aegolius_amd/_lower.py, Lowerer.emit) therefore replaces the
// pair by this operator, which compares the EXPONENT with float64's limit.
// P = (kind: 0 Gaussian, 1 capped exponential, 2 sigmoid; scale; clamp flag | shift; limit; result while non-zero; result at zero)
SDFK_DEV float val_expflag(float v, const float* __restrict__ P) {
    float t;
    if (P[0] == 0.0f) {                      // (wave-uniform: parameters)
        float u = (P[2] != 0.0f) ? sd_max(v, 0.0f) : v;
        u = u * P[1];
        t = -4.0f * (u * u);
    } else if (P[0] == 1.0f) {
        t = v * P[1];
    } else {
        t = -((v - P[2]) * P[1]);
    }
    return (t >= P[3]) ? P[4] : P[5];
}

// =============================================================================================
// (value, value) -> value     signature: float f(float a, float b, const float* P)
// =============================================================================================
template <typename T> SDFK_DEV T cmb_mul(T a, T b, const float* __restrict__) { return a * b; }
template <typename T> SDFK_DEV T cmb_add(T a, T b, const float* __restrict__) { return a + b; }
template <typename T> SDFK_DEV T cmb_diff(T a, T b, const float* __restrict__) { return a - b; }
template <typename T> SDFK_DEV T cmb_min(T a, T b, const float* __restrict__) { return sd_min(a, b); }
template <typename T> SDFK_DEV T cmb_max(T a, T b, const float* __restrict__) { return sd_max(a, b); }
template <typename T> SDFK_DEV T cmb_subtract(T a, T b, const float* __restrict__) { return sd_max(a, -b); }
// smoothmin_poly2 C/combine.py:12-18 : min(a,b) - h^2 w/4, h = max(w - |a-b|, 0)/w  ==  min - t^2/(4w)
// with t = max(w - |a-b|, 0). P = (w, 1/(4w)). The reference's `w == 0 -> plain min` case is resolved at
// lowering time (VMIN is emitted instead), so the kernel stays branch-free.
template <typename T> SDFK_DEV T cmb_smin2(T a, T b, const float* __restrict__ P) {
    T t = sd_max0(P[0] - sd_abs(a - b));
    return sd_fma(-(t * t), sp<T>(P[1]), sd_min(a, b));
}
// smoothmin_poly3 C/combine.py:20-26 : min(a,b) - h^3 w/6  ==  min - t^3/(6 w^2). P = (w, 1/(6 w^2))
template <typename T> SDFK_DEV T cmb_smin3(T a, T b, const float* __restrict__ P) {
    T t = sd_max0(P[0] - sd_abs(a - b));
    return sd_fma(-(t * t * t), sp<T>(P[1]), sd_min(a, b));
}
template <typename T> SDFK_DEV T cmb_smax3(T a, T b, const float* __restrict__ P) { return -cmb_smin3(-a, -b, P); }
template <typename T> SDFK_DEV T cmb_ssub3(T a, T b, const float* __restrict__ P) { return -cmb_smin3(-a, b, P); }
// smoothmax_boltz C/combine.py:29-34 ; P = 1/w. Both exponentials are shifted by max(a,b)/w, which
// cancels between numerator and denominator (identical in exact arithmetic, no fp32 overflow).
SDFK_DEV float cmb_boltz(float a, float b, const float* __restrict__ P) {
    float xa = a * P[0], xb = b * P[0];
    float m = sd_max(xa, xb);
    float ea = expf(xa - m), eb = expf(xb - m);
    return sd_fma(a, ea, b * eb) / (ea + eb);
}
SDFK_DEV float cmb_boltzsub(float a, float b, const float* __restrict__ P) { return cmb_boltz(a, -b, P); }
// extrusion tail C/modifications.py:494-496 ; a = d(x, y, 0), b = |z| - h/2
template <typename T> SDFK_DEV T cmb_extrude(T a, T b, const float* __restrict__) {
    return sd_min0(sd_max(a, b)) + sd_len2(sd_max0(a), sd_max0(b));
}

// V_FIELD: the value of an auxiliary per-point field (the output of an earlier evaluation stage that went through a
// grid-neighbourhood operator). The kernels read it themselves (they know the point index): sdfk_aux below; this
// placeholder only keeps the opcode table uniform.
template <typename T> SDFK_DEV T prim_field(V3T<T>, const float* __restrict__, const float* __restrict__) { return sp<T>(0.0f); }
// AUX points at this lane's first point in auxiliary row 0, rows are AUXS elements apart
template <typename T> SDFK_DEV T sdfk_aux(const float* __restrict__ AUX, long long AUXS, int k);
template <> SDFK_DEV float sdfk_aux<float>(const float* __restrict__ AUX, long long AUXS, int k) { return AUX[k * AUXS]; }
template <> SDFK_DEV f2 sdfk_aux<f2>(const float* __restrict__ AUX, long long AUXS, int k) {
    f2 r = {AUX[k * AUXS], AUX[k * AUXS + 1]};
    return r;
}

// =============================================================================================
// pair adapters: operators written for one point per lane, applied to each half of an f2 lane
// =============================================================================================
#define SDFK_PAIR_C_C(F)                                                                                   \
    SDFK_DEV V3P F(V3P p, const float* __restrict__ P, const float* __restrict__ T, int imm) {             \
        return sd_join(F(sd_lo(p), P, T, imm), F(sd_hi(p), P, T, imm));                                    \
    }
#define SDFK_PAIR_V_C(F)                                                                                   \
    SDFK_DEV f2 F(V3P p, const float* __restrict__ P, const float* __restrict__ T) {                       \
        f2 r = {F(sd_lo(p), P, T), F(sd_hi(p), P, T)};                                                     \
        return r;                                                                                          \
    }
#define SDFK_PAIR_V_V(F)                                                                                   \
    SDFK_DEV f2 F(f2 v, const float* __restrict__ P) {                                                     \
        f2 r = {F(v.x, P), F(v.y, P)};                                                                     \
        return r;                                                                                          \
    }
#define SDFK_PAIR_V_VV(F)                                                                                  \
    SDFK_DEV f2 F(f2 a, f2 b, const float* __restrict__ P) {                                               \
        f2 r = {F(a.x, b.x, P), F(a.y, b.y, P)};                                                           \
        return r;                                                                                          \
    }
SDFK_PAIR_C_C(op_twist) SDFK_PAIR_C_C(op_bend) SDFK_PAIR_C_C(op_infrep) SDFK_PAIR_C_C(op_finrep)
SDFK_PAIR_C_C(op_rotsym) SDFK_PAIR_C_C(op_lininst) SDFK_PAIR_C_C(op_curveinst) SDFK_PAIR_C_C(op_curveinstt)
SDFK_PAIR_V_C(prim_braid) SDFK_PAIR_V_C(prim_arc3d) SDFK_PAIR_V_C(prim_segment3) SDFK_PAIR_V_C(prim_infcone)
SDFK_PAIR_V_C(prim_solidangle) SDFK_PAIR_V_C(prim_triangle3) SDFK_PAIR_V_C(prim_quad3) SDFK_PAIR_V_C(prim_segline3)
SDFK_PAIR_V_C(prim_nearest3) SDFK_PAIR_V_C(prim_neucircle) SDFK_PAIR_V_C(prim_segment2) SDFK_PAIR_V_C(prim_rbox2)
SDFK_PAIR_V_C(prim_triangle2) SDFK_PAIR_V_C(prim_arc2) SDFK_PAIR_V_C(prim_sector) SDFK_PAIR_V_C(prim_infsector)
SDFK_PAIR_V_C(prim_ngon) SDFK_PAIR_V_C(prim_segline2) SDFK_PAIR_V_C(prim_nearest2) SDFK_PAIR_V_C(prim_polysign)
SDFK_PAIR_V_C(prim_shapesign) SDFK_PAIR_V_C(prim_neartree)
SDFK_PAIR_V_V(val_sigmoid) SDFK_PAIR_V_V(val_capexp) SDFK_PAIR_V_V(val_hardbin) SDFK_PAIR_V_V(val_gauss) SDFK_PAIR_V_V(val_expflag)
SDFK_PAIR_V_VV(cmb_boltz) SDFK_PAIR_V_VV(cmb_boltzsub)

#endif  // SDFK_DEVICE_H

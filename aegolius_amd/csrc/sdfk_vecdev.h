// sdfk_vecdev.h — device helpers of the field consumers and the vector-field kernels, shared verbatim by
//   * the kernels compiled by hipcc into libsdfk.so (sdfk_fieldops.inc, sdfk_vector.inc) and
//   * every chain-specialised vector kernel (the text of this file is embedded in libsdfk.so and handed to hiprtc
//     behind sdfk_device.h), so the interpreter and the specialised kernels execute the same fp32 sequences.
#ifndef SDFK_VECDEV_H
#define SDFK_VECDEV_H

typedef float sdfk_f4 __attribute__((ext_vector_type(4)));
typedef float sdfk_f4u __attribute__((ext_vector_type(4), aligned(4)));

// (the library is built with -fno-honor-nans for the evaluation kernels: zero / infinite / NaN vectors are
// classified on the bit patterns, which no floating-point assumption can rewrite)
// x, y, z: differences; hx, hy, hz: 1 where the component still owes its factor 1/2. The halving is folded into the
// power-of-two scaling, so a subnormal difference keeps its last bit (0.5 * d would round it away).
SDFK_DEV void sdfk_unit3(float& x, float& y, float& z, int hx, int hy, int hz) {
    const unsigned ax = __float_as_uint(x) & 0x7fffffffu, ay = __float_as_uint(y) & 0x7fffffffu,
                   az = __float_as_uint(z) & 0x7fffffffu;
    const unsigned am = max(ax, max(ay, az));                 // bits of the largest |difference|; NaN sorts above inf
    if (am == 0u) return;                                     // zero vector: left as it is
    const float mx = __uint_as_float(am);
    if (am >= 0x7f800000u) {                                  // an infinite or NaN component: plain v / |v| (NaN, or 0)
        x /= mx;
        y /= mx;
        z /= mx;
        return;
    }
    const int e = __builtin_amdgcn_frexp_expf(mx);            // mx = m * 2^e, m in [0.5, 1)
    const float a = __builtin_amdgcn_ldexpf(x, -e - hx), b = __builtin_amdgcn_ldexpf(y, -e - hy),
                c = __builtin_amdgcn_ldexpf(z, -e - hz);      // largest in [0.25, 1)
    const float r = __builtin_amdgcn_rsqf(fmaf(a, a, fmaf(b, b, c * c)));
    x = a * r;
    y = b * r;
    z = c * r;
}


// ---- vector-field programs (sdfk_vector.inc) ----------------------------------------------------------------------
enum {
    VOP_INIT_P = 0,        // cartesian_define: v = p                                  C/vector_functions.py:15-20
    VOP_INIT_SPHERICAL,    // (r, phi, theta) = p                                       :23-32
    VOP_INIT_CYLINDRICAL,  // (r, phi, z) = p                                           :35-43
    VOP_INIT_RADIAL_SPH,   // p / |p|                                                   :46-48
    VOP_INIT_RADIAL_CYL,   // (x, y, 0) / |(x, y)|                                      :51-55
    VOP_INIT_VORTEX,       // radial_cyl turned by 90 degrees                           :71-79
    VOP_INIT_AAR,          // radial_cyl turned by A                                    :82-94
    VOP_INIT_AAV,          // vortex turned by A                                        :97-109
    VOP_INIT_CONST,        // v = A (x / y / z_vector_field)                            :112-127
    VOP_INIT_STREAM,       // v = A (an already evaluated field: user callables, from_sdf)
    VOP_ADD,               // add_vectors       C/vector_modification_functions.py:23-28
    VOP_SUB,               // subtract_vectors  :31-36
    VOP_MUL,               // rescale_vectors   :39-41
    VOP_ROT_Z,             // rotate_vectors_phi / rotate_vectors_z_axis  :44-52, :95-103
    VOP_ROT_X,             // :71-80
    VOP_ROT_Y,             // :83-92
    VOP_ROT_THETA,         // :55-68
    VOP_ROT_AXIS,          // A = axis, B = angle  :106-119
    VOP_REVOLVE_X,         // A = coordinates      :122-133
    VOP_REVOLVE_Y,         // :136-146
    VOP_REVOLVE_Z,         // :149-159
    VOP_NORMALIZE,         // batch_normalize      :14-20
    VOP_COUNT
};
enum { VK_NONE = 0, VK_IMM1, VK_IMM3, VK_ROW1, VK_ROW3, VK_P };   // operand kinds: number, 3-vector, (N,) row, (3, N) rows, p itself

struct W3 {
    float x, y, z;
};

SDFK_DEV void sdfk_turn(float& a, float& b, float angle) {   // (a, b) <- R(angle) (a, b)
    float s, c;
    sd_sincos(angle, &s, &c);
    const float t = a * c - b * s;
    b = a * s + b * c;
    a = t;
}
// cos and sin of atan2(y, x) without the angle (atan2(0, 0) = 0)
SDFK_DEV void sdfk_dir2(float x, float y, float* c, float* s) {
    float ux = x, uy = y, uz = 0.0f;
    sdfk_unit3(ux, uy, uz, 0, 0, 0);
    const bool zero = (__float_as_uint(x) & 0x7fffffffu) == 0u && (__float_as_uint(y) & 0x7fffffffu) == 0u;
    *c = zero ? (__float_as_uint(x) >> 31 ? -1.0f : 1.0f) : ux;     // atan2(+-0, -0) = +-pi: cos = -1
    *s = zero ? 0.0f : uy;
}

// one instruction on one point: v <- op(v; p, A, B)
SDFK_DEV W3 sdfk_vec_apply(int op, W3 v, W3 p, W3 A, W3 B) {
    {
        switch (op) {
        case VOP_INIT_P:
            v = p;
            break;
        case VOP_INIT_SPHERICAL: {
            float sp, cp, st, ct;
            sd_sincos(p.y, &sp, &cp);
            sd_sincos(p.z, &st, &ct);
            v = W3{p.x * cp * st, p.x * sp * st, p.x * ct};
            break;
        }
        case VOP_INIT_CYLINDRICAL: {
            float sp, cp;
            sd_sincos(p.y, &sp, &cp);
            v = W3{p.x * cp, p.x * sp, p.z};
            break;
        }
        case VOP_INIT_RADIAL_SPH:
            v = p;
            sdfk_unit3(v.x, v.y, v.z, 0, 0, 0);
            break;
        case VOP_INIT_RADIAL_CYL:
        case VOP_INIT_VORTEX:
        case VOP_INIT_AAR:
        case VOP_INIT_AAV: {
            v = W3{p.x, p.y, 0.0f};
            sdfk_unit3(v.x, v.y, v.z, 0, 0, 0);
            if (op == VOP_INIT_VORTEX) v = W3{-v.y, v.x, 0.0f};
            if (op == VOP_INIT_AAR) sdfk_turn(v.x, v.y, A.x);
            if (op == VOP_INIT_AAV) {                          // (-x sa - y ca, x ca - y sa): the vortex, turned
                float s, c;
                sd_sincos(A.x, &s, &c);
                v = W3{-v.x * s - v.y * c, v.x * c - v.y * s, 0.0f};
            }
            break;
        }
        case VOP_INIT_CONST:
        case VOP_INIT_STREAM:
            v = A;
            break;
        case VOP_ADD:
            v = W3{v.x + A.x, v.y + A.y, v.z + A.z};
            break;
        case VOP_SUB:
            v = W3{v.x - A.x, v.y - A.y, v.z - A.z};
            break;
        case VOP_MUL:
            v = W3{v.x * A.x, v.y * A.y, v.z * A.z};
            break;
        case VOP_ROT_Z:
            sdfk_turn(v.x, v.y, A.x);
            break;
        case VOP_ROT_X:
            sdfk_turn(v.y, v.z, A.x);
            break;
        case VOP_ROT_Y: {                                      // (x ca - z sa, y, x sa + z ca)
            sdfk_turn(v.x, v.z, A.x);
            break;
        }
        case VOP_ROT_THETA: {
            W3 r = W3{v.x, v.y, 0.0f};
            sdfk_unit3(r.x, r.y, r.z, 0, 0, 0);
            float s, c;
            sd_sincos(A.x, &s, &c);
            const W3 t = W3{r.x * v.z, r.y * v.z, -r.x * v.x - r.y * v.y};
            v = W3{v.x * c + t.x * s, v.y * c + t.y * s, v.z * c + t.z * s};
            break;
        }
        case VOP_ROT_AXIS: {                                   // v ca + sa (a x v) + (1 - ca) a (a . v); a as given
            float s, c;
            sd_sincos(B.x, &s, &c);
            const W3 cr = W3{A.y * v.z - A.z * v.y, A.z * v.x - A.x * v.z, A.x * v.y - A.y * v.x};
            const float d = (1.0f - c) * (A.x * v.x + A.y * v.y + A.z * v.z);
            v = W3{v.x * c + s * cr.x + d * A.x, v.y * c + s * cr.y + d * A.y, v.z * c + s * cr.z + d * A.z};
            break;
        }
        case VOP_REVOLVE_X:
        case VOP_REVOLVE_Y:
        case VOP_REVOLVE_Z: {
            float c, s;
            if (op == VOP_REVOLVE_X) {
                sdfk_dir2(A.y, A.z, &c, &s);                   // alpha = atan2(r.z, r.y)
                const float t = v.y * c - v.z * s;
                v.z = v.y * s + v.z * c;
                v.y = t;
            } else if (op == VOP_REVOLVE_Y) {
                sdfk_dir2(A.x, A.z, &c, &s);                   // alpha = atan2(r.z, r.x)
                const float t = v.x * c - v.z * s;
                v.z = v.x * s + v.z * c;
                v.x = t;
            } else {
                sdfk_dir2(A.x, A.y, &c, &s);                   // alpha = atan2(r.y, r.x)
                const float t = v.x * c - v.y * s;
                v.y = v.x * s + v.y * c;
                v.x = t;
            }
            break;
        }
        case VOP_NORMALIZE:
            sdfk_unit3(v.x, v.y, v.z, 0, 0, 0);
            break;
        default:
            break;
        }
    }
    return v;
}


// read-outs of C/geom.py:256-362: 1 x | 2 y | 3 z | 4 phi = atan2(y, x) | 5 theta = acos(z) | 6 length
SDFK_DEV float sdfk_vec_readout(int out_kind, W3 vv) {
    if (out_kind <= 3) return out_kind == 1 ? vv.x : (out_kind == 2 ? vv.y : vv.z);
    if (out_kind == 4) return sd_atan2(vv.y, vv.x);
    if (out_kind == 5) return acosf(vv.z);
    float x = vv.x, y = vv.y, z = vv.z;                        // |v| with the scaling of the normalisation
    const unsigned am = max(__float_as_uint(x) & 0x7fffffffu, max(__float_as_uint(y) & 0x7fffffffu,
                                                                  __float_as_uint(z) & 0x7fffffffu));
    if (am == 0u) return 0.0f;
    if (am >= 0x7f800000u) return sqrtf(x * x + y * y + z * z);
    const int ex = __builtin_amdgcn_frexp_expf(__uint_as_float(am));
    x = __builtin_amdgcn_ldexpf(x, -ex);
    y = __builtin_amdgcn_ldexpf(y, -ex);
    z = __builtin_amdgcn_ldexpf(z, -ex);
    return __builtin_amdgcn_ldexpf(sqrtf(fmaf(x, x, fmaf(y, y, z * z))), ex);
}

#endif

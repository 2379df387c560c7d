// sdfk_codegen.cpp — program -> HIP source for hiprtc (see sdfk_codegen.h).
#include "sdfk_codegen.h"

#include <cstdio>
#include <set>

// text of sdfk_device.h / sdfk_access.h, generated at build time by __graft_entry__.build()
#include "sdfk_embedded.inc"

static const char kWrappers[] = R"SDFKW(
template <int VEC, typename SRC>
static __device__ __forceinline__ void sdfk_body(const float* __restrict__ PRM, const float* __restrict__ TAB,
                                                 const SRC& src, long long off, long long n,
                                                 float* __restrict__ out) {
    const long long block_base = (long long)blockIdx.x * (SDFK_BLOCK * VEC);
    const unsigned lane_off = threadIdx.x * VEC;
    if (block_base + lane_off >= n) return;
    V3 p[VEC];
    sdfk_load<VEC>(src, off + block_base, lane_off, p);
    float v[VEC];
    if constexpr (VEC == 4) {
        // two points per lane value: packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32)
        const f2 ra = sdfk_point<f2>(sd_join(p[0], p[1]), PRM, TAB);
        const f2 rb = sdfk_point<f2>(sd_join(p[2], p[3]), PRM, TAB);
        v[0] = ra.x; v[1] = ra.y; v[2] = rb.x; v[3] = rb.y;
    } else {
#pragma unroll
        for (int k = 0; k < VEC; ++k) v[k] = sdfk_point<float>(p[k], PRM, TAB);
    }
    sdfk_store<VEC>(out, off + block_base + lane_off, v);
}
extern "C" __global__ __launch_bounds__(SDFK_BLOCK) void sdfk_spec_v4(
    const float* __restrict__ PRM, const float* __restrict__ TAB, const float* __restrict__ co, long long stride,
    long long off, long long n, float* __restrict__ out) {
    SrcArray s = {co, stride};
    sdfk_body<4>(PRM, TAB, s, off, n, out);
}
extern "C" __global__ __launch_bounds__(SDFK_BLOCK) void sdfk_spec_v1(
    const float* __restrict__ PRM, const float* __restrict__ TAB, const float* __restrict__ co, long long stride,
    long long off, long long n, float* __restrict__ out) {
    SrcArray s = {co, stride};
    sdfk_body<1>(PRM, TAB, s, off, n, out);
}
extern "C" __global__ __launch_bounds__(SDFK_BLOCK) void sdfk_spec_g4(
    const float* __restrict__ PRM, const float* __restrict__ TAB, SrcGrid g, long long off, long long n,
    float* __restrict__ out) {
    sdfk_body<4>(PRM, TAB, g, off, n, out);
}
extern "C" __global__ __launch_bounds__(SDFK_BLOCK) void sdfk_spec_g1(
    const float* __restrict__ PRM, const float* __restrict__ TAB, SrcGrid g, long long off, long long n,
    float* __restrict__ out) {
    sdfk_body<1>(PRM, TAB, g, off, n, out);
}
)SDFKW";

std::string sdfk_generate_source(const sdfk_opinfo* ops, int n_ops, const uint32_t* code, size_t n_instr,
                                 int result_reg) {
    std::string s;
    s.reserve(sizeof(kEmbeddedDevice) + sizeof(kEmbeddedAccess) + sizeof(kWrappers) + 96 * n_instr + 512);
    s += kEmbeddedDevice;
    s += "\n";
    s += kEmbeddedAccess;
    s += "\ntemplate <typename T> static __device__ __forceinline__ T sdfk_point(V3T<T> C0, "
         "const float* __restrict__ PRM, const float* __restrict__ TAB) {\n";
    std::set<unsigned> cregs, vregs;
    for (size_t i = 0; i < n_instr; ++i) {
        const uint32_t w = code[2 * i];
        const unsigned op = w & 255u, a = (w >> 8) & 255u;
        if ((int)op >= n_ops) continue;
        if (ops[op].kind == SDFK_KIND_C_C) cregs.insert(a);
        else vregs.insert(a);
    }
    char buf[256];
    for (unsigned c : cregs)
        if (c != 0) {
            snprintf(buf, sizeof buf, "    V3T<T> C%u;\n", c);
            s += buf;
        }
    for (unsigned v : vregs) {
        snprintf(buf, sizeof buf, "    T V%u;\n", v);
        s += buf;
    }
    for (size_t i = 0; i < n_instr; ++i) {
        const uint32_t w = code[2 * i], poff = code[2 * i + 1];
        const unsigned op = w & 255u, a = (w >> 8) & 255u, b = (w >> 16) & 255u, c = w >> 24;
        if ((int)op >= n_ops) continue;
        const sdfk_opinfo& o = ops[op];
        switch (o.kind) {
            case SDFK_KIND_C_C:
                snprintf(buf, sizeof buf, "    C%u = %s(C%u, PRM + %u, TAB, %u);\n", a, o.func, b, poff, c);
                break;
            case SDFK_KIND_V_C:
                snprintf(buf, sizeof buf, "    V%u = %s(C%u, PRM + %u, TAB);\n", a, o.func, b, poff);
                break;
            case SDFK_KIND_V_V:
                snprintf(buf, sizeof buf, "    V%u = %s(V%u, PRM + %u);\n", a, o.func, b, poff);
                break;
            default:
                snprintf(buf, sizeof buf, "    V%u = %s(V%u, V%u, PRM + %u);\n", a, o.func, b, c, poff);
                break;
        }
        s += buf;
    }
    snprintf(buf, sizeof buf, "    return V%d;\n}\n", result_reg);
    s += buf;
    s += kWrappers;
    return s;
}

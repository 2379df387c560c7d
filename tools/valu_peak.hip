// Developer micro-benchmark: VALU issue cost of plain vs packed fp32 forms on gfx950, in shader
// cycles (s_memtime) per wave64 instruction per SIMD, at 8 waves per SIMD. Calibrates the compute
// roof used in DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f2 __attribute__((ext_vector_type(2)));
#define ITER 16384
#define KEEP8(a,b,c,d,e,f,g,h) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h))
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, const float* __restrict__ prm) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x0}, p5 = {x3, x2}, p6 = {x5, x4}, p7 = {x7, x6};
    const float a = prm[0], b = prm[1];            // SGPRs
    f2 av = {x0 * 0 + a, x0 * 0 + a}, bv = {x0 * 0 + b, x0 * 0 + b};  // VGPR copies
    f2 as = {a, a}, bs = {b, b};                   // SGPR splats
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITER; ++i) {
        if (KIND == 0) {  // v_fma_f32, SGPR multiplier
            x0 = __builtin_fmaf(x0, a, x1); x1 = __builtin_fmaf(x1, a, x2); x2 = __builtin_fmaf(x2, a, x3); x3 = __builtin_fmaf(x3, a, x4);
            x4 = __builtin_fmaf(x4, a, x5); x5 = __builtin_fmaf(x5, a, x6); x6 = __builtin_fmaf(x6, a, x7); x7 = __builtin_fmaf(x7, a, x0);
            KEEP8(x0, x1, x2, x3, x4, x5, x6, x7);
        } else if (KIND == 1) {  // v_pk_fma_f32, all VGPR
            p0 = __builtin_elementwise_fma(p0, av, p1); p1 = __builtin_elementwise_fma(p1, av, p2);
            p2 = __builtin_elementwise_fma(p2, av, p3); p3 = __builtin_elementwise_fma(p3, av, p4);
            p4 = __builtin_elementwise_fma(p4, av, p5); p5 = __builtin_elementwise_fma(p5, av, p6);
            p6 = __builtin_elementwise_fma(p6, av, p7); p7 = __builtin_elementwise_fma(p7, av, p0);
            KEEP8(p0, p1, p2, p3, p4, p5, p6, p7);
        } else if (KIND == 2) {  // v_pk_fma_f32, SGPR splat multiplier (op_sel broadcast)
            p0 = __builtin_elementwise_fma(p0, as, p1); p1 = __builtin_elementwise_fma(p1, as, p2);
            p2 = __builtin_elementwise_fma(p2, as, p3); p3 = __builtin_elementwise_fma(p3, as, p4);
            p4 = __builtin_elementwise_fma(p4, as, p5); p5 = __builtin_elementwise_fma(p5, as, p6);
            p6 = __builtin_elementwise_fma(p6, as, p7); p7 = __builtin_elementwise_fma(p7, as, p0);
            KEEP8(p0, p1, p2, p3, p4, p5, p6, p7);
        } else if (KIND == 3) {  // v_sqrt_f32
            x0 = __builtin_amdgcn_sqrtf(x0); x1 = __builtin_amdgcn_sqrtf(x1); x2 = __builtin_amdgcn_sqrtf(x2); x3 = __builtin_amdgcn_sqrtf(x3);
            x4 = __builtin_amdgcn_sqrtf(x4); x5 = __builtin_amdgcn_sqrtf(x5); x6 = __builtin_amdgcn_sqrtf(x6); x7 = __builtin_amdgcn_sqrtf(x7);
            KEEP8(x0, x1, x2, x3, x4, x5, x6, x7);
        } else if (KIND == 4) {  // v_add_f32 VOP2 with SGPR
            x0 += a; x1 += b; x2 += a; x3 += b; x4 += a; x5 += b; x6 += a; x7 += b;
            KEEP8(x0, x1, x2, x3, x4, x5, x6, x7);
        } else if (KIND == 5) {  // v_pk_mul_f32 VGPR
            p0 *= av; p1 *= bv; p2 *= av; p3 *= bv; p4 *= av; p5 *= bv; p6 *= av; p7 *= bv;
            KEEP8(p0, p1, p2, p3, p4, p5, p6, p7);
        } else if (KIND == 6) {  // v_pk_add_f32 with SGPR splat
            p0 += as; p1 += bs; p2 += as; p3 += bs; p4 += as; p5 += bs; p6 += as; p7 += bs;
            KEEP8(p0, p1, p2, p3, p4, p5, p6, p7);
        } else if (KIND == 7) {  // v_max_f32 with inline constant 0
            x0 = __builtin_fmaxf(x0, 0.f); x1 = __builtin_fmaxf(x1, 0.f); x2 = __builtin_fmaxf(x2, 0.f); x3 = __builtin_fmaxf(x3, 0.f);
            x4 = __builtin_fmaxf(x4, 0.f); x5 = __builtin_fmaxf(x5, 0.f); x6 = __builtin_fmaxf(x6, 0.f); x7 = __builtin_fmaxf(x7, 0.f);
            KEEP8(x0, x1, x2, x3, x4, x5, x6, x7);
        } else if (KIND == 8) {  // v_fma_f32 all VGPR
            x0 = __builtin_fmaf(x0, av.x, x1); x1 = __builtin_fmaf(x1, av.x, x2); x2 = __builtin_fmaf(x2, av.x, x3); x3 = __builtin_fmaf(x3, av.x, x4);
            x4 = __builtin_fmaf(x4, av.x, x5); x5 = __builtin_fmaf(x5, av.x, x6); x6 = __builtin_fmaf(x6, av.x, x7); x7 = __builtin_fmaf(x7, av.x, x0);
            KEEP8(x0, x1, x2, x3, x4, x5, x6, x7);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    float r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y +
              p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
    if (r == 12345.678f) out[0] = r;
}
template <int KIND>
void run(const char* name, float* d, unsigned long long* dc, const float* prm) {
    const int blocks = 256 * 8;  // 8 blocks of 4 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, dc, prm);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, dc, prm);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(blocks);
    (void)hipMemcpy(c.data(), dc, blocks * 8, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    double med = (double)c[blocks / 2];
    // one wave issues ITER*8 instructions while 8 waves share its SIMD: per-SIMD cost = cycles / (ITER*8*8)
    printf("%-34s %7.3f ms  median wave cycles %.0f -> %.2f cycles per wave-instr per SIMD (clock %.2f GHz)\n", name, ms,
           med, med / (ITER * 8.0 * 8.0), med / (ms * 1e6));
}
int main() {
    float* d; unsigned long long* dc; float* prm;
    (void)hipMalloc(&d, 4); (void)hipMalloc(&dc, 256 * 8 * 8); (void)hipMalloc(&prm, 8);
    float h[2] = {1.0001f, 0.9999f};
    (void)hipMemcpy(prm, h, 8, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("v_fma_f32 (sgpr mult)", d, dc, prm); run<8>("v_fma_f32 (vgpr)", d, dc, prm);
        run<1>("v_pk_fma_f32 (vgpr)", d, dc, prm); run<2>("v_pk_fma_f32 (sgpr splat)", d, dc, prm);
        run<4>("v_add_f32 (sgpr)", d, dc, prm); run<6>("v_pk_add_f32 (sgpr splat)", d, dc, prm);
        run<5>("v_pk_mul_f32 (vgpr)", d, dc, prm); run<7>("v_max_f32 (inline 0)", d, dc, prm);
        run<3>("v_sqrt_f32", d, dc, prm);
    }
    return 0;
}

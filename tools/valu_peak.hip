// Developer micro-benchmark: VALU issue rate of plain vs packed fp32 on gfx950 (per-SIMD cycles per
// wave64 instruction), to calibrate the compute roof used in DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define ITER 4096
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x0}, p5 = {x3, x2}, p6 = {x5, x4}, p7 = {x7, x6};
    f2 av = {a, a}, bv = {b, b};
    for (int i = 0; i < ITER; ++i) {
        if (KIND == 0) {  // v_fma_f32 x8
            x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
            x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
            asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        } else if (KIND == 1) {  // v_pk_fma_f32 x8
            p0 = __builtin_elementwise_fma(p0, av, bv); p1 = __builtin_elementwise_fma(p1, av, bv);
            p2 = __builtin_elementwise_fma(p2, av, bv); p3 = __builtin_elementwise_fma(p3, av, bv);
            p4 = __builtin_elementwise_fma(p4, av, bv); p5 = __builtin_elementwise_fma(p5, av, bv);
            p6 = __builtin_elementwise_fma(p6, av, bv); p7 = __builtin_elementwise_fma(p7, av, bv);
        } else if (KIND == 2) {  // v_max_f32 x8
            x0 = __builtin_fmaxf(x0, a) ; x1 = __builtin_fminf(x1, b); x2 = __builtin_fmaxf(x2, a); x3 = __builtin_fminf(x3, b);
            x4 = __builtin_fmaxf(x4, a) ; x5 = __builtin_fminf(x5, b); x6 = __builtin_fmaxf(x6, a); x7 = __builtin_fminf(x7, b);
            asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        } else if (KIND == 3) {  // v_sqrt_f32 x8
            x0 = __builtin_amdgcn_sqrtf(x0); x1 = __builtin_amdgcn_sqrtf(x1); x2 = __builtin_amdgcn_sqrtf(x2); x3 = __builtin_amdgcn_sqrtf(x3);
            x4 = __builtin_amdgcn_sqrtf(x4); x5 = __builtin_amdgcn_sqrtf(x5); x6 = __builtin_amdgcn_sqrtf(x6); x7 = __builtin_amdgcn_sqrtf(x7);
        } else if (KIND == 4) {  // v_add_f32 x8
            x0 += a; x1 += b; x2 += a; x3 += b; x4 += a; x5 += b; x6 += a; x7 += b;
            asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        } else if (KIND == 5) {  // v_pk_mul_f32 x8
            p0 *= av; p1 *= bv; p2 *= av; p3 *= bv; p4 *= av; p5 *= bv; p6 *= av; p7 *= bv;
            asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));
        } else if (KIND == 6) {  // v_mul_f32 x8
            x0 *= a; x1 *= b; x2 *= a; x3 *= b; x4 *= a; x5 *= b; x6 *= a; x7 *= b;
            asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        }
    }
    float r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y +
              p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
    if (r == 12345.678f) out[0] = r;
}
template <int KIND>
void run(const char* name, float* d) {
    const int blocks = 256 * 8;  // 8 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.9999f);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.9999f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double wave_instr = (double)blocks * 4 * ITER * 8;          // per launch
    double per_simd = wave_instr / 1024.0;                       // 256 CU * 4 SIMD
    printf("%-14s %.3f ms  -> %.2f ns per wave-instr per SIMD  (= %.2f cycles @2.4GHz)\n", name, ms,
           ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
}
int main() {
    float* d; hipMalloc(&d, 4);
    run<0>("v_fma_f32", d); run<1>("v_pk_fma_f32", d); run<2>("v_max/min_f32", d); run<3>("v_sqrt_f32", d);
    run<4>("v_add_f32", d); run<5>("v_pk_mul_f32", d); run<6>("v_mul_f32", d);
    return 0;
}

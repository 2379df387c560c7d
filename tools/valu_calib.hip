// Developer tool (GPU box): calibration of the VALU roof and of the SQ counters that DESIGN.md / tools/pmc_summarize.py
// read it from. Kernels with a KNOWN number of independent VALU wave-instructions and nothing else, at 1 / 2 / 4 / 8
// waves per SIMD, timed with HIP events; run it plain for the wall-clock rates and under
//   rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv
// for the counter readings of the same launches (kernel names carry KIND and waves per SIMD).
//   hipcc -O3 --offload-arch=gfx950 tools/valu_calib.hip -o tools/bin/valu_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
#define KEEP8(a, b, c, d, e, f, g, h) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h))
#define UNROLL 8            // 8 groups of 8 instructions per trip
// KIND: 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_sqrt_f32, 3 v_rcp_f32, 4 v_sin_f32, 5 v_max_f32, 6 v_cndmask (select), 7 mix 6 fma : 1 sqrt : 1 rcp
template <int KIND, int WPS>
__global__ __launch_bounds__(256) void calib(float* out, const float* __restrict__ prm, int trips) {
    float x0 = threadIdx.x * 1e-3f + 1.f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x0}, p5 = {x3, x2}, p6 = {x5, x4}, p7 = {x7, x6};
    const float a = prm[0], b = prm[1];
    f2 av = {x0 * 0 + a, x0 * 0 + a};
    const float sa = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, a)));
    const unsigned long long spair = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, a)) | (1ull << 62);
    for (int i = 0; i < trips; ++i) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (KIND == 0) {
                x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
                x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
                KEEP8(x0, x1, x2, x3, x4, x5, x6, x7);
            } else if (KIND == 1) {
                p0 = __builtin_elementwise_fma(p0, av, p1); p1 = __builtin_elementwise_fma(p1, av, p2);
                p2 = __builtin_elementwise_fma(p2, av, p3); p3 = __builtin_elementwise_fma(p3, av, p4);
                p4 = __builtin_elementwise_fma(p4, av, p5); p5 = __builtin_elementwise_fma(p5, av, p6);
                p6 = __builtin_elementwise_fma(p6, av, p7); p7 = __builtin_elementwise_fma(p7, av, p0);
                KEEP8(p0, p1, p2, p3, p4, p5, p6, p7);
            } else if (KIND == 2) {
                x0 = __builtin_amdgcn_sqrtf(x0); x1 = __builtin_amdgcn_sqrtf(x1); x2 = __builtin_amdgcn_sqrtf(x2); x3 = __builtin_amdgcn_sqrtf(x3);
                x4 = __builtin_amdgcn_sqrtf(x4); x5 = __builtin_amdgcn_sqrtf(x5); x6 = __builtin_amdgcn_sqrtf(x6); x7 = __builtin_amdgcn_sqrtf(x7);
                KEEP8(x0, x1, x2, x3, x4, x5, x6, x7);
            } else if (KIND == 3) {
                x0 = __builtin_amdgcn_rcpf(x0); x1 = __builtin_amdgcn_rcpf(x1); x2 = __builtin_amdgcn_rcpf(x2); x3 = __builtin_amdgcn_rcpf(x3);
                x4 = __builtin_amdgcn_rcpf(x4); x5 = __builtin_amdgcn_rcpf(x5); x6 = __builtin_amdgcn_rcpf(x6); x7 = __builtin_amdgcn_rcpf(x7);
                KEEP8(x0, x1, x2, x3, x4, x5, x6, x7);
            } else if (KIND == 4) {
                x0 = __builtin_amdgcn_sinf(x0); x1 = __builtin_amdgcn_sinf(x1); x2 = __builtin_amdgcn_sinf(x2); x3 = __builtin_amdgcn_sinf(x3);
                x4 = __builtin_amdgcn_sinf(x4); x5 = __builtin_amdgcn_sinf(x5); x6 = __builtin_amdgcn_sinf(x6); x7 = __builtin_amdgcn_sinf(x7);
                KEEP8(x0, x1, x2, x3, x4, x5, x6, x7);
            } else if (KIND == 5) {
                x0 = __builtin_fmaxf(x0, a); x1 = __builtin_fmaxf(x1, a); x2 = __builtin_fmaxf(x2, a); x3 = __builtin_fmaxf(x3, a);
                x4 = __builtin_fmaxf(x4, a); x5 = __builtin_fmaxf(x5, a); x6 = __builtin_fmaxf(x6, a); x7 = __builtin_fmaxf(x7, a);
                KEEP8(x0, x1, x2, x3, x4, x5, x6, x7);
            } else if (KIND == 6) {
                x0 = x0 > a ? x1 : b; x1 = x1 > a ? x2 : b; x2 = x2 > a ? x3 : b; x3 = x3 > a ? x4 : b;     // v_cmp + v_cndmask: 2 each
                KEEP8(x0, x1, x2, x3, x4, x5, x6, x7);
            } else if (KIND == 7) {
                x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
                x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_amdgcn_sqrtf(x6); x7 = __builtin_amdgcn_rcpf(x7);
                KEEP8(x0, x1, x2, x3, x4, x5, x6, x7);
            } else {
                // exact instruction forms (inline asm), 8 independent chains
#define A8(INS) asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(av.x), "v"(av.y))
#define I_ADD(k) "v_add_f32_e32 %" #k ", %8, %" #k "\n"
#define I_MUL(k) "v_mul_f32_e32 %" #k ", %8, %" #k "\n"
#define I_MOV(k) "v_mov_b32_e32 %" #k ", %8\n"
#define I_AND(k) "v_and_b32_e32 %" #k ", %8, %" #k "\n"
#define I_CND(k) "v_cndmask_b32_e32 %" #k ", %8, %" #k ", vcc\n"
#define I_FMAC(k) "v_fmac_f32_e32 %" #k ", %8, %9\n"
#define I_CMP(k) "v_cmp_gt_f32_e32 vcc, %8, %" #k "\n"
#define I_MAX(k) "v_max_f32_e32 %" #k ", %8, %" #k "\n"
#define I_XOR(k) "v_xor_b32_e32 %" #k ", %8, %" #k "\n"
#define P8(INS) asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "s"(spair), "v"(av))
#define I_PKS(k) "v_pk_fma_f32 %" #k ", %" #k ", %8, %" #k " op_sel_hi:[1,0,1]\n"      /* constant: low half of an SGPR pair */
#define I_PKV(k) "v_pk_fma_f32 %" #k ", %" #k ", %9, %" #k "\n"                         /* constant: a VGPR pair */
#define I_FMAAK(k) "v_fmaak_f32 %" #k ", %8, %" #k ", 0x3c08839e\n"                     /* VOP2 with a 32-bit literal */
#define I_FMA3(k) "v_fma_f32 %" #k ", %" #k ", %8, %9\n"                                 /* VOP3, all VGPRs */
#define I_FMAMK(k) "v_fmamk_f32 %" #k ", %" #k ", 0x3c08839e, %8\n"                     /* VOP2, literal multiplier */
#define I_FMAS(k) "v_fma_f32 %" #k ", %" #k ", %10, %8\n"                                /* VOP3, one SGPR operand */
#define I_FMAI(k) "v_fma_f32 %" #k ", %" #k ", %8, 1.0\n"                                /* VOP3, inline constant addend */
#define I_SUB(k) "v_sub_f32_e32 %" #k ", %8, %" #k "\n"
#define I_MIN(k) "v_min_f32_e32 %" #k ", %8, %" #k "\n"
#define I_MAX3(k) "v_max3_f32 %" #k ", %" #k ", %8, %9\n"
#define I_BFI(k) "v_bfi_b32 %" #k ", %8, %" #k ", %9\n"
#define I_BITOP(k) "v_bitop3_b32 %" #k ", %" #k ", %8, %9 bitop3:0x6c\n"
#define I_LSHL(k) "v_lshlrev_b32_e32 %" #k ", 3, %" #k "\n"
#define I_ASHR(k) "v_ashrrev_i32_e32 %" #k ", 1, %" #k "\n"
#define I_ADDU(k) "v_add_u32_e32 %" #k ", %8, %" #k "\n"
#define I_CVT(k) "v_cvt_i32_f32_e32 %" #k ", %" #k "\n"
#define I_RND(k) "v_rndne_f32_e32 %" #k ", %" #k "\n"
#define I_FLOOR(k) "v_floor_f32_e32 %" #k ", %" #k "\n"
#define I_PKMUL(k) "v_pk_mul_f32 %" #k ", %" #k ", %9\n"
#define I_PKADD(k) "v_pk_add_f32 %" #k ", %" #k ", %9\n"
#define I_CNDS(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %8, s[10:11]\n"                   /* select on an SGPR mask: no vcc dependency */
#define I_MULS(k) "v_mul_f32_e32 %" #k ", %10, %" #k "\n"                                /* SGPR multiplier */
#define I_ABSMAX(k) "v_max_f32_e64 %" #k ", |%" #k "|, %8\n"
#define A8S(INS) asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(av.x), "v"(av.y), "s"(sa) : "s10", "s11")
                if (KIND == 21) A8(I_FMAMK);
                else if (KIND == 22) A8S(I_FMAS);
                else if (KIND == 23) A8(I_FMAI);
                else if (KIND == 24) A8(I_SUB);
                else if (KIND == 25) A8(I_MIN);
                else if (KIND == 26) A8(I_MAX3);
                else if (KIND == 27) A8(I_BFI);
                else if (KIND == 28) A8(I_BITOP);
                else if (KIND == 29) A8(I_LSHL);
                else if (KIND == 30) A8(I_ASHR);
                else if (KIND == 31) A8(I_ADDU);
                else if (KIND == 32) A8(I_CVT);
                else if (KIND == 33) A8(I_RND);
                else if (KIND == 34) A8(I_FLOOR);
                else if (KIND == 35) P8(I_PKMUL);
                else if (KIND == 36) P8(I_PKADD);
                else if (KIND == 37) A8S(I_CNDS);
                else if (KIND == 38) A8S(I_MULS);
                else if (KIND == 39) A8(I_ABSMAX);
                else if (KIND == 17) P8(I_PKS);
                else if (KIND == 18) P8(I_PKV);
                else if (KIND == 19) A8(I_FMAAK);
                else if (KIND == 20) A8(I_FMA3);
                else if (KIND == 8) A8(I_ADD);
                else if (KIND == 9) A8(I_MUL);
                else if (KIND == 10) A8(I_MOV);
                else if (KIND == 11) A8(I_AND);
                else if (KIND == 12) A8(I_CND);
                else if (KIND == 13) A8(I_FMAC);
                else if (KIND == 14) A8(I_CMP);
                else if (KIND == 15) A8(I_MAX);
                else A8(I_XOR);
            }
        }
    }
    float r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y +
              p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
    if (r == 12345.678f) out[0] = r;
}

static float* g_out;
static float* g_prm;

template <int KIND, int WPS>
void run(const char* name, int trips) {
    const int blocks = 256 * WPS;          // one 4-wave workgroup per CU and per wave-per-SIMD step
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((calib<KIND, WPS>), dim3(blocks), dim3(256), 0, 0, g_out, g_prm, trips);
    (void)hipEventRecord(e0);
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((calib<KIND, WPS>), dim3(blocks), dim3(256), 0, 0, g_out, g_prm, trips);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double per_wave = (double)trips * UNROLL * 8.0;
    const double wave_instr = per_wave * 4.0 * blocks;                  // whole launch
    const double per_simd = per_wave * WPS;                            // instructions one SIMD issues
    // cycles at 2.4 GHz nominal; the real clock comes from GRBM_GUI_ACTIVE of the counter run
    printf("{\"kind\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"wave_instructions\": %.0f, \"ns_per_instr_per_simd\": %.4f, "
           "\"cycles_per_instr_at_2400MHz\": %.3f}\n", name, WPS, ms, wave_instr, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
}

template <int KIND>
void sweep(const char* name, int trips) {
    run<KIND, 1>(name, trips); run<KIND, 2>(name, trips); run<KIND, 4>(name, trips); run<KIND, 8>(name, trips);
}

int main(int argc, char** argv) {
    int trips = argc > 1 ? atoi(argv[1]) : 4096;
    (void)hipMalloc(&g_out, 4); (void)hipMalloc(&g_prm, 8);
    float h[2] = {0.9999f, 1.0e-4f};
    (void)hipMemcpy(g_prm, h, 8, hipMemcpyHostToDevice);
    sweep<0>("v_fma_f32", trips); sweep<1>("v_pk_fma_f32", trips); sweep<2>("v_sqrt_f32", trips); sweep<3>("v_rcp_f32", trips);
    sweep<4>("v_sin_f32", trips); sweep<5>("v_max_f32", trips); sweep<6>("v_cmp+v_cndmask x4", trips); sweep<7>("mix 6 fma 1 sqrt 1 rcp", trips);
    sweep<8>("v_add_f32_e32", trips); sweep<9>("v_mul_f32_e32", trips); sweep<10>("v_mov_b32_e32", trips); sweep<11>("v_and_b32_e32", trips);
    sweep<12>("v_cndmask_b32_e32", trips); sweep<13>("v_fmac_f32_e32", trips); sweep<14>("v_cmp_gt_f32_e32", trips); sweep<15>("v_max_f32_e32", trips);
    sweep<16>("v_xor_b32_e32", trips);
    sweep<17>("v_pk_fma_f32 sgpr-pair constant", trips); sweep<18>("v_pk_fma_f32 vgpr-pair constant", trips);
    sweep<19>("v_fmaak_f32 literal", trips); sweep<20>("v_fma_f32 vop3 vgprs", trips);
    sweep<21>("v_fmamk_f32 literal", trips); sweep<22>("v_fma_f32 vop3 one sgpr", trips); sweep<23>("v_fma_f32 vop3 inline constant", trips);
    sweep<24>("v_sub_f32_e32", trips); sweep<25>("v_min_f32_e32", trips); sweep<26>("v_max3_f32", trips); sweep<27>("v_bfi_b32", trips);
    sweep<28>("v_bitop3_b32", trips); sweep<29>("v_lshlrev_b32_e32", trips); sweep<30>("v_ashrrev_i32_e32", trips);
    sweep<31>("v_add_u32_e32", trips); sweep<32>("v_cvt_i32_f32_e32", trips); sweep<33>("v_rndne_f32_e32", trips);
    sweep<34>("v_floor_f32_e32", trips); sweep<35>("v_pk_mul_f32", trips); sweep<36>("v_pk_add_f32", trips);
    sweep<37>("v_cndmask_b32_e64 sgpr mask", trips); sweep<38>("v_mul_f32_e32 sgpr operand", trips); sweep<39>("v_max_f32_e64 abs modifier", trips);
    return 0;
}

// Developer tool: achievable bandwidth of (3,N)->(N) streaming under different point-to-lane mappings.
// out = x + y + z. Variants: flat float4; row blocks of ZL points x RW rows per load instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void flat4(const float* __restrict__ co, long long stride, long long n, float* __restrict__ out) {
    long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        float4 x = *(const float4*)(co + i), y = *(const float4*)(co + stride + i), z = *(const float4*)(co + 2 * stride + i);
        *(float4*)(out + i) = make_float4(x.x + y.x + z.x, x.y + y.y + z.y, x.z + y.z + z.z, x.w + y.w + z.w);
    }
}
__global__ __launch_bounds__(256) void flat4nt(const float* __restrict__ co, long long stride, long long n, float* __restrict__ out) {
    long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    typedef float f4 __attribute__((ext_vector_type(4)));
    if (i + 3 < n) {
        f4 x = __builtin_nontemporal_load((const f4*)(co + i)), y = __builtin_nontemporal_load((const f4*)(co + stride + i)),
           z = __builtin_nontemporal_load((const f4*)(co + 2 * stride + i));
        __builtin_nontemporal_store(x + y + z, (f4*)(out + i));
    }
}
__global__ __launch_bounds__(256) void flat4nts(const float* __restrict__ co, long long stride, long long n, float* __restrict__ out) {
    long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    typedef float f4 __attribute__((ext_vector_type(4)));
    if (i + 3 < n) {
        f4 x = *((const f4*)(co + i)), y = *((const f4*)(co + stride + i)), z = *((const f4*)(co + 2 * stride + i));
        __builtin_nontemporal_store(x + y + z, (f4*)(out + i));
    }
}
// 2 quads per thread (more bytes in flight per wave)
__global__ __launch_bounds__(256) void flat8(const float* __restrict__ co, long long stride, long long n, float* __restrict__ out) {
    long long i = ((long long)blockIdx.x * 512 + threadIdx.x) * 4;
    typedef float f4 __attribute__((ext_vector_type(4)));
    if (i + 1024 + 3 < n) {
        f4 x = *((const f4*)(co + i)), y = *((const f4*)(co + stride + i)), z = *((const f4*)(co + 2 * stride + i));
        f4 x2 = *((const f4*)(co + i + 1024)), y2 = *((const f4*)(co + stride + i + 1024)), z2 = *((const f4*)(co + 2 * stride + i + 1024));
        *(f4*)(out + i) = x + y + z;
        *(f4*)(out + i + 1024) = x2 + y2 + z2;
    }
}
// one wave instruction = LPR lanes x 16 B per row, 64/LPR rows; a wave handles NB consecutive chunks along z
template <int LPR, int NB>
__global__ __launch_bounds__(256) void rows(const float* __restrict__ co, long long stride, unsigned L, long long R, unsigned nchunk,
                                            float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    constexpr int RW = 64 / LPR, ZL = LPR * 4;
    const unsigned groups = (nchunk + NB - 1) / NB;
    const long long rb = wave / groups;
    const unsigned cg = (unsigned)(wave - rb * groups);
    long long row = rb * RW + lane / LPR;
    if (row >= R) return;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const unsigned z0 = (cg * NB + j) * ZL + 4 * (lane % LPR);
        if (z0 + 3 < L) {
            const long long i = row * L + z0;
            f4u x = *(const f4u*)(co + i), y = *(const f4u*)(co + stride + i), z = *(const f4u*)(co + 2 * stride + i);
            f4u r = x + y + z;
            *(f4u*)(out + i) = r;
        }
    }
}
template <typename F> float time_it(F f, int reps) {
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    f(); CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}
template <int LPR, int NB> void run_rows(const float* co, long long stride, unsigned L, long long R, float* out, const char* name) {
    constexpr int RW = 64 / LPR, ZL = LPR * 4;
    unsigned nchunk = (L + ZL - 1) / ZL, groups = (nchunk + NB - 1) / NB;
    long long waves = ((R + RW - 1) / RW) * groups;
    unsigned blocks = (unsigned)((waves + 3) / 4);
    float ms = time_it([&] { hipLaunchKernelGGL((rows<LPR, NB>), dim3(blocks), dim3(256), 0, 0, co, stride, L, R, nchunk, out); }, 10);
    printf("%-28s L=%u: %.3f ms  %.0f GB/s\n", name, L, ms, 16.0 * L * R / ms / 1e6);
}
int main(int argc, char** argv) {
    for (unsigned L : {1024u}) {
        const long long R = 1025LL * 1025LL, n = R * L, stride = (n + 255) / 256 * 256;
        float *co, *out;
        CHK(hipMalloc(&co, 3 * stride * 4)); CHK(hipMalloc(&out, stride * 4));
        CHK(hipMemset(co, 0, 3 * stride * 4));
        float ms = time_it([&] { hipLaunchKernelGGL(flat4, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, 0, co, stride, n, out); }, 10);
        printf("%-28s L=%u: %.3f ms  %.0f GB/s\n", "flat float4", L, ms, 16.0 * n / ms / 1e6);
        ms = time_it([&] { hipLaunchKernelGGL(flat4nt, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, 0, co, stride, n, out); }, 10);
        printf("%-28s L=%u: %.3f ms  %.0f GB/s\n", "flat float4 nontemporal", L, ms, 16.0 * n / ms / 1e6);
        ms = time_it([&] { hipLaunchKernelGGL(flat4nts, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, 0, co, stride, n, out); }, 10);
        printf("%-28s L=%u: %.3f ms  %.0f GB/s\n", "flat float4 nt store only", L, ms, 16.0 * n / ms / 1e6);
        ms = time_it([&] { hipLaunchKernelGGL(flat8, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, 0, co, stride, n, out); }, 10);
        printf("%-28s L=%u: %.3f ms  %.0f GB/s\n", "flat 2 quads/thread", L, ms, 16.0 * n / ms / 1e6);
        run_rows<8, 1>(co, stride, L, R, out, "8 rows x 32, 1 chunk/wave");
        run_rows<8, 4>(co, stride, L, R, out, "8 rows x 32, 4 chunks/wave");
        run_rows<8, 8>(co, stride, L, R, out, "8 rows x 32, 8 chunks/wave");
        run_rows<16, 4>(co, stride, L, R, out, "4 rows x 64, 4 chunks/wave");
        run_rows<4, 4>(co, stride, L, R, out, "16 rows x 16, 4 chunks/wave");
        run_rows<32, 4>(co, stride, L, R, out, "2 rows x 128, 4 chunks/wave");
        run_rows<64, 4>(co, stride, L, R, out, "1 row x 256, 4 chunks/wave");
        CHK(hipFree(co)); CHK(hipFree(out));
    }
    return 0;
}

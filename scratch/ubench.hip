// VALU rate microbenchmark: cycles per wave64 instruction per SIMD for fma / pk_fma / rsq / mix.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define ITERS 2048
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float seed, long long *cyc)
{
    float a[8]; v2f p[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; p[i] = v2f{a[i], a[i] + 1}; }
    const float m = 1.0001f, c = 0.5f; const v2f m2 = {m, m}, c2 = {c, c};
    long long t0 = clock64();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) a[i] = fmaf(a[i], m, c);
            if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], m2, c2);
            if (MODE == 2) a[i] = __builtin_amdgcn_rsqf(a[i]);
            if (MODE == 3) { p[i] = p[i] * m2; }
            if (MODE == 4) { p[i] = p[i] + c2; }
            if (MODE == 5) { a[i] = fmaf(a[i], m, c); p[i] = __builtin_elementwise_fma(p[i], m2, c2); }
        }
    }
    long long t1 = clock64();
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE> void run(const char *name, int blocks_per_cu, int ops_per_iter)
{
    float *out; long long *cyc, hc;
    int blocks = 256 * blocks_per_cu;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 1.0f, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, 1.0f, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    double winstr = (double) blocks * 4 * ITERS * ops_per_iter;  // wave-instructions total
    double per_simd = winstr / 1024.0;
    printf("%-28s blocks/CU=%d  %.3f ms  ns per wave-instr per SIMD = %.3f  (clock64 delta %lld -> %.2f clk/instr/wave; implied GHz %.2f)\n",
           name, blocks_per_cu, ms, ms * 1e6 / per_simd, hc, (double) hc / (ITERS * ops_per_iter),
           (double) hc / (ms * 1e6));
    hipFree(out); hipFree(cyc);
}
int main()
{
    for (int b : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", b, 8);
        run<1>("v_pk_fma_f32", b, 8);
        run<2>("v_rsq_f32", b, 8);
        run<3>("v_pk_mul_f32", b, 8);
        run<4>("v_pk_add_f32", b, 8);
        run<5>("fma+pk_fma", b, 16);
    }
    return 0;
}

// Force-loop microbenchmark: interactions/s of candidate inner loops, sources broadcast from LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define REPS 256
struct F4 { float x, y, z, w; };

template <int MODE>
__global__ __launch_bounds__(256) void k(const float *src, float *out, float eps2)
{
    __shared__ v4f lxy[4][32];
    __shared__ v4f lzm[4][32];
    __shared__ F4 aos[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    {   // fill 64 sources
        const float x = src[lane * 4 + 0], y = src[lane * 4 + 1], z = src[lane * 4 + 2], m = src[lane * 4 + 3];
        aos[w][lane] = F4{x, y, z, m};
        float *a = (float *) lxy[w], *b = (float *) lzm[w];
        const int p = lane >> 1, h = lane & 1;
        a[4 * p + h] = x; a[4 * p + 2 + h] = y; b[4 * p + h] = z; b[4 * p + 2 + h] = m;
    }
    __syncthreads();
    const float px = src[256 + threadIdx.x], py = px * 0.5f, pz = px * 0.25f;
    if (MODE == 0) {  // scalar
        float fx = 0, fy = 0, fz = 0;
        for (int r = 0; r < REPS; ++r) {
#pragma unroll 8
            for (int j = 0; j < 64; ++j) {
                const F4 s = aos[w][j];
                const float dx = s.x - px, dy = s.y - py, dz = s.z - pz;
                const float t = fmaf(dz, dz, fmaf(dy, dy, fmaf(dx, dx, eps2)));
                float inv = __builtin_amdgcn_rsqf(t * t * t) * s.w;
                fx = fmaf(dx, inv, fx); fy = fmaf(dy, inv, fy); fz = fmaf(dz, inv, fz);
            }
        }
        out[blockIdx.x * 256 + threadIdx.x] = fx + fy + fz;
    } else {
        v2f fx = {0, 0}, fy = {0, 0}, fz = {0, 0};
        v2f gx = {0, 0}, gy = {0, 0}, gz = {0, 0};
        const v2f px2 = {px, px}, py2 = {py, py}, pz2 = {pz, pz}, e2 = {eps2, eps2};
        auto pf = [&](const v4f a, const v4f c, v2f &ax, v2f &ay, v2f &az) {
            const v2f dx = a.xy - px2, dy = a.zw - py2, dz = c.xy - pz2;
            const v2f s = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, __builtin_elementwise_fma(dx, dx, e2)));
            const v2f s3 = s * s * s;
            v2f inv = {__builtin_amdgcn_rsqf(s3.x), __builtin_amdgcn_rsqf(s3.y)};
            inv *= c.zw;
            ax = __builtin_elementwise_fma(dx, inv, ax);
            ay = __builtin_elementwise_fma(dy, inv, ay);
            az = __builtin_elementwise_fma(dz, inv, az);
        };
        for (int r = 0; r < REPS; ++r) {
            if (MODE == 1) {
#pragma unroll 8
                for (int j = 0; j < 32; ++j) pf(lxy[w][j], lzm[w][j], fx, fy, fz);
            } else {  // two independent pairs per iteration
#pragma unroll 4
                for (int j = 0; j < 32; j += 2) {
                    const v4f a0 = lxy[w][j], c0 = lzm[w][j], a1 = lxy[w][j + 1], c1 = lzm[w][j + 1];
                    const v2f dx0 = a0.xy - px2, dx1 = a1.xy - px2;
                    const v2f dy0 = a0.zw - py2, dy1 = a1.zw - py2;
                    const v2f dz0 = c0.xy - pz2, dz1 = c1.xy - pz2;
                    v2f s0 = __builtin_elementwise_fma(dx0, dx0, e2), s1 = __builtin_elementwise_fma(dx1, dx1, e2);
                    s0 = __builtin_elementwise_fma(dy0, dy0, s0); s1 = __builtin_elementwise_fma(dy1, dy1, s1);
                    s0 = __builtin_elementwise_fma(dz0, dz0, s0); s1 = __builtin_elementwise_fma(dz1, dz1, s1);
                    v2f q0 = s0 * s0, q1 = s1 * s1;
                    q0 = q0 * s0; q1 = q1 * s1;
                    v2f i0 = {__builtin_amdgcn_rsqf(q0.x), __builtin_amdgcn_rsqf(q0.y)};
                    v2f i1 = {__builtin_amdgcn_rsqf(q1.x), __builtin_amdgcn_rsqf(q1.y)};
                    i0 *= c0.zw; i1 *= c1.zw;
                    fx = __builtin_elementwise_fma(dx0, i0, fx); gx = __builtin_elementwise_fma(dx1, i1, gx);
                    fy = __builtin_elementwise_fma(dy0, i0, fy); gy = __builtin_elementwise_fma(dy1, i1, gy);
                    fz = __builtin_elementwise_fma(dz0, i0, fz); gz = __builtin_elementwise_fma(dz1, i1, gz);
                }
            }
        }
        out[blockIdx.x * 256 + threadIdx.x] = fx.x + fx.y + fy.x + fy.y + fz.x + fz.y + gx.x + gx.y + gy.x + gy.y + gz.x + gz.y;
    }
}
template <int MODE> void run(const char *name, int bpc, const float *src, float *out)
{
    const int blocks = 256 * bpc * 4;  // several rounds
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) k<MODE><<<blocks, 256>>>(src, out, 0.01f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) k<MODE><<<blocks, 256>>>(src, out, 0.01f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    const double inter = (double) blocks * 256 * 64 * REPS;
    printf("%-22s grid=%d  %.3f ms  %.3e interactions/s  (%.1f TF @20)\n", name, blocks, ms, inter / (ms * 1e-3), inter * 20 / (ms * 1e-3) / 1e12);
}
int main()
{
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 0.1f + 0.001f * i;
    float *src, *out; hipMalloc(&src, sizeof h); hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice);
    hipMalloc(&out, 256 * 8 * 4 * 256 * 4);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("scalar", 8, src, out);
        run<1>("packed", 8, src, out);
        run<2>("packed x2 ILP", 8, src, out);
    }
    return 0;
}

// Force-loop microbenchmark, sources through the SCALAR data path (s_load_dwordx16 -> SGPR
// operands of the packed ops), as in k_traverse's leaf runs; compared with LDS-broadcast
// sources at the same occupancy.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v8f __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define REPS 64

// MODE 0: LDS broadcast, 1: SGPR from a small (cache-resident) array, 2: SGPR streaming
// through a large array (every wave its own 64 KB window per rep), 3: like 2 but runs of 16 pairs with
// the first load of every run exposed (as at the start of a leaf run)
template <int MODE>
__global__ __launch_bounds__(256, 5) void k(const float *__restrict__ src, const v16f *__restrict__ big, size_t big_recs,
                                            float *__restrict__ out, float eps2)
{
    extern __shared__ char dyn[];   // occupancy limiter
    __shared__ v4f lxy[4][32];
    __shared__ v4f lzm[4][32];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    {
        const float x = src[lane * 4 + 0], y = src[lane * 4 + 1], z = src[lane * 4 + 2], m = src[lane * 4 + 3];
        float *a = (float *) lxy[w], *b = (float *) lzm[w];
        const int p = lane >> 1, h = lane & 1;
        a[4 * p + h] = x; a[4 * p + 2 + h] = y; b[4 * p + h] = z; b[4 * p + 2 + h] = m;
    }
    __syncthreads();
    const float px = src[256 + threadIdx.x], py = px * 0.5f, pz = px * 0.25f;
    v2f fx = {0, 0}, fy = {0, 0}, fz = {0, 0};
    const v2f px2 = {px, px}, py2 = {py, py}, pz2 = {pz, pz}, e2 = {eps2, eps2};
    auto pf = [&](const v2f X, const v2f Y, const v2f Z, const v2f M) {
        const v2f dx = X - px2, dy = Y - py2, dz = Z - pz2;
        const v2f s = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, __builtin_elementwise_fma(dx, dx, e2)));
        const v2f s3 = s * s * s;
        v2f inv = {__builtin_amdgcn_rsqf(s3.x), __builtin_amdgcn_rsqf(s3.y)};
        inv *= M;
        fx = __builtin_elementwise_fma(dx, inv, fx);
        fy = __builtin_elementwise_fma(dy, inv, fy);
        fz = __builtin_elementwise_fma(dz, inv, fz);
    };
    auto pf8 = [&](const v8f r) { pf(v2f{r[0], r[1]}, v2f{r[2], r[3]}, v2f{r[4], r[5]}, v2f{r[6], r[7]}); };
    const int gw = __builtin_amdgcn_readfirstlane((int) (blockIdx.x * 4 + w));
    if (MODE == 0) {
        for (int r = 0; r < REPS * 16; ++r) {
#pragma unroll 8
            for (int j = 0; j < 32; ++j) pf(lxy[w][j].xy, lxy[w][j].zw, lzm[w][j].xy, lzm[w][j].zw);
        }
    } else {
        for (int r = 0; r < REPS; ++r) {
            // 1024 pairs = 512 x16 records per rep
            const size_t window = (MODE == 4 || MODE == 6) ? 32768 : big_recs;
            size_t base = MODE == 1 ? 0 : ((size_t) gw * 977 + (size_t) r * 131) * 512 % (window - 512);
            const v16f *P = big + base;
            if (MODE == 5 || MODE == 6) {
                // three register sets: every wait covers requests that were issued one and two
                // record evaluations earlier
                v16f a = P[0], b = P[1], c;
                for (int d = 0; d < 510; d += 3) {
                    __builtin_amdgcn_s_waitcnt(0xC07F);      // a, b ready
                    c = P[d + 2];
                    pf8(a.lo); pf8(a.hi);
                    a = P[d + 3];
                    pf8(b.lo); pf8(b.hi);
                    __builtin_amdgcn_s_waitcnt(0xC07F);      // c, a ready
                    b = P[d + 4];
                    pf8(c.lo); pf8(c.hi);
                }
            } else if (MODE == 3) {
                for (int run = 0; run < 64; ++run) {   // 64 runs of 8 records
                    const v16f *Q = P + run * 8;
                    int d = 0; const int d1 = 8;
                    v16f ra = Q[0], rb;
                    while (true) {
                        const bool hb = d + 1 < d1;
                        __builtin_amdgcn_s_waitcnt(0xC07F);
                        if (hb) rb = Q[d + 1];
                        pf8(ra.lo); pf8(ra.hi);
                        if (!hb) break;
                        const bool ha = d + 2 < d1;
                        __builtin_amdgcn_s_waitcnt(0xC07F);
                        if (ha) ra = Q[d + 2];
                        pf8(rb.lo); pf8(rb.hi);
                        if (!ha) break;
                        d += 2;
                    }
                }
            } else {
                const int d1 = MODE == 1 ? 16 : 512;
                for (int rr = 0; rr < (MODE == 1 ? 32 : 1); ++rr) {
                    int d = 0;
                    v16f ra = P[0], rb;
                    while (true) {
                        const bool hb = d + 1 < d1;
                        __builtin_amdgcn_s_waitcnt(0xC07F);
                        if (hb) rb = P[d + 1];
                        pf8(ra.lo); pf8(ra.hi);
                        if (!hb) break;
                        const bool ha = d + 2 < d1;
                        __builtin_amdgcn_s_waitcnt(0xC07F);
                        if (ha) ra = P[d + 2];
                        pf8(rb.lo); pf8(rb.hi);
                        if (!ha) break;
                        d += 2;
                    }
                }
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = fx.x + fx.y + fy.x + fy.y + fz.x + fz.y;
}
template <int MODE> void run(const char *name, int lds_bytes, const float *src, const v16f *big, size_t recs, float *out)
{
    const int blocks = 256 * 5 * 6;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void *) k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    for (int i = 0; i < 2; ++i) k<MODE><<<blocks, 256, lds_bytes>>>(src, big, recs, out, 0.01f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) k<MODE><<<blocks, 256, lds_bytes>>>(src, big, recs, out, 0.01f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double inter = (double) blocks * 256 * (MODE == 0 ? 1024.0 : 2048.0) * REPS;
    printf("%-34s lds=%6d  %.3f ms  %.3e interactions/s  (%.1f TF @20)\n", name, lds_bytes, ms, inter / (ms * 1e-3), inter * 20 / (ms * 1e-3) / 1e12);
}
int main()
{
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 0.1f + 0.001f * i;
    float *src, *out; hipMalloc(&src, sizeof h); hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice);
    hipMalloc(&out, 256 * 5 * 6 * 256 * 4);
    const size_t recs = (size_t) 1 << 20;   // 64 MB
    v16f *big; hipMalloc(&big, recs * 64);
    float *hb = (float *) malloc(recs * 64);
    for (size_t i = 0; i < recs * 16; ++i) hb[i] = 0.1f + 1e-7f * (float) (i % 1000003);
    hipMemcpy(big, hb, recs * 64, hipMemcpyHostToDevice);
    for (int lds : {0, 30 * 1024}) {   // ~8, 6(5), 5 waves/SIMD
        run<0>("LDS broadcast", lds, src, big, recs, out);
        run<1>("SGPR, cache-resident", lds, src, big, recs, out);
        run<2>("SGPR, streaming 64 MB", lds, src, big, recs, out);
        run<3>("SGPR, streaming, runs of 16 pairs", lds, src, big, recs, out);
        run<4>("SGPR, 2 MB window", lds, src, big, recs, out);
        run<5>("SGPR, streaming, 3 sets", lds, src, big, recs, out);
        run<6>("SGPR, 2 MB window, 3 sets", lds, src, big, recs, out);
    }
    return 0;
}

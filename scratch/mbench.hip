// Microbenchmark: pairwise softened forces with the distance and accumulation GEMMs on the
// matrix cores (v_mfma_f32_4x4x1_16B_f32) and only the m * rsq(s)^3 step on the VALU.
//   phase 1: S[src j][body l]   = sum_k srcAug[j][k] * bodyAug[l][k]      (5 MFMA / 4 sources)
//   phase 2: w = m * rsq(S)^3                                              (VALU, 4 per lane)
//   phase 3: F[body][c] += w[body][src] * (x, y, z, 1)[src][c]            (4 MFMA / 4 sources)
// checked against the plain formula.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define NSRC 64
#define REPS 256

struct alignas(16) SrcLds {
    float aug[NSRC][8];        // x, y, z, |x|^2, 1, (pad)
    float tr[NSRC / 4][4][4];  // [group][component x,y,z,1][source in group]
    float m[NSRC];
};

template <int MODE>
__global__ __launch_bounds__(256) void k(const float *__restrict__ src, const float *__restrict__ bodies,
                                         float *__restrict__ out, float eps2)
{
    __shared__ SrcLds L[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    SrcLds &S = L[w];
    for (int s = lane; s < NSRC; s += 64) {
        const float x = src[4 * s], y = src[4 * s + 1], z = src[4 * s + 2], m = src[4 * s + 3];
        S.aug[s][0] = x; S.aug[s][1] = y; S.aug[s][2] = z; S.aug[s][3] = x * x + y * y + z * z; S.aug[s][4] = 1.f;
        S.tr[s >> 2][0][s & 3] = x; S.tr[s >> 2][1][s & 3] = y; S.tr[s >> 2][2][s & 3] = z; S.tr[s >> 2][3][s & 3] = 1.f;
        S.m[s] = m;
    }
    __syncthreads();
    const int b = (blockIdx.x * 4 + w) * 64 + lane;
    const float px = bodies[3 * (b % 4096)], py = bodies[3 * (b % 4096) + 1], pz = bodies[3 * (b % 4096) + 2];
    float ax = 0, ay = 0, az = 0;
    if (MODE == 0) {   // packed VALU reference (LDS broadcast)
        v2f fx = {0, 0}, fy = {0, 0}, fz = {0, 0};
        const v2f px2 = {px, px}, py2 = {py, py}, pz2 = {pz, pz}, e2 = {eps2, eps2};
        for (int r = 0; r < REPS; ++r)
#pragma unroll 4
            for (int s = 0; s < NSRC; s += 2) {
                const v2f X = {S.aug[s][0], S.aug[s + 1][0]}, Y = {S.aug[s][1], S.aug[s + 1][1]}, Z = {S.aug[s][2], S.aug[s + 1][2]};
                const v2f M = {S.m[s], S.m[s + 1]};
                const v2f dx = X - px2, dy = Y - py2, dz = Z - pz2;
                const v2f q = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, __builtin_elementwise_fma(dx, dx, e2)));
                const v2f q3 = q * q * q;
                v2f inv = {__builtin_amdgcn_rsqf(q3.x), __builtin_amdgcn_rsqf(q3.y)};
                inv *= M;
                fx = __builtin_elementwise_fma(dx, inv, fx); fy = __builtin_elementwise_fma(dy, inv, fy); fz = __builtin_elementwise_fma(dz, inv, fz);
            }
        ax = fx.x + fx.y; ay = fy.x + fy.y; az = fz.x + fz.y;
    } else {
        // body side of the distance GEMM: (-2x, -2y, -2z, 1, |x|^2 + eps^2)
        const float bk[5] = {-2.f * px, -2.f * py, -2.f * pz, 1.f, px * px + py * py + pz * pz + eps2};
        v4f F = {0, 0, 0, 0};
        const int j = lane & 3;
        for (int r = 0; r < REPS; ++r) {
#pragma unroll 2
            for (int g = 0; g < NSRC / 4; ++g) {
                const v4f a0 = *reinterpret_cast<const v4f *>(&S.aug[4 * g + j][0]);
                const float a4 = S.aug[4 * g + j][4];
                v4f D = {0, 0, 0, 0};
                D = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.x, bk[0], D, 0, 0, 0);
                D = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.y, bk[1], D, 0, 0, 0);
                D = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.z, bk[2], D, 0, 0, 0);
                D = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.w, bk[3], D, 0, 0, 0);
                D = __builtin_amdgcn_mfma_f32_4x4x1f32(a4, bk[4], D, 0, 0, 0);
                const v4f M = *reinterpret_cast<const v4f *>(&S.m[4 * g]);
                const v2f q01 = D.xy, q23 = D.zw;
                const v2f c01 = q01 * q01 * q01, c23 = q23 * q23 * q23;
                v2f w01 = {__builtin_amdgcn_rsqf(c01.x), __builtin_amdgcn_rsqf(c01.y)};
                v2f w23 = {__builtin_amdgcn_rsqf(c23.x), __builtin_amdgcn_rsqf(c23.y)};
                w01 *= M.xy; w23 *= M.zw;
                const v4f T = *reinterpret_cast<const v4f *>(&S.tr[g][j][0]);
                F = __builtin_amdgcn_mfma_f32_4x4x1f32(w01.x, T.x, F, 0, 0, 0);
                F = __builtin_amdgcn_mfma_f32_4x4x1f32(w01.y, T.y, F, 0, 0, 0);
                F = __builtin_amdgcn_mfma_f32_4x4x1f32(w23.x, T.z, F, 0, 0, 0);
                F = __builtin_amdgcn_mfma_f32_4x4x1f32(w23.y, T.w, F, 0, 0, 0);
            }
        }
        // lane (blk, c) holds F[body 4 blk + i][component c] in F[i]: gather the own body's row
        float comp[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int from = (lane & ~3) | c;    // lane holding component c of this quad
            const float v0 = __shfl(F[0], from, 64), v1 = __shfl(F[1], from, 64), v2 = __shfl(F[2], from, 64), v3 = __shfl(F[3], from, 64);
            comp[c] = j == 0 ? v0 : j == 1 ? v1 : j == 2 ? v2 : v3;
        }
        ax = comp[0] - px * comp[3]; ay = comp[1] - py * comp[3]; az = comp[2] - pz * comp[3];
    }
    out[3 * b] = ax; out[3 * b + 1] = ay; out[3 * b + 2] = az;
}

template <int MODE> double run(const char *name, const float *src, const float *bodies, float *out, std::vector<float> *res)
{
    const int blocks = 256 * 8 * 2;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(src, bodies, out, 0.0001f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) k<MODE><<<blocks, 256>>>(src, bodies, out, 0.0001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double inter = (double) blocks * 256 * NSRC * REPS;
    printf("%-28s %.3f ms  %.3e interactions/s  (%.1f TF @20)\n", name, ms, inter / (ms * 1e-3), inter * 20 / (ms * 1e-3) / 1e12);
    res->resize(3 * 4096);
    hipMemcpy(res->data(), out, res->size() * 4, hipMemcpyDeviceToHost);
    return ms;
}
int main()
{
    std::vector<float> hs(4 * NSRC), hb(3 * 4096);
    unsigned st = 12345;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return (st >> 8) * (1.0f / 16777216.0f); };
    for (int s = 0; s < NSRC; ++s) { hs[4 * s] = rnd() * 0.2f - 0.1f; hs[4 * s + 1] = rnd() * 0.2f - 0.1f; hs[4 * s + 2] = rnd() * 0.2f - 0.1f; hs[4 * s + 3] = 0.5f + rnd(); }
    for (auto &v : hb) v = rnd() * 0.02f - 0.01f;
    float *src, *bodies, *out;
    hipMalloc(&src, hs.size() * 4); hipMemcpy(src, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&bodies, hb.size() * 4); hipMemcpy(bodies, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&out, (size_t) 256 * 8 * 2 * 256 * 3 * 4);
    std::vector<float> r0, r1;
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("packed VALU (LDS)", src, bodies, out, &r0);
        run<1>("MFMA 4x4x1 + VALU rsq", src, bodies, out, &r1);
    }
    // accuracy of both against float64
    double e0 = 0, e1 = 0;
    for (int b = 0; b < 4096; ++b) {
        double a[3] = {0, 0, 0};
        for (int s = 0; s < NSRC; ++s) {
            const double dx = hs[4 * s] - (double) hb[3 * b], dy = hs[4 * s + 1] - (double) hb[3 * b + 1], dz = hs[4 * s + 2] - (double) hb[3 * b + 2];
            const double q = dx * dx + dy * dy + dz * dz + (double) 0.0001f;
            const double wgt = hs[4 * s + 3] / (q * std::sqrt(q));
            a[0] += dx * wgt; a[1] += dy * wgt; a[2] += dz * wgt;
        }
        const double nrm = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]) * REPS;
        double d0 = 0, d1 = 0;
        for (int c = 0; c < 3; ++c) { d0 += std::pow(r0[3 * b + c] - a[c] * REPS, 2); d1 += std::pow(r1[3 * b + c] - a[c] * REPS, 2); }
        e0 = std::max(e0, std::sqrt(d0) / nrm); e1 = std::max(e1, std::sqrt(d1) / nrm);
    }
    printf("max relative error vs float64: packed VALU %.3e   MFMA %.3e\n", e0, e1);
    return 0;
}

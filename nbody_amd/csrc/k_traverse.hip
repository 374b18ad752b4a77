// k_traverse.hip -- theta-criterion group traversal + softened FP32 force accumulation,
// and the tiled O(N^2) direct sum.
//
// Semantics restated from the reference (src/cuda/barnes_hut.cu:31-61,199-537;
// include/cuda/physics_common.cuh:36-75; src/cuda/validator.cu:134-174):
//   * a group is G consecutive Morton-sorted bodies and shares one interaction list;
//   * for a candidate node, d2min = min over the group's bodies of |x_b - com|^2,
//     starting from 3*domain^2;
//   * a node without children is evaluated directly against every particle of its range
//     (self included: contributes exactly 0 because eps^2 > 0);
//   * otherwise size^2 / d2min < theta^2 -> the group interacts with (com, mass), else the
//     node's children become candidates;
//   * acc = gravity * sum m_j d / (|d|^2 + eps^2)^(3/2).
// Design here: one wave64 per group, no workgroup barriers, no global-memory queues.
// Candidates live on a per-wave LDS stack; each round pops up to 64 of them, one per
// lane, and every lane tests its node against the 64 group bodies (broadcast LDS reads).
// Accepted (com, mass) records and the particles of leaf nodes are appended to one LDS
// source list; whenever it holds >= 64 entries each lane (= one body) runs the force
// loop over 64 broadcast entries.  The set of sources a group interacts with is
// independent of the pop order, so list lengths match the breadth-first reference walk.
#include "dev_utils.hpp"
#include "nbody_internal.hpp"

namespace nb {

constexpr int TRAV_RESERVE = 7 * 24;  // single-pop fallback never needs more than 7/level

__device__ __forceinline__ void interact(const Float4 s, float px, float py, float pz,
                                         float eps2, float &fx, float &fy, float &fz)
{
    const float dx = s.x - px, dy = s.y - py, dz = s.z - pz;
    const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    const float t = d2 + eps2;
    float inv = __builtin_amdgcn_rsqf(t * t * t);
    inv *= s.w;
    fx = fmaf(dx, inv, fx);
    fy = fmaf(dy, inv, fy);
    fz = fmaf(dz, inv, fz);
}

template <int G, bool FUSE_VEL>
__global__ __launch_bounds__(64 * TRAV_WAVES) void k_traverse(
    int n, int g0, int g1, const Float4 *__restrict__ posm, const Float4 *__restrict__ nodeA,
    const NodeB *__restrict__ nodeB, float theta, float eps2, float gravity, float domain,
    float *__restrict__ ax, float *__restrict__ ay, float *__restrict__ az,
    uint32_t *__restrict__ list_len, float dt, float keep, const float *__restrict__ vhx,
    const float *__restrict__ vhy, const float *__restrict__ vhz, float *__restrict__ vx,
    float *__restrict__ vy, float *__restrict__ vz, uint32_t *err)
{
    __shared__ Float4 s_body[TRAV_WAVES][64];
    __shared__ Float4 s_list[TRAV_WAVES][128];
    __shared__ int s_stack[TRAV_WAVES][TRAV_STACK];

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int lblk = xcd_remap((int) blockIdx.x, (int) gridDim.x);
    const int g = g0 + lblk * TRAV_WAVES + w;
    if (g >= g1) return;
    Float4 *body = s_body[w];
    Float4 *list = s_list[w];
    int *stack = s_stack[w];

    const int b = g * G + lane;
    const bool active = lane < G && b < n;
    const Float4 me = posm[active ? b : g * G];
    if (lane < G) body[lane] = me;
    if (lane == 0) stack[0] = 0;
    __builtin_amdgcn_wave_barrier();

    float fx = 0.f, fy = 0.f, fz = 0.f;
    int list_n = 0;
    uint32_t total = 0;
    int sp = 1;
    const float th2 = theta * theta;
    const float dinit = domain * domain * 3.0f;

    auto flush64 = [&]() {
        __builtin_amdgcn_wave_barrier();
        const Float4 *e = list + (list_n - 64);
#pragma unroll 8
        for (int j = 0; j < 64; ++j) interact(e[j], me.x, me.y, me.z, eps2, fx, fy, fz);
        list_n -= 64;
        __builtin_amdgcn_wave_barrier();
    };

    while (sp > 0) {
        const int room = TRAV_STACK - TRAV_RESERVE - sp;
        const int k = min(min(sp, 64), max(room / 7, 1));
        const bool valid = lane < k;
        const int node = valid ? stack[sp - 1 - lane] : 0;
        sp -= k;
        const Float4 A = nodeA[node];
        const NodeB B = nodeB[node];
        const bool leaf = valid && B.leaf != 0;
        const bool inner = valid && B.leaf == 0;

        float dmin = dinit;
#pragma unroll 8
        for (int j = 0; j < G; ++j) {
            const Float4 q = body[j];
            const float dx = q.x - A.x, dy = q.y - A.y, dz = q.z - A.z;
            const float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
            dmin = d <= dmin ? d : dmin;
        }
        const bool approx = inner && (__fdiv_rn(B.size * B.size, dmin) < th2);
        const bool open = inner && !approx;

        // accepted nodes -> source list
        const unsigned long long am = __ballot(approx);
        if (approx) list[list_n + mbcnt(am)] = A;
        const int na = (int) __popcll(am);
        list_n += na;
        total += (uint32_t) na;

        // opened nodes -> children onto the stack
        const uint32_t cnt = open ? (uint32_t) B.count : 0u;
        const uint32_t inc = wave_incl_scan_u32(cnt);
        const int tot = (int) __shfl(inc, 63, 64);
        if (sp + tot > TRAV_STACK) {  // cannot happen by construction; never write past LDS
            if (lane == 0) atomicOr(err, (uint32_t) DEV_ERR_STACK);
            break;
        }
        const int dst = sp + (int) (inc - cnt);
        for (uint32_t c = 0; c < cnt; ++c) stack[dst + (int) c] = B.first + (int) c;
        sp += tot;

        if (list_n >= 64) flush64();

        // leaf nodes -> their particles into the source list
        unsigned long long lm = __ballot(leaf);
        while (lm) {
            const int src = (int) __builtin_ctzll(lm);
            lm &= lm - 1;
            const int pb = __shfl(B.first, src, 64);
            const int np = __shfl(B.count, src, 64);
            for (int o = 0; o < np; o += 64) {
                const int c = min(64, np - o);
                if (lane < c) list[list_n + lane] = posm[pb + o + lane];
                list_n += c;
                total += (uint32_t) c;
                if (list_n >= 64) flush64();
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    // remainder (< 64 entries)
    __builtin_amdgcn_wave_barrier();
    for (int j = 0; j < list_n; ++j) interact(list[j], me.x, me.y, me.z, eps2, fx, fy, fz);

    if (lane == 0) list_len[g] = total;
    if (active) {
        const float axv = gravity * fx, ayv = gravity * fy, azv = gravity * fz;
        ax[b] = axv; ay[b] = ayv; az[b] = azv;
        if (FUSE_VEL) {
            vx[b] = fmaf(0.5f * axv, dt, vhx[b]) * keep;
            vy[b] = fmaf(0.5f * ayv, dt, vhy[b]) * keep;
            vz[b] = fmaf(0.5f * azv, dt, vhz[b]) * keep;
        }
    }
}

// sum over groups of list_len[g] * bodies_in_group(g) -> *out (u64), single workgroup.
__global__ __launch_bounds__(256) void k_count_interactions(int n, int group, int g0, int g1,
                                                            const uint32_t *__restrict__ list_len,
                                                            unsigned long long *out)
{
    __shared__ unsigned long long s[256];
    unsigned long long acc = 0;
    for (int g = g0 + (int) threadIdx.x; g < g1; g += 256) {
        const int nb = min(group, n - g * group);
        acc += (unsigned long long) list_len[g] * (unsigned long long) nb;
    }
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int) threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = s[0];
}

// ---------------------------------------------------------------------------
// Direct sum: 256 targets per workgroup, sources staged through LDS 256 at a time.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_direct(const Float4 *__restrict__ targets, int nt,
                                                const Float4 *__restrict__ sources, int ns,
                                                float eps2, float gravity, float *__restrict__ ax,
                                                float *__restrict__ ay, float *__restrict__ az,
                                                int accumulate)
{
    __shared__ Float4 s_src[256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const Float4 me = targets[i < nt ? i : 0];
    float fx = 0.f, fy = 0.f, fz = 0.f;
    for (int t0 = 0; t0 < ns; t0 += 256) {
        const int c = min(256, ns - t0);
        if ((int) threadIdx.x < c) s_src[threadIdx.x] = sources[t0 + threadIdx.x];
        __syncthreads();
        if (c == 256) {
#pragma unroll 8
            for (int j = 0; j < 256; ++j) interact(s_src[j], me.x, me.y, me.z, eps2, fx, fy, fz);
        } else {
            for (int j = 0; j < c; ++j) interact(s_src[j], me.x, me.y, me.z, eps2, fx, fy, fz);
        }
        __syncthreads();
    }
    if (i < nt) {
        float rx = gravity * fx, ry = gravity * fy, rz = gravity * fz;
        if (accumulate) { rx += ax[i]; ry += ay[i]; rz += az[i]; }
        ax[i] = rx; ay[i] = ry; az[i] = rz;
    }
}

hipError_t launch_traverse(Sim *s, bool fuse_vel)
{
    const int ngroups = (s->n + s->group - 1) / s->group;
    const int g0 = s->trav_g1 < 0 ? 0 : s->trav_g0;
    const int g1 = s->trav_g1 < 0 ? ngroups : std::min(s->trav_g1, ngroups);
    const int c = s->cur;
    if (g1 > g0) {
        const int grid = (g1 - g0 + TRAV_WAVES - 1) / TRAV_WAVES;
        const float eps2 = s->p.softening_factor * s->p.softening_factor;
        const float keep = 1.0f - s->p.velocity_dampening;
#define NB_TRAV(GG, FV)                                                                          \
    hipLaunchKernelGGL((k_traverse<GG, FV>), dim3(grid), dim3(64 * TRAV_WAVES), 0, s->stream,    \
                       s->n, g0, g1, s->posm, s->nodeA, s->nodeB, s->p.theta, eps2,              \
                       s->p.gravity, s->p.domain_size, s->ax, s->ay, s->az, s->list_len,         \
                       s->p.dt, keep, s->vhx[c], s->vhy[c], s->vhz[c], s->vx[c], s->vy[c],       \
                       s->vz[c], &s->counts->err)
        if (s->group == 64) {
            if (fuse_vel) NB_TRAV(64, true); else NB_TRAV(64, false);
        } else {
            if (fuse_vel) NB_TRAV(32, true); else NB_TRAV(32, false);
        }
#undef NB_TRAV
    }
    return hipGetLastError();
}

hipError_t launch_count_interactions(Sim *s, unsigned long long *out)
{
    const int ngroups = (s->n + s->group - 1) / s->group;
    const int g0 = s->trav_g1 < 0 ? 0 : s->trav_g0;
    const int g1 = s->trav_g1 < 0 ? ngroups : std::min(s->trav_g1, ngroups);
    hipLaunchKernelGGL(k_count_interactions, dim3(1), dim3(256), 0, s->stream, s->n, s->group, g0,
                       g1, s->list_len, out);
    return hipGetLastError();
}

hipError_t launch_direct(const Float4 *targets, int nt, const Float4 *sources, int ns, float eps2,
                         float gravity, float *ax, float *ay, float *az, int accumulate,
                         hipStream_t st)
{
    if (nt <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_direct, dim3((nt + 255) / 256), dim3(256), 0, st, targets, nt, sources, ns,
                       eps2, gravity, ax, ay, az, accumulate);
    return hipGetLastError();
}

}  // namespace nb

// k_points.hip -- per-particle streaming kernels: kick-drift + Morton encode, particle
// permute, velocity kick.  All are HBM-bound: one coalesced SoA pass each.
//
// Reference behaviour restated (not translated):
//   leapfrog_integrate_pos / leapfrog_integrate_vel, impose_boundary_conditions
//     (src/cuda/physics_common.cu:69-147)
//   morton_encode, expand_bits (src/cuda/points.cu:3-38)
//   Points::sort_by_codes gathers + BarnesHut::sort_bodies gathers
//     (src/cuda/points.cu:86-113, src/cuda/barnes_hut.cu:577-587)
#include "dev_utils.hpp"
#include "nbody_internal.hpp"

namespace nb {

// q = (uint)(p / domain * (2^B - 1)): IEEE divide, then multiply, truncate.  The host
// oracle performs the same two float operations; codes at cell boundaries depend on it.
template <typename KeyT>
__device__ __forceinline__ KeyT morton_code(float px, float py, float pz, float domain);

// The cube the keys are taken relative to: [0, domain]^3 like the reference, or, with
// NBODY_FLAG_FIT_KEYS, the cube (lo xyz, side) k_keyfit keeps in device memory.
struct KeyCube { float ox, oy, oz, size; };
__device__ __forceinline__ KeyCube key_cube(const float *__restrict__ keybox, float domain)
{
    return keybox ? KeyCube{keybox[0], keybox[1], keybox[2], keybox[3]} : KeyCube{0.f, 0.f, 0.f, domain};
}

template <>
__device__ __forceinline__ uint32_t morton_code<uint32_t>(float px, float py, float pz,
                                                          float domain)
{
    const float s = 1023.0f;
    uint32_t qx = (uint32_t) (__fdiv_rn(px, domain) * s);
    uint32_t qy = (uint32_t) (__fdiv_rn(py, domain) * s);
    uint32_t qz = (uint32_t) (__fdiv_rn(pz, domain) * s);
    return spread3_10(qx) * 4u + spread3_10(qy) * 2u + spread3_10(qz);
}

template <>
__device__ __forceinline__ uint64_t morton_code<uint64_t>(float px, float py, float pz,
                                                          float domain)
{
    const float s = 2097151.0f;
    uint64_t qx = (uint64_t) (__fdiv_rn(px, domain) * s);
    uint64_t qy = (uint64_t) (__fdiv_rn(py, domain) * s);
    uint64_t qz = (uint64_t) (__fdiv_rn(pz, domain) * s);
    return spread3_21(qx) * 4ull + spread3_21(qy) * 2ull + spread3_21(qz);
}

__device__ __forceinline__ void reflect(float &p, float &v, float domain)
{
    if (p < 0.0f || p > domain) {
        v *= -1.0f;
        p = fmaxf(0.0f, fminf(domain, p));
    }
}

// v_half = v + 0.5*a*dt;  x += v_half*dt;  reflect;  code = morton(x).
// 36 B read + 24 B + sizeof(KeyT) written per particle.  The first zero16 threads also clear 16
// bytes each of the scratch the sort and the tree build want zeroed (tickets, histograms,
// look-back words: a few MB), which saves the step a memset launch.
template <typename KeyT>
__global__ __launch_bounds__(256) void k_integrate_morton(
    int n, float dt, float domain, float *__restrict__ x, float *__restrict__ y,
    float *__restrict__ z, const float *__restrict__ vx, const float *__restrict__ vy,
    const float *__restrict__ vz, const float *__restrict__ ax, const float *__restrict__ ay,
    const float *__restrict__ az, float *__restrict__ vhx, float *__restrict__ vhy,
    float *__restrict__ vhz, KeyT *__restrict__ codes, uint4 *__restrict__ zero, int zero16)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < zero16) zero[i] = uint4{0u, 0u, 0u, 0u};
    if (i >= n) return;
    float hx = fmaf(0.5f * ax[i], dt, vx[i]);
    float hy = fmaf(0.5f * ay[i], dt, vy[i]);
    float hz = fmaf(0.5f * az[i], dt, vz[i]);
    float px = fmaf(hx, dt, x[i]);
    float py = fmaf(hy, dt, y[i]);
    float pz = fmaf(hz, dt, z[i]);
    reflect(px, hx, domain);
    reflect(py, hy, domain);
    reflect(pz, hz, domain);
    x[i] = px; y[i] = py; z[i] = pz;
    vhx[i] = hx; vhy[i] = hy; vhz[i] = hz;
    codes[i] = morton_code<KeyT>(px, py, pz, domain);
}

template <typename KeyT>
__global__ __launch_bounds__(256) void k_morton(int n, float domain, const float *__restrict__ keybox,
                                                const float *__restrict__ x,
                                                const float *__restrict__ y,
                                                const float *__restrict__ z,
                                                KeyT *__restrict__ codes)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const KeyCube c = key_cube(keybox, domain);
    codes[i] = morton_code<KeyT>(x[i] - c.ox, y[i] - c.oy, z[i] - c.oz, c.size);
}

// ---------------------------------------------------------------------------
// Fitted key domain (NBODY_FLAG_FIT_KEYS; not in the reference, whose keys always span
// [0, domain_size]^3 -- with 10 bits per axis a system that fills a small part of the domain
// is then a handful of cells across).  keybox = {lo x, lo y, lo z, side, valid} in device
// memory, followed by six words of bounding-box scratch (min xyz, max xyz as bit patterns:
// positions are >= 0, so float order is unsigned order).
//   k_integrate_bbox : kick-drift without the encode, bounding box of the new positions
//   k_bbox_soa       : bounding box of positions that are already there (spawn, set_state)
//   k_keyfit         : one thread; keeps the cube while it holds every particle and the
//                      particles' largest extent is at least 0.4 of its side, else re-draws it
//                      around the box: side = 1.25 x the largest extent (at least
//                      domain / 2^20), centred.  Float arithmetic in exactly this order (the
//                      host checker does the same).  Resets the scratch.
// then k_morton encodes relative to the cube.
// ---------------------------------------------------------------------------
// Workgroup reduction of per-thread bounds (bit patterns) and one guarded atomic per bound.
__device__ __forceinline__ void block_bbox(uint32_t (&mn)[3], uint32_t (&mx)[3], uint32_t *box)
{
    __shared__ uint32_t s[6];
    if (threadIdx.x < 6) s[threadIdx.x] = threadIdx.x < 3 ? 0x7f7fffffu : 0u;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        uint32_t a = mn[k], b = mx[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            a = min(a, (uint32_t) __shfl_xor((int) a, o, 64));
            b = max(b, (uint32_t) __shfl_xor((int) b, o, 64));
        }
        if ((threadIdx.x & 63) == 0) { atomicMin(&s[k], a); atomicMax(&s[3 + k], b); }
    }
    __syncthreads();
    // The global bounds only ever tighten: a workgroup that cannot improve one leaves it alone.
    // Every access to these words, here and in k_keyfit, is an agent-scope atomic: they are
    // updated from all XCDs, and an ordinary load may be served from an XCD's own L2.
    if (threadIdx.x < 6) {
        const uint32_t cur = __hip_atomic_load(&box[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x < 3) { if (s[threadIdx.x] < cur) atomicMin(&box[threadIdx.x], s[threadIdx.x]); }
        else if (s[threadIdx.x] > cur) atomicMax(&box[threadIdx.x], s[threadIdx.x]);
    }
}

constexpr int BBOX_GRID = 1024;   // workgroups of the two kernels below (grid-stride): 6 atomics each

__global__ __launch_bounds__(256) void k_integrate_bbox(
    int n, float dt, float domain, float *__restrict__ x, float *__restrict__ y,
    float *__restrict__ z, const float *__restrict__ vx, const float *__restrict__ vy,
    const float *__restrict__ vz, const float *__restrict__ ax, const float *__restrict__ ay,
    const float *__restrict__ az, float *__restrict__ vhx, float *__restrict__ vhy,
    float *__restrict__ vhz, uint32_t *box)
{
    uint32_t mn[3] = {0x7f7fffffu, 0x7f7fffffu, 0x7f7fffffu}, mx[3] = {0u, 0u, 0u};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        float hx = fmaf(0.5f * ax[i], dt, vx[i]);
        float hy = fmaf(0.5f * ay[i], dt, vy[i]);
        float hz = fmaf(0.5f * az[i], dt, vz[i]);
        float px = fmaf(hx, dt, x[i]);
        float py = fmaf(hy, dt, y[i]);
        float pz = fmaf(hz, dt, z[i]);
        reflect(px, hx, domain);
        reflect(py, hy, domain);
        reflect(pz, hz, domain);
        x[i] = px; y[i] = py; z[i] = pz;
        vhx[i] = hx; vhy[i] = hy; vhz[i] = hz;
        const uint32_t v[3] = {__float_as_uint(px), __float_as_uint(py), __float_as_uint(pz)};
#pragma unroll
        for (int k = 0; k < 3; ++k) { mn[k] = min(mn[k], v[k]); mx[k] = max(mx[k], v[k]); }
    }
    block_bbox(mn, mx, box);
}

__global__ __launch_bounds__(256) void k_bbox_soa(int n, const float *__restrict__ x,
                                                  const float *__restrict__ y,
                                                  const float *__restrict__ z, uint32_t *box)
{
    uint32_t mn[3] = {0x7f7fffffu, 0x7f7fffffu, 0x7f7fffffu}, mx[3] = {0u, 0u, 0u};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint32_t v[3] = {__float_as_uint(x[i]), __float_as_uint(y[i]), __float_as_uint(z[i])};
#pragma unroll
        for (int k = 0; k < 3; ++k) { mn[k] = min(mn[k], v[k]); mx[k] = max(mx[k], v[k]); }
    }
    block_bbox(mn, mx, box);
}

__global__ void k_keyfit(float *keybox, float domain)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t *box = reinterpret_cast<uint32_t *>(keybox + 5);
    float lo[3], hi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = __uint_as_float(__hip_atomic_load(&box[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        hi[a] = __uint_as_float(__hip_atomic_load(&box[3 + a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        __hip_atomic_store(&box[a], 0x7f7fffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&box[3 + a], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const float ksize = keybox[3];
    float side = 0.f;
    bool inside = keybox[4] != 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        side = fmaxf(side, hi[a] - lo[a]);
        inside = inside && lo[a] >= keybox[a] && hi[a] <= keybox[a] + ksize;
    }
    if (inside && side >= 0.4f * ksize) return;
    float size_new = 1.25f * side;
    const float floor_ = domain * (1.0f / 1048576.0f);
    if (!(size_new >= floor_)) size_new = floor_;
#pragma unroll
    for (int a = 0; a < 3; ++a) keybox[a] = 0.5f * (lo[a] + hi[a]) - 0.5f * size_new;
    keybox[3] = size_new;
    keybox[4] = 1.0f;
}

// Gather every per-particle array through the sort permutation in one pass and emit the
// packed (x,y,z,m) record the traversal reads.  ALL = also velocity and acceleration
// (needed only right after a spawn; during a step both are overwritten later in the step).
template <bool ALL>
__global__ __launch_bounds__(256) void k_permute(
    int n, const uint32_t *__restrict__ perm, const float *__restrict__ xi,
    const float *__restrict__ yi, const float *__restrict__ zi, const float *__restrict__ mi,
    const float *__restrict__ hxi, const float *__restrict__ hyi, const float *__restrict__ hzi,
    const uint32_t *__restrict__ idi, float *__restrict__ xo, float *__restrict__ yo,
    float *__restrict__ zo, float *__restrict__ mo, float *__restrict__ hxo,
    float *__restrict__ hyo, float *__restrict__ hzo, uint32_t *__restrict__ ido,
    Float4 *__restrict__ posm, float *__restrict__ posm2, const float *__restrict__ vxi,
    const float *__restrict__ vyi,
    const float *__restrict__ vzi, float *__restrict__ vxo, float *__restrict__ vyo,
    float *__restrict__ vzo)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < n;
    float px = 0.f, py = 0.f, pz = 0.f, pm = 0.f;   // (a massless partner pads an odd count)
    if (in) {
        const uint32_t s = perm[i];
        px = xi[s]; py = yi[s]; pz = zi[s]; pm = mi[s];
        xo[i] = px; yo[i] = py; zo[i] = pz; mo[i] = pm;
        hxo[i] = hxi[s]; hyo[i] = hyi[s]; hzo[i] = hzi[s];
        ido[i] = idi[s];
        posm[i] = Float4{px, py, pz, pm};
        if (ALL) {
            vxo[i] = vxi[s]; vyo[i] = vyi[s]; vzo[i] = vzi[s];
        }
    }
    // pair record (x0,x1,y0,y1,z0,z1,m0,m1): the two lanes of a pair trade values, the even one
    // stores the first half of the record, the odd one the second -- 16 bytes per lane, consecutive
    const float qx = __shfl_xor(px, 1, 64), qy = __shfl_xor(py, 1, 64);
    const float qz = __shfl_xor(pz, 1, 64), qm = __shfl_xor(pm, 1, 64);
    if ((i | 1) - 1 < n) {   // (the odd lane past an odd n completes the last record)
        const bool odd = i & 1;
        reinterpret_cast<Float4 *>(posm2)[i] = odd ? Float4{qz, pz, qm, pm} : Float4{px, qx, py, qy};
    }
}

// v = (v_half + 0.5*a*dt) * (1 - damp)
__global__ __launch_bounds__(256) void k_vel_kick(int n, float dt, float keep,
                                                  const float *__restrict__ vhx,
                                                  const float *__restrict__ vhy,
                                                  const float *__restrict__ vhz,
                                                  const float *__restrict__ ax,
                                                  const float *__restrict__ ay,
                                                  const float *__restrict__ az,
                                                  float *__restrict__ vx, float *__restrict__ vy,
                                                  float *__restrict__ vz)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    vx[i] = fmaf(0.5f * ax[i], dt, vhx[i]) * keep;
    vy[i] = fmaf(0.5f * ay[i], dt, vhy[i]) * keep;
    vz[i] = fmaf(0.5f * az[i], dt, vhz[i]) * keep;
}

// Kick-drift of the all-pairs twin (no Morton code; positions live in the packed record).
__global__ __launch_bounds__(256) void k_twin_kick_drift(int n, float dt, float domain,
                                                         Float4 *__restrict__ posm,
                                                         const float *__restrict__ vx,
                                                         const float *__restrict__ vy,
                                                         const float *__restrict__ vz,
                                                         const float *__restrict__ ax,
                                                         const float *__restrict__ ay,
                                                         const float *__restrict__ az,
                                                         float *__restrict__ vhx,
                                                         float *__restrict__ vhy,
                                                         float *__restrict__ vhz)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Float4 p = posm[i];
    float hx = fmaf(0.5f * ax[i], dt, vx[i]);
    float hy = fmaf(0.5f * ay[i], dt, vy[i]);
    float hz = fmaf(0.5f * az[i], dt, vz[i]);
    p.x = fmaf(hx, dt, p.x);
    p.y = fmaf(hy, dt, p.y);
    p.z = fmaf(hz, dt, p.z);
    reflect(p.x, hx, domain);
    reflect(p.y, hy, domain);
    reflect(p.z, hz, domain);
    posm[i] = p;
    vhx[i] = hx; vhy[i] = hy; vhz[i] = hz;
}

static inline int nblk(int n, int t) { return (n + t - 1) / t; }

// Boxes of `ranks` domains (6 words each, gathered) -> their union, into box[6].
__global__ void k_bbox_merge(const uint32_t *__restrict__ all, int ranks, uint32_t *box)
{
    const int k = threadIdx.x;
    if (k >= 6) return;
    uint32_t v = all[k];
    for (int r = 1; r < ranks; ++r) v = k < 3 ? min(v, all[6 * r + k]) : max(v, all[6 * r + k]);
    __hip_atomic_store(&box[k], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

hipError_t launch_bbox_merge(const uint32_t *all_dev, int ranks, uint32_t *box_dev, hipStream_t st)
{
    hipLaunchKernelGGL(k_bbox_merge, dim3(1), dim3(64), 0, st, all_dev, ranks, box_dev);
    return hipGetLastError();
}

hipError_t launch_twin_kick_drift(int n, float dt, float domain, Float4 *posm, const float *vx,
                                  const float *vy, const float *vz, const float *ax,
                                  const float *ay, const float *az, float *vhx, float *vhy,
                                  float *vhz, hipStream_t st)
{
    hipLaunchKernelGGL(k_twin_kick_drift, dim3(nblk(n, 256)), dim3(256), 0, st, n, dt, domain, posm,
                       vx, vy, vz, ax, ay, az, vhx, vhy, vhz);
    return hipGetLastError();
}

hipError_t launch_vel_kick_raw(int n, float dt, float keep, const float *vhx, const float *vhy,
                               const float *vhz, const float *ax, const float *ay, const float *az,
                               float *vx, float *vy, float *vz, hipStream_t st);

static hipError_t launch_fit_and_encode(Sim *s)
{
    const int c = s->cur;
    if (s->bbox_hook && s->bbox_hook(s->bbox_hook_ctx, reinterpret_cast<uint32_t *>(s->keybox + 5), s->stream) != 0)
        return hipErrorUnknown;
    hipLaunchKernelGGL(k_keyfit, dim3(1), dim3(1), 0, s->stream, s->keybox, s->p.domain_size);
    if (s->key_bytes == 4)
        hipLaunchKernelGGL(k_morton<uint32_t>, dim3(nblk(s->n, 256)), dim3(256), 0, s->stream, s->n,
                           s->p.domain_size, s->keybox, s->x[c], s->y[c], s->z[c], (uint32_t *) s->keys[0]);
    else
        hipLaunchKernelGGL(k_morton<uint64_t>, dim3(nblk(s->n, 256)), dim3(256), 0, s->stream, s->n,
                           s->p.domain_size, s->keybox, s->x[c], s->y[c], s->z[c], (uint64_t *) s->keys[0]);
    return hipGetLastError();
}

hipError_t launch_keyfit_only(Sim *s)
{
    hipLaunchKernelGGL(k_keyfit, dim3(1), dim3(1), 0, s->stream, s->keybox, s->p.domain_size);
    return hipGetLastError();
}

hipError_t launch_integrate_morton(Sim *s)
{
    const int c = s->cur;
    if (s->keybox) {   // fitted key domain: the cube is decided from the new positions
        hipLaunchKernelGGL(k_integrate_bbox, dim3(std::min(nblk(s->n, 256), BBOX_GRID)), dim3(256), 0, s->stream, s->n,
                           s->p.dt, s->p.domain_size, s->x[c], s->y[c], s->z[c], s->vx[c], s->vy[c],
                           s->vz[c], s->ax, s->ay, s->az, s->vhx[c], s->vhy[c], s->vhz[c],
                           reinterpret_cast<uint32_t *>(s->keybox + 5));
        return launch_fit_and_encode(s);
    }
    // the scratch of the sort and build that follow, cleared on the way when one thread per 16
    // bytes is there for it (else the sort's own memset does it)
    const size_t zb = scratch_zero_bytes(s);
    const int zero16 = (zb % 16 == 0 && zb / 16 <= (size_t) nblk(s->n, 256) * 256) ? (int) (zb / 16) : 0;
    if (s->key_bytes == 4)
        hipLaunchKernelGGL(k_integrate_morton<uint32_t>, dim3(nblk(s->n, 256)), dim3(256), 0,
                           s->stream, s->n, s->p.dt, s->p.domain_size, s->x[c], s->y[c], s->z[c],
                           s->vx[c], s->vy[c], s->vz[c], s->ax, s->ay, s->az, s->vhx[c],
                           s->vhy[c], s->vhz[c], (uint32_t *) s->keys[0], (uint4 *) s->zero_block, zero16);
    else
        hipLaunchKernelGGL(k_integrate_morton<uint64_t>, dim3(nblk(s->n, 256)), dim3(256), 0,
                           s->stream, s->n, s->p.dt, s->p.domain_size, s->x[c], s->y[c], s->z[c],
                           s->vx[c], s->vy[c], s->vz[c], s->ax, s->ay, s->az, s->vhx[c],
                           s->vhy[c], s->vhz[c], (uint64_t *) s->keys[0], (uint4 *) s->zero_block, zero16);
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) s->scratch_prezeroed = zero16 ? zb : 0;
    return e;
}

hipError_t launch_morton_only(Sim *s)
{
    const int c = s->cur;
    if (s->keybox) {
        hipLaunchKernelGGL(k_bbox_soa, dim3(std::min(nblk(s->n, 256), BBOX_GRID)), dim3(256), 0, s->stream, s->n, s->x[c],
                           s->y[c], s->z[c], reinterpret_cast<uint32_t *>(s->keybox + 5));
        return launch_fit_and_encode(s);
    }
    if (s->key_bytes == 4)
        hipLaunchKernelGGL(k_morton<uint32_t>, dim3(nblk(s->n, 256)), dim3(256), 0, s->stream,
                           s->n, s->p.domain_size, (const float *) nullptr, s->x[c], s->y[c], s->z[c],
                           (uint32_t *) s->keys[0]);
    else
        hipLaunchKernelGGL(k_morton<uint64_t>, dim3(nblk(s->n, 256)), dim3(256), 0, s->stream,
                           s->n, s->p.domain_size, (const float *) nullptr, s->x[c], s->y[c], s->z[c],
                           (uint64_t *) s->keys[0]);
    return hipGetLastError();
}

hipError_t launch_permute(Sim *s, bool all_fields)
{
    const int c = s->cur, o = c ^ 1;
    const uint32_t *perm = s->vals[s->sorted_buf];
    if (all_fields)
        hipLaunchKernelGGL(k_permute<true>, dim3(nblk(s->n, 256)), dim3(256), 0, s->stream, s->n,
                           perm, s->x[c], s->y[c], s->z[c], s->m[c], s->vhx[c], s->vhy[c],
                           s->vhz[c], s->id[c], s->x[o], s->y[o], s->z[o], s->m[o], s->vhx[o],
                           s->vhy[o], s->vhz[o], s->id[o], s->posm, s->posm2, s->vx[c], s->vy[c], s->vz[c],
                           s->vx[o], s->vy[o], s->vz[o]);
    else
        hipLaunchKernelGGL(k_permute<false>, dim3(nblk(s->n, 256)), dim3(256), 0, s->stream, s->n,
                           perm, s->x[c], s->y[c], s->z[c], s->m[c], s->vhx[c], s->vhy[c],
                           s->vhz[c], s->id[c], s->x[o], s->y[o], s->z[o], s->m[o], s->vhx[o],
                           s->vhy[o], s->vhz[o], s->id[o], s->posm, s->posm2, nullptr, nullptr, nullptr,
                           nullptr, nullptr, nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    // Every pair flips.  When velocity was not gathered its new buffer holds stale values;
    // the velocity kick that ends the step overwrites all of them.
    s->cur = o;
    return hipSuccess;
}

hipError_t launch_vel_kick_raw(int n, float dt, float keep, const float *vhx, const float *vhy,
                               const float *vhz, const float *ax, const float *ay, const float *az,
                               float *vx, float *vy, float *vz, hipStream_t st)
{
    hipLaunchKernelGGL(k_vel_kick, dim3(nblk(n, 256)), dim3(256), 0, st, n, dt, keep, vhx, vhy, vhz,
                       ax, ay, az, vx, vy, vz);
    return hipGetLastError();
}

hipError_t launch_vel_kick(Sim *s)
{
    const int c = s->cur;
    hipLaunchKernelGGL(k_vel_kick, dim3(nblk(s->n, 256)), dim3(256), 0, s->stream, s->n, s->p.dt,
                       1.0f - s->p.velocity_dampening, s->vhx[c], s->vhy[c], s->vhz[c], s->ax,
                       s->ay, s->az, s->vx[c], s->vy[c], s->vz[c]);
    return hipGetLastError();
}

// The reference's GL constructor makes seven viewer-owned buffers the storage of the particle
// positions and of the octree nodes' centres of mass and sizes (src/cuda/simulation.cu:39-51,
// 138-158).  Here the caller's device buffers receive a copy after every build instead.
__global__ __launch_bounds__(256) void k_export_view(int n, const Counts *counts,
                                                     const Float4 *__restrict__ posm,
                                                     const Float4 *__restrict__ nodeA,
                                                     const NodeB *__restrict__ nodeB, float *px,
                                                     float *py, float *pz, float *cx, float *cy,
                                                     float *cz, float *sz)
{
    const int n_oct = counts->n_oct;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < max(n, n_oct); i += gridDim.x * 256) {
        if (i < n) {
            const Float4 p = posm[i];
            if (px) px[i] = p.x;
            if (py) py[i] = p.y;
            if (pz) pz[i] = p.z;
        }
        if (i < n_oct) {
            const Float4 a = nodeA[i];
            if (cx) cx[i] = a.x;
            if (cy) cy[i] = a.y;
            if (cz) cz[i] = a.z;
            if (sz) sz[i] = nodeB[i].size;
        }
    }
}

hipError_t launch_export_view(Sim *s)
{
    if (!s->has_ext || s->n < 1) return hipSuccess;
    hipLaunchKernelGGL(k_export_view, dim3(std::min(nblk(s->n, 256), 2048)), dim3(256), 0, s->stream,
                       s->n, s->counts, s->posm, s->nodeA, s->nodeB, s->ext[0], s->ext[1], s->ext[2],
                       s->ext[3], s->ext[4], s->ext[5], s->ext[6]);
    return hipGetLastError();
}

}  // namespace nb

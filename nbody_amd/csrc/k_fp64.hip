// k_fp64.hip -- the floating-point stages of a double-precision handle.
//
// The reference instantiates every class for float and for double
// (src/cuda/simulation.cu:255-256, points.cu:228, octree.cu:367, barnes_hut.cu:677,
// validator.cu:381, physics_common.cuh:103) and chooses one at build time; its
// "all-pairs (double)" energy curves come from the double build.  Here precision is a
// property of the handle (NBODY_FLAG_FP64).  Keys, the sort and every tree array are
// integer work and are shared with the float path; this file holds what touches
// coordinates: kick-drift + Morton encode, the particle permute, centres of mass, the
// group traversal, the velocity kick and the direct sum, all in double.
//
// MI355X executes v_fma_f64 at half the FP32 rate and has no packed double ops, so these
// kernels are plain one-body-per-lane code: sources staged through LDS and broadcast.
#include <algorithm>

#include "com_kernel.hpp"
#include "dev_utils.hpp"
#include "nbody_internal.hpp"

namespace nb {

namespace {

// q = (uint)(p / domain * (2^B - 1)) with the division and the product in double
// (src/cuda/points.cu:28-30 with T = double; domain_size stays a float parameter).
template <typename KeyT>
__device__ __forceinline__ KeyT morton_code_d(double px, double py, double pz, double domain)
{
    if constexpr (sizeof(KeyT) == 4) {
        const double s = 1023.0;
        const uint32_t qx = (uint32_t) (uint64_t) ((px / domain) * s);
        const uint32_t qy = (uint32_t) (uint64_t) ((py / domain) * s);
        const uint32_t qz = (uint32_t) (uint64_t) ((pz / domain) * s);
        return spread3_10(qx) * 4u + spread3_10(qy) * 2u + spread3_10(qz);
    } else {
        const double s = 2097151.0;
        const uint64_t qx = (uint64_t) ((px / domain) * s);
        const uint64_t qy = (uint64_t) ((py / domain) * s);
        const uint64_t qz = (uint64_t) ((pz / domain) * s);
        return spread3_21(qx) * 4ull + spread3_21(qy) * 2ull + spread3_21(qz);
    }
}

__device__ __forceinline__ void reflect_d(double &p, double &v, double domain)
{
    if (p < 0.0 || p > domain) {
        v *= -1.0;
        p = fmax(0.0, fmin(domain, p));
    }
}

// leapfrog_integrate_pos + impose_boundary_conditions (src/cuda/physics_common.cu:69-147)
// followed by the Morton code of the new position.
template <typename KeyT>
__global__ __launch_bounds__(256) void k64_integrate_morton(
    int n, double dt, double domain, double *__restrict__ x, double *__restrict__ y,
    double *__restrict__ z, const double *__restrict__ vx, const double *__restrict__ vy,
    const double *__restrict__ vz, const double *__restrict__ ax, const double *__restrict__ ay,
    const double *__restrict__ az, double *__restrict__ vhx, double *__restrict__ vhy,
    double *__restrict__ vhz, KeyT *__restrict__ codes)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double hx = fma(0.5 * ax[i], dt, vx[i]);
    double hy = fma(0.5 * ay[i], dt, vy[i]);
    double hz = fma(0.5 * az[i], dt, vz[i]);
    double px = fma(hx, dt, x[i]);
    double py = fma(hy, dt, y[i]);
    double pz = fma(hz, dt, z[i]);
    reflect_d(px, hx, domain);
    reflect_d(py, hy, domain);
    reflect_d(pz, hz, domain);
    x[i] = px; y[i] = py; z[i] = pz;
    vhx[i] = hx; vhy[i] = hy; vhz[i] = hz;
    codes[i] = morton_code_d<KeyT>(px, py, pz, domain);
}

template <typename KeyT>
__global__ __launch_bounds__(256) void k64_morton(int n, double domain, const double *__restrict__ x,
                                                  const double *__restrict__ y,
                                                  const double *__restrict__ z,
                                                  KeyT *__restrict__ codes)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    codes[i] = morton_code_d<KeyT>(x[i], y[i], z[i], domain);
}

struct Perm64 {
    const double *xi, *yi, *zi, *mi, *hxi, *hyi, *hzi, *vxi, *vyi, *vzi;
    double *xo, *yo, *zo, *mo, *hxo, *hyo, *hzo, *vxo, *vyo, *vzo;
    const uint32_t *idi;
    uint32_t *ido;
    Double4 *posm;
};

// Points::sort_by_codes / BarnesHut::sort_bodies gathers (src/cuda/points.cu:86-113,
// src/cuda/barnes_hut.cu:577-587) in one pass; ALL = velocities too (after a spawn).
template <bool ALL>
__global__ __launch_bounds__(256) void k64_permute(int n, const uint32_t *__restrict__ perm, Perm64 a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = perm[i];
    const double px = a.xi[s], py = a.yi[s], pz = a.zi[s], pm = a.mi[s];
    a.xo[i] = px; a.yo[i] = py; a.zo[i] = pz; a.mo[i] = pm;
    a.hxo[i] = a.hxi[s]; a.hyo[i] = a.hyi[s]; a.hzo[i] = a.hzi[s];
    a.ido[i] = a.idi[s];
    a.posm[i] = Double4{px, py, pz, pm};
    if (ALL) {
        a.vxo[i] = a.vxi[s]; a.vyo[i] = a.vyi[s]; a.vzo[i] = a.vzi[s];
    }
}

// leapfrog_integrate_vel (src/cuda/physics_common.cu:120-147): v = (v_half + 0.5 a dt)(1 - damp)
__global__ __launch_bounds__(256) void k64_vel_kick(int n, double dt, double keep,
                                                    const double *__restrict__ vhx,
                                                    const double *__restrict__ vhy,
                                                    const double *__restrict__ vhz,
                                                    const double *__restrict__ ax,
                                                    const double *__restrict__ ay,
                                                    const double *__restrict__ az,
                                                    double *__restrict__ vx, double *__restrict__ vy,
                                                    double *__restrict__ vz)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    vx[i] = fma(0.5 * ax[i], dt, vhx[i]) * keep;
    vy[i] = fma(0.5 * ay[i], dt, vhy[i]) * keep;
    vz[i] = fma(0.5 * az[i], dt, vhz[i]) * keep;
}

// Kick-drift of the validator's all-pairs twin (positions in the packed record).
__global__ __launch_bounds__(256) void k64_twin_kick_drift(
    int n, double dt, double domain, Double4 *__restrict__ posm, const double *__restrict__ vx,
    const double *__restrict__ vy, const double *__restrict__ vz, const double *__restrict__ ax,
    const double *__restrict__ ay, const double *__restrict__ az, double *__restrict__ vhx,
    double *__restrict__ vhy, double *__restrict__ vhz)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Double4 p = posm[i];
    double hx = fma(0.5 * ax[i], dt, vx[i]);
    double hy = fma(0.5 * ay[i], dt, vy[i]);
    double hz = fma(0.5 * az[i], dt, vz[i]);
    p.x = fma(hx, dt, p.x);
    p.y = fma(hy, dt, p.y);
    p.z = fma(hz, dt, p.z);
    reflect_d(p.x, hx, domain);
    reflect_d(p.y, hy, domain);
    reflect_d(p.z, hz, domain);
    posm[i] = p;
    vhx[i] = hx; vhy[i] = hy; vhz[i] = hz;
}

// 1/sqrt(t), t > 0: v_rsq_f64 seeds one third-order step.  With e = 1 - t y^2,
// 1/sqrt(t) = y (1 - e)^(-1/2) = y (1 + e/2 + 3 e^2/8 + O(e^3)): the seed's relative error is
// cubed, which is below half an ulp of a double for a seed good to 2^-18 or better.
__device__ __forceinline__ double rsq64(double t)
{
    const double y = __builtin_amdgcn_rsq(t);
    const double e = fma(-t * y, y, 1.0);
    return fma(y, e * fma(0.375, e, 0.5), y);
}

// ---------------------------------------------------------------------------
// Group traversal (src/cuda/barnes_hut.cu:199-537) in double.  One wave64 per group of G
// Morton-consecutive bodies, lane = body.  The wave keeps a stack of node ids in LDS and
// pops up to 64 per round, one per lane:
//   * a node without children contributes its particles (always direct, self included);
//   * otherwise the lane computes d2min = min over the group's bodies of |x_b - com|^2
//     (initial value 3 domain^2) and accepts (com, mass) iff size^2 / d2min < theta^2,
//     else pushes the children.
// The decisions use the operation sequence of the host checker, so each group's source
// list is identical to it.  Accepted monopoles and leaf particles go through an LDS ring
// that every lane reads back (broadcast) 64 sources at a time.
// A massless ring entry contributes exactly 0 (softening > 0 is enforced at create time).
// Stack discipline: a round pops only as many nodes as can push 8 children each without
// passing capacity - reserve; when not even one fits the round pops one node and may use
// the reserve, which makes those rounds depth-first: the entries above the mark are then
// at most 7 siblings per level of a chain of strictly increasing depth, and the reserve
// of 7 * 24 covers every level a 63-bit key has.
// ---------------------------------------------------------------------------
constexpr int T64_WAVES = 4;
constexpr int T64_STACK = 1024;
constexpr int T64_RESERVE = 7 * 24;
constexpr int T64_RING = 128;
constexpr int T64_DIRECT = 1 << 30;   // stack entry flag: take the node without a test

struct alignas(16) T64Lds {
    Double4 ring[T64_RING];
    double bx[64], by[64], bz[64];
    int stack[T64_STACK];
    int lfirst[64];
    int lscan[64];
};

template <int G, bool POT>
__global__ __launch_bounds__(64 * T64_WAVES) void k64_traverse(
    int n, int g0, int ngroups, const Double4 *__restrict__ posm, const Double4 *__restrict__ nodeA,
    const NodeB *__restrict__ nodeB, double theta2, double eps2, double gravity, double dinit,
    double *__restrict__ ax, double *__restrict__ ay, double *__restrict__ az,
    double *__restrict__ pot, uint32_t *__restrict__ list_len, uint32_t *err, int stack_cap,
    const int32_t *__restrict__ roots, int nroots, int acc_flag)
{
    // Stack entries: node id, with T64_DIRECT set when the node is to be taken without a test:
    // a leaf its parent marks in the child mask.  In the handle's own tree every leaf child is
    // marked (a leaf is always evaluated directly, as in the reference); in the pruned copies of
    // other domains' trees (nroots > 0) only the children that pass for the whole domain are,
    // and an unmarked leaf is tested like an inner node.  acc_flag: bit 0 = add to ax/ay/az,
    // bit 2 = leave the own root out.
    __shared__ T64Lds lds[T64_WAVES];
    const int lane = (int) threadIdx.x & 63, w = (int) threadIdx.x >> 6;
    if ((int) blockIdx.x * T64_WAVES + w >= ngroups) return;   // whole wave; no workgroup barrier here
    const int g = g0 + (int) blockIdx.x * T64_WAVES + w;
    T64Lds &L = lds[w];
    const int b0 = g * G;
    const int nb = min(G, n - b0);
    // G = 32: the wave holds the group twice; each half evaluates every other source of the
    // ring and the two partial sums are added at the end
    const int bl = G == 32 ? (lane & 31) : lane;
    const int half = G == 32 ? (lane >> 5) : 0;
    const bool active = bl < nb;
    const Double4 me = posm[active ? b0 + bl : b0];
    L.bx[lane] = me.x; L.by[lane] = me.y; L.bz[lane] = me.z;
    int sp = 0;
    if (!(acc_flag & 4)) {
        if (lane == 0) L.stack[0] = (nodeB[0].meta & NODE_LEAF) ? T64_DIRECT : 0;
        sp = 1;
    }
    for (int r0 = 0; r0 < nroots; r0 += 64) {   // nroots <= LET_MAX_BOXES < stack size
        const int k = min(64, nroots - r0);
        if (lane < k) L.stack[sp + lane] = roots[r0 + lane];
        sp += k;
    }
    __builtin_amdgcn_wave_barrier();

    double fx = 0.0, fy = 0.0, fz = 0.0, ps = 0.0;
    int head = 0, cnt = 0;   // ring entries [head, head + cnt) mod T64_RING
    uint32_t total = 0;

    auto force = [&](const Double4 s) {
        const double dx = s.x - me.x, dy = s.y - me.y, dz = s.z - me.z;
        const double t = fma(dz, dz, fma(dy, dy, fma(dx, dx, eps2)));
        if (POT) {
            ps = fma(s.w, rsq64(t), ps);
        } else {
            const double inv = s.w * rsq64(t * t * t);
            fx = fma(dx, inv, fx);
            fy = fma(dy, inv, fy);
            fz = fma(dz, inv, fz);
        }
    };
    // Every lane evaluates the ring's first k entries.  head only ever advances by 64 (the
    // one shorter flush ends the kernel), so it is 0 or 64 and the k <= 64 entries are
    // contiguous: the loop reads LDS at constant offsets from one base.
    auto flush = [&](int k) {
        __builtin_amdgcn_wave_barrier();
        Double4 *r = &L.ring[head];
        if constexpr (G == 32) {
            if (k == 64) {
#pragma unroll 8
                for (int j = 0; j < 32; ++j) force(r[2 * j + half]);
            } else {   // the last, shorter flush: a massless entry evens out an odd count
                if ((k & 1) && lane == 0) r[k] = Double4{0.0, 0.0, 0.0, 0.0};
                __builtin_amdgcn_wave_barrier();
                for (int j = 0; j < (k + 1) / 2; ++j) force(r[2 * j + half]);
            }
        } else if (k == 64) {
#pragma unroll 8
            for (int j = 0; j < 64; ++j) force(r[j]);
        } else {
            for (int j = 0; j < k; ++j) force(r[j]);
        }
        head = (head + k) & (T64_RING - 1);
        cnt -= k;
        __builtin_amdgcn_wave_barrier();
    };
    auto append = [&](bool want, const Double4 v) {
        const unsigned long long wm = __ballot(want);
        if (wm == 0ull) return;
        if (want) L.ring[(head + cnt + mbcnt(wm)) & (T64_RING - 1)] = v;
        cnt += (int) __popcll(wm);
        if (cnt >= 64) flush(64);
    };

    int rounds = 0;
    while (sp > 0) {
        if (++rounds > (1 << 22) || sp > stack_cap - 8) {   // cannot happen: see the discipline above
            if (lane == 0) atomicOr(err, DEV_ERR_STACK);
            break;
        }
        const int fit = (stack_cap - T64_RESERVE - sp) / 7;
        const int p = fit >= 1 ? min(min(64, sp), fit) : 1;
        const int entry = lane < p ? L.stack[sp - p + lane] : -1;
        const int node = entry < 0 ? -1 : (entry & ~T64_DIRECT);
        sp -= p;
        __builtin_amdgcn_wave_barrier();

        bool is_leaf = false, accept = false, open = false;
        NodeB B{0.f, 0, 0, 0u};
        Double4 A{0.0, 0.0, 0.0, 0.0};
        if (node >= 0) {
            B = nodeB[node];
            is_leaf = (B.meta & NODE_LEAF) != 0u;
            if (!(entry & T64_DIRECT)) {
                A = nodeA[node];
                double dmin = dinit;
                for (int j = 0; j < nb; ++j) {
                    const double dx = L.bx[j] - A.x, dy = L.by[j] - A.y, dz = L.bz[j] - A.z;
                    const double d = fma(dz, dz, fma(dy, dy, dx * dx));
                    if (d <= dmin) dmin = d;
                }
                const double sz = (double) B.size;
                accept = sz * sz / dmin < theta2;
                open = !accept && !is_leaf;
                is_leaf = is_leaf && !accept;   // a tested leaf that fails is taken particle by particle
            }
        }
        total += (uint32_t) __popcll(__ballot(accept));
        append(accept, A);

        {   // children of opened nodes back on the stack
            const uint32_t nc = open ? (uint32_t) B.count : 0u;
            const uint32_t incl = wave_incl_scan_u32(nc);
            const int tot = (int) __shfl(incl, 63, 64);
            const int at = sp + (int) (incl - nc);
            for (uint32_t c = 0; c < nc; ++c)
                L.stack[at + (int) c] = (B.first + (int) c) | (((B.meta >> c) & 1u) ? T64_DIRECT : 0);
            sp += tot;
        }
        {   // particles of the popped leaves, 64 at a time in leaf order
            const uint32_t lc = is_leaf ? (uint32_t) B.count : 0u;
            const uint32_t incl = wave_incl_scan_u32(lc);
            const int ltot = (int) __shfl(incl, 63, 64);
            if (ltot > 0) {
                L.lfirst[lane] = B.first;
                L.lscan[lane] = (int) (incl - lc);
                __builtin_amdgcn_wave_barrier();
                for (int k0 = 0; k0 < ltot; k0 += 64) {
                    const int k = k0 + lane;
                    const bool want = k < ltot;
                    int i = 0;   // the last lane whose first particle index is <= k owns particle k
#pragma unroll
                    for (int step = 32; step > 0; step >>= 1)
                        if (L.lscan[i + step] <= k) i += step;
                    const Double4 v = posm[want ? L.lfirst[i] + (k - L.lscan[i]) : 0];
                    append(want, v);
                }
                total += (uint32_t) ltot;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (cnt > 0) flush(cnt);
    if constexpr (G == 32) {
        fx += __shfl_xor(fx, 32, 64); fy += __shfl_xor(fy, 32, 64); fz += __shfl_xor(fz, 32, 64);
        ps += __shfl_xor(ps, 32, 64);
    }

    if (active && half == 0) {
        const int b = b0 + bl;
        if (POT) {
            pot[b] = ps - me.w * rsq64(eps2);   // the body's own term, met exactly once
        } else if (acc_flag & 1) {
            ax[b] += gravity * fx; ay[b] += gravity * fy; az[b] += gravity * fz;
        } else {
            ax[b] = gravity * fx; ay[b] = gravity * fy; az[b] = gravity * fz;
        }
    }
    if (lane == 0 && !POT) list_len[g] = total;
}

// ---------------------------------------------------------------------------
// Direct sum (compute_ap_forces, src/cuda/validator.cu:134-174) in double: 256 targets per
// workgroup, sources staged through LDS 256 at a time.  With `pot` the kernel also returns
// pot_i = sum_j m_j / sqrt(d^2 + eps^2); self_pairs removes the j = i term again (targets
// and sources are the same array), exact because both sides evaluate m_i * rsq(eps^2).
// ---------------------------------------------------------------------------
template <bool POT>
__global__ __launch_bounds__(256) void k64_direct(const Double4 *__restrict__ targets, int nt,
                                                  const Double4 *__restrict__ sources, int ns,
                                                  double eps2, double gravity,
                                                  double *__restrict__ ax, double *__restrict__ ay,
                                                  double *__restrict__ az, double *__restrict__ pot,
                                                  int self_pairs)
{
    __shared__ Double4 s_src[256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const Double4 me = targets[i < nt ? i : 0];
    double fx = 0.0, fy = 0.0, fz = 0.0, ps = 0.0;
    for (int t0 = 0; t0 < ns; t0 += 256) {
        const int c = min(256, ns - t0);
        const int t = (int) threadIdx.x;
        s_src[t] = t < c ? sources[t0 + t] : Double4{0.0, 0.0, 0.0, 0.0};   // massless pad
        __syncthreads();
#pragma unroll 4
        for (int j = 0; j < c; ++j) {
            const Double4 s = s_src[j];
            const double dx = s.x - me.x, dy = s.y - me.y, dz = s.z - me.z;
            const double tt = fma(dz, dz, fma(dy, dy, fma(dx, dx, eps2)));
            double inv;
            if (POT) {
                const double r = rsq64(tt);
                const double mr = s.w * r;
                ps += mr;
                inv = mr * r * r;
            } else {
                inv = s.w * rsq64(tt * tt * tt);
            }
            fx = fma(dx, inv, fx);
            fy = fma(dy, inv, fy);
            fz = fma(dz, inv, fz);
        }
        __syncthreads();
    }
    if (i < nt) {
        ax[i] = gravity * fx; ay[i] = gravity * fy; az[i] = gravity * fz;
        if (POT) pot[i] = self_pairs ? ps - me.w * rsq64(eps2) : ps;
    }
}

// Viewer buffers of Simulation<double>(Params&, GLuint[7]) (src/cuda/simulation.cu:39-51,
// 138-158): positions and the nodes' centres of mass and sizes, as doubles.
__global__ __launch_bounds__(256) void k64_export_view(int n, const Counts *counts,
                                                       const Double4 *__restrict__ posm,
                                                       const Double4 *__restrict__ nodeA,
                                                       const NodeB *__restrict__ nodeB, double *px,
                                                       double *py, double *pz, double *cx,
                                                       double *cy, double *cz, double *sz)
{
    const int n_oct = counts->n_oct;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < max(n, n_oct); i += gridDim.x * 256) {
        if (i < n) {
            const Double4 p = posm[i];
            if (px) px[i] = p.x;
            if (py) py[i] = p.y;
            if (pz) pz[i] = p.z;
        }
        if (i < n_oct) {
            const Double4 a = nodeA[i];
            if (cx) cx[i] = a.x;
            if (cy) cy[i] = a.y;
            if (cz) cz[i] = a.z;
            if (sz) sz[i] = (double) nodeB[i].size;
        }
    }
}

inline int nblk(int n, int t) { return (n + t - 1) / t; }

}  // namespace

hipError_t launch64_integrate_morton(Sim *s)
{
    State64 &d = s->d;
    const int c = s->cur;
    if (s->key_bytes == 4)
        hipLaunchKernelGGL(k64_integrate_morton<uint32_t>, dim3(nblk(s->n, 256)), dim3(256), 0,
                           s->stream, s->n, (double) s->p.dt, (double) s->p.domain_size, d.x[c],
                           d.y[c], d.z[c], d.vx[c], d.vy[c], d.vz[c], d.ax, d.ay, d.az, d.vhx[c],
                           d.vhy[c], d.vhz[c], (uint32_t *) s->keys[0]);
    else
        hipLaunchKernelGGL(k64_integrate_morton<uint64_t>, dim3(nblk(s->n, 256)), dim3(256), 0,
                           s->stream, s->n, (double) s->p.dt, (double) s->p.domain_size, d.x[c],
                           d.y[c], d.z[c], d.vx[c], d.vy[c], d.vz[c], d.ax, d.ay, d.az, d.vhx[c],
                           d.vhy[c], d.vhz[c], (uint64_t *) s->keys[0]);
    return hipGetLastError();
}

hipError_t launch64_morton_only(Sim *s)
{
    State64 &d = s->d;
    const int c = s->cur;
    if (s->key_bytes == 4)
        hipLaunchKernelGGL(k64_morton<uint32_t>, dim3(nblk(s->n, 256)), dim3(256), 0, s->stream,
                           s->n, (double) s->p.domain_size, d.x[c], d.y[c], d.z[c],
                           (uint32_t *) s->keys[0]);
    else
        hipLaunchKernelGGL(k64_morton<uint64_t>, dim3(nblk(s->n, 256)), dim3(256), 0, s->stream,
                           s->n, (double) s->p.domain_size, d.x[c], d.y[c], d.z[c],
                           (uint64_t *) s->keys[0]);
    return hipGetLastError();
}

hipError_t launch64_permute(Sim *s, bool all_fields)
{
    State64 &d = s->d;
    const int c = s->cur, o = c ^ 1;
    Perm64 a{d.x[c], d.y[c], d.z[c], d.m[c], d.vhx[c], d.vhy[c], d.vhz[c], d.vx[c], d.vy[c], d.vz[c],
             d.x[o], d.y[o], d.z[o], d.m[o], d.vhx[o], d.vhy[o], d.vhz[o], d.vx[o], d.vy[o], d.vz[o],
             s->id[c], s->id[o], d.posm};
    const uint32_t *perm = s->vals[s->sorted_buf];
    if (all_fields)
        hipLaunchKernelGGL(k64_permute<true>, dim3(nblk(s->n, 256)), dim3(256), 0, s->stream, s->n, perm, a);
    else
        hipLaunchKernelGGL(k64_permute<false>, dim3(nblk(s->n, 256)), dim3(256), 0, s->stream, s->n, perm, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    // every pair flips; velocities not gathered are overwritten by the kick that ends the step
    s->cur = o;
    return hipSuccess;
}

hipError_t launch64_com(Sim *s)
{
    const int grid = std::min((2 * s->n + 255) / 256, COM_GRID);
    hipLaunchKernelGGL((k_com_t<double, Double4>), dim3(grid), dim3(256), 0, s->stream, s->counts, s->o_num_children,
                       s->o_first_child, s->o_pb, s->o_pe, s->o_parent, s->d.posm, s->d.wsum,
                       s->d.nodeA, s->com_counter, reinterpret_cast<const uint4 *>(s->depth_key),
                       s->tickets + COM_ROOTS_TICKET, &s->counts->err);
    return hipGetLastError();
}

// what: 0 = the own tree, 1 = own tree and imported copies in one walk, 2 = the copies alone,
// added to the accelerations (see launch_traverse_forest)
static hipError_t traverse64_impl(Sim *s, bool potential, int what)
{
    State64 &d = s->d;
    const int all = (s->n + s->group - 1) / s->group;
    // the step's traversal covers the range set by nbody_set_traversal_range; the potential
    // always covers every group
    const int g0 = (potential || s->trav_g1 < 0) ? 0 : s->trav_g0;
    const int g1 = (potential || s->trav_g1 < 0) ? all : std::min(s->trav_g1, all);
    const int ngroups = g1 - g0;
    if (ngroups < 1) return hipSuccess;
    const int grid = (ngroups + T64_WAVES - 1) / T64_WAVES;
    const double theta = (double) s->p.theta, eps = (double) s->p.softening_factor;
    const double dom = (double) s->p.domain_size;
    // test aid: 32 usable entries make single-node depth-first rounds the common case
    const int stack_cap = (s->p.flags & NBODY_FLAG_DEBUG_SMALL_STACK) ? T64_RESERVE + 32 : T64_STACK;
    const int32_t *roots = what ? s->imp_roots : nullptr;
    const int nroots = what ? s->imp_nroots : 0;
    const int acc_flag = what == 2 ? 5 : 0;
#define NB_T64(GG, PP)                                                                           \
    hipLaunchKernelGGL((k64_traverse<GG, PP>), dim3(grid), dim3(64 * T64_WAVES), 0, s->stream,    \
                       s->n, g0, ngroups, d.posm, d.nodeA, s->nodeB, theta * theta, eps * eps,    \
                       (double) s->p.gravity, dom * dom * 3.0, d.ax, d.ay, d.az, d.pot,           \
                       s->list_len, &s->counts->err, stack_cap, roots, nroots, acc_flag)
    if (s->group == 64) { if (potential) NB_T64(64, true); else NB_T64(64, false); }
    else                { if (potential) NB_T64(32, true); else NB_T64(32, false); }
#undef NB_T64
    return hipGetLastError();
}

hipError_t launch64_traverse(Sim *s, bool potential) { return traverse64_impl(s, potential, 0); }

hipError_t launch64_traverse_forest(Sim *s, int what) { return traverse64_impl(s, false, what); }

hipError_t launch64_vel_kick(Sim *s)
{
    State64 &d = s->d;
    const int c = s->cur;
    hipLaunchKernelGGL(k64_vel_kick, dim3(nblk(s->n, 256)), dim3(256), 0, s->stream, s->n,
                       (double) s->p.dt, 1.0 - (double) s->p.velocity_dampening, d.vhx[c], d.vhy[c],
                       d.vhz[c], d.ax, d.ay, d.az, d.vx[c], d.vy[c], d.vz[c]);
    return hipGetLastError();
}

hipError_t launch64_export_view(Sim *s)
{
    if (!s->has_ext || s->n < 1) return hipSuccess;
    State64 &d = s->d;
    hipLaunchKernelGGL(k64_export_view, dim3(std::min(nblk(s->n, 256), 2048)), dim3(256), 0, s->stream,
                       s->n, s->counts, d.posm, d.nodeA, s->nodeB, d.ext[0], d.ext[1], d.ext[2],
                       d.ext[3], d.ext[4], d.ext[5], d.ext[6]);
    return hipGetLastError();
}

hipError_t launch64_direct(const Double4 *targets, int nt, const Double4 *sources, int ns,
                           double eps2, double gravity, double *ax, double *ay, double *az,
                           double *pot, int self_pairs, hipStream_t st)
{
    if (nt <= 0) return hipSuccess;
    if (pot)
        hipLaunchKernelGGL(k64_direct<true>, dim3(nblk(nt, 256)), dim3(256), 0, st, targets, nt,
                           sources, ns, eps2, gravity, ax, ay, az, pot, self_pairs);
    else
        hipLaunchKernelGGL(k64_direct<false>, dim3(nblk(nt, 256)), dim3(256), 0, st, targets, nt,
                           sources, ns, eps2, gravity, ax, ay, az, pot, 0);
    return hipGetLastError();
}

hipError_t launch64_twin_kick_drift(int n, double dt, double domain, Double4 *posm,
                                    double *const v[3], double *const a[3], double *const vh[3],
                                    hipStream_t st)
{
    hipLaunchKernelGGL(k64_twin_kick_drift, dim3(nblk(n, 256)), dim3(256), 0, st, n, dt, domain,
                       posm, v[0], v[1], v[2], a[0], a[1], a[2], vh[0], vh[1], vh[2]);
    return hipGetLastError();
}

hipError_t launch64_vel_kick_raw(int n, double dt, double keep, double *const vh[3],
                                 double *const a[3], double *const v[3], hipStream_t st)
{
    hipLaunchKernelGGL(k64_vel_kick, dim3(nblk(n, 256)), dim3(256), 0, st, n, dt, keep, vh[0], vh[1],
                       vh[2], a[0], a[1], a[2], v[0], v[1], v[2]);
    return hipGetLastError();
}

}  // namespace nb

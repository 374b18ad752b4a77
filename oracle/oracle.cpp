// oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Host (CPU) restatement of the Barnes-Hut step of lechebs/nbody, used only as
// the checker for the HIP library (tests/, __graft_entry__.smoke(), and the
// cpu_baseline leg of bench.py).  The product library never links, loads or
// calls anything in this file.
//
// PARITY UNPINNED BY REFERENCE FIXTURES: the reference ships no tests, golden
// vectors or known-answer files for this path and its CUDA sources cannot be
// built in this environment, so this restatement is pinned only by (a) a
// line-by-line reading of the cited sources, (b) hand-checkable cases in
// tests/test_oracle.py and (c) the published accuracy/energy curves.
//
// Every function cites the reference file:line it follows (paths relative to
// the reference checkout).  Deliberate, documented deviations:
//   * bits per axis B is a parameter (10 -> 30-bit keys, 21 -> the reference's
//     63-bit keys); lcp(i,j) = clz64(ci^cj) - (64-3B), which is the reference's
//     `__clzll(x) - 1` at B=21 (src/cuda/btree.cu:17-21).
//   * the sort is STABLE (ties by original index); the reference's
//     cub::DeviceMergeSort::SortPairs (src/cuda/points.cu:79-84) leaves tie
//     order unspecified.
//   * U <= max_num_codes_per_leaf (reference: non-terminating loop in
//     src/cuda/btree.cu:72-76) is defined as a single root leaf.
//   * floating point contractions are written out with fmaf() so that the GPU
//     and this file execute the same IEEE operations.
//   * centre-of-mass summation order is fixed (sequential over a leaf's points,
//     sequential over an internal node's children) so it can be reproduced
//     bit-for-bit.

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>
#include <parallel/algorithm>   // __gnu_parallel::stable_sort (libstdc++ parallel mode, OpenMP)

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

using key_t_ = uint64_t;

inline int clz64(uint64_t v) { return v ? __builtin_clzll(v) : 64; }

struct Keys {
    const key_t_ *c;
    int n;
    int kb;  // key bits = 3 * bits-per-axis
};

// src/cuda/btree.cu:17-21 (lcp_safe): `__clzll(a ^ b) - 1` for 63-bit keys.
inline int lcp_safe(const Keys &k, int i, int j)
{
    return clz64(k.c[i] ^ k.c[j]) - (64 - k.kb);
}

// src/cuda/btree.cu:23-30 (lcp): out-of-range neighbour -> -3.
inline int lcp(const Keys &k, int i, int j)
{
    if (j < 0 || j > k.n - 1) return -3;
    return lcp_safe(k, i, j);
}

inline int sign(int x) { return (x > 0) - (x < 0); }

// src/cuda/points.cu:3-12 (expand_bits), written as the definition it
// implements: bit k of u goes to bit 3k.
inline key_t_ spread3(key_t_ u, int bits)
{
    key_t_ r = 0;
    for (int k = 0; k < bits; ++k) r |= ((u >> k) & 1ull) << (3 * k);
    return r;
}

// src/cuda/btree.cu:60-88 (search_lr).
int search_lr(const Keys &k, const int *leaf_depth, int dir, int idx, int depth_delta)
{
    int l = 0;
    int l_max = 2;
    int target_depth = leaf_depth[idx] - depth_delta;
    while (lcp(k, idx, idx + dir * l_max) / 3 >= target_depth) l_max <<= 1;
    while (l_max > 0) {
        l_max >>= 1;
        if (lcp(k, idx, idx + dir * (l + l_max)) / 3 >= target_depth) l += l_max;
    }
    return l;
}

// src/cuda/btree.cu:37-58 (find_split).
int find_split(const Keys &k, int first, int last, int node_lcp, int dir)
{
    int step = 0;
    int length = (last - first) * dir;
    do {
        length = (length + 1) >> 1;
        if (lcp(k, first, first + (step + length) * dir) > node_lcp) step += length;
    } while (length > 1);
    return first + step * dir + std::min(dir, 0);
}

struct Btree {
    std::vector<int> left, right, parent, lcp, edge_delta, depth, lb, le;
    void resize(int n)
    {
        left.assign(n, 0); right.assign(n, 0); parent.assign(n, 0); lcp.assign(n, 0);
        edge_delta.assign(n, 0); depth.assign(n, 0); lb.assign(n, 0); le.assign(n, 0);
    }
};

// src/cuda/btree.cu:90-122 (set_node_child).
void set_node_child(const Keys &k, Btree &t, std::vector<int> &child, int parent_idx,
                    int child_idx, int first_idx, int parent_lcp)
{
    bool is_leaf = child_idx == first_idx;
    int node = 2 * child_idx + !is_leaf;
    child[2 * parent_idx + 1] = node;
    int l = lcp_safe(k, child_idx, first_idx);
    int delta = std::min(1, l / 3 - parent_lcp / 3 + (int) is_leaf);
    t.parent[node] = 2 * parent_idx + 1;
    t.edge_delta[node] = delta;
    t.depth[node] = delta;
    if (is_leaf) {
        t.lcp[node] = parent_lcp + 3;
        t.left[node] = 0;
        t.right[node] = 0;
        t.lb[node] = child_idx;
        t.le[node] = child_idx;
    }
}

// Pairwise softened force in the precision T of the instantiation (the reference
// instantiates every class for float and for double, src/cuda/simulation.cu:255-256).
// include/cuda/physics_common.cuh:46-75 (accumulate_pairwise_force); the
// reference's rsqrt is restated as 1/sqrt in T.  std::fma / std::sqrt on float
// arguments are fmaf / sqrtf.
template <typename T>
inline void pair_force(T p1x, T p1y, T p1z, T p2x, T p2y, T p2z, T m2, T &fx, T &fy, T &fz, T eps2)
{
    T dx = p2x - p1x, dy = p2y - p1y, dz = p2z - p1z;
    T d2 = std::fma(dz, dz, std::fma(dy, dy, dx * dx));
    T s = d2 + eps2;
    T inv = T(1) / std::sqrt(s * s * s);
    inv *= m2;
    fx = std::fma(dx, inv, fx);
    fy = std::fma(dy, inv, fy);
    fz = std::fma(dz, inv, fz);
}

// include/cuda/physics_common.cuh:36-44 (compute_dist_sq).
template <typename T>
inline T dist_sq(T ax, T ay, T az, T bx, T by, T bz)
{
    T dx = ax - bx, dy = ay - by, dz = az - bz;
    return std::fma(dz, dz, std::fma(dy, dy, dx * dx));
}

// out[] = the indices i in [0, n) with keep(i), ascending; returns how many.  Every thread
// counts the kept indices of one contiguous chunk, the chunk offsets are summed serially,
// then every thread writes its chunk: the same list as a serial sweep.
template <typename Keep>
static int compact_indices(int n, Keep keep, int32_t *out)
{
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = omp_get_max_threads();
#endif
    std::vector<int> start(nthreads + 1, 0);
#pragma omp parallel num_threads(nthreads)
    {
        int t = 0, nt = 1;
#ifdef _OPENMP
        t = omp_get_thread_num();
        nt = omp_get_num_threads();
#endif
        const int lo = (int) ((int64_t) n * t / nt), hi = (int) ((int64_t) n * (t + 1) / nt);
        int c = 0;
        for (int i = lo; i < hi; ++i) c += keep(i) ? 1 : 0;
        start[t + 1] = c;
#pragma omp barrier
#pragma omp single
        for (int j = 0; j < nt; ++j) start[j + 1] += start[j];
        int w = start[t];
        for (int i = lo; i < hi; ++i)
            if (keep(i)) out[w++] = i;
        if (t == nt - 1) start[nthreads] = start[nt];
    }
    return start[nthreads];
}

template <typename T>
static void permute(std::vector<T> &v, const std::vector<int32_t> &perm, std::vector<T> &tmp)
{
    const int n = (int) perm.size();
    tmp.resize(n);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) tmp[i] = v[perm[i]];
    v.swap(tmp);
}

}  // namespace

extern "C" {
void orc_sort(const uint64_t *codes, int n, uint64_t *sorted, int32_t *perm);
int orc_unique(const uint64_t *sorted, int n, uint64_t *unique, int32_t *first_pt);
int orc_coarsen(const uint64_t *unique, int U, int bits, int max_codes, int32_t *lfc);
int orc_build_octree(const uint64_t *unique, int U, const int32_t *lfc, int M,
                     const int32_t *first_pt, int bits, float domain, int32_t *first_child,
                     int32_t *num_children, int32_t *leaves_begin, int32_t *leaves_end,
                     int32_t *depth, int32_t *level, float *size, int32_t *points_begin,
                     int32_t *points_end);
}

// ---------------------------------------------------------------------------
// Floating-point stages, written once for the precision T of the instantiation
// (float: Simulation<float>; double: Simulation<double>, src/cuda/simulation.cu:255-256).
// The exported functions further down document each stage and cite its source.
// ---------------------------------------------------------------------------
namespace {

// Keys are taken relative to the cube [lo, lo + domain]^3: the reference's is [0, domain_size]^3
// (lo = 0, and x - 0 is x), the fitted key domain of this build (key_fit below) a cube around
// the particles.
template <typename T>
void morton_impl(const T *x, const T *y, const T *z, int n, float domain, int bits,
                 uint64_t *codes, const float *lo = nullptr)
{
    const T scale = (T) ((1u << bits) - 1u);
    const T dom = (T) domain;
    const T ox = lo ? (T) lo[0] : T(0), oy = lo ? (T) lo[1] : T(0), oz = lo ? (T) lo[2] : T(0);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        volatile T fx = (x[i] - ox) / dom;
        volatile T fy = (y[i] - oy) / dom;
        volatile T fz = (z[i] - oz) / dom;
        key_t_ qx = (key_t_) (fx * scale);
        key_t_ qy = (key_t_) (fy * scale);
        key_t_ qz = (key_t_) (fz * scale);
        codes[i] = spread3(qx, bits) * 4 + spread3(qy, bits) * 2 + spread3(qz, bits);
    }
}

template <typename T>
void com_impl(int n_oct, const int32_t *first_child, const int32_t *num_children,
              const int32_t *points_begin, const int32_t *points_end, const T *x, const T *y,
              const T *z, const T *m, T *comx, T *comy, T *comz, T *mass)
{
    std::vector<T> wx(n_oct), wy(n_oct), wz(n_oct);
    // Nodes are numbered depth-major and a node's children lie one depth further down, so the
    // nodes of one depth are a contiguous range whose successor starts at the first child of
    // its first inner node.  Depths are swept deepest first; the nodes of a depth are
    // independent of each other, and every node's own sum keeps the order stated above.
    std::vector<int> level_begin;
    for (int lo = 0, hi = 1; lo < hi;) {
        level_begin.push_back(lo);
        int first = -1, last = -1;
        for (int o = lo; o < hi; ++o)
            if (num_children[o] != 0) { if (first < 0) first = o; last = o; }
        if (first < 0) { level_begin.push_back(hi); break; }
        lo = first_child[first];
        hi = first_child[last] + num_children[last];
        if (lo >= hi) { level_begin.push_back(lo); break; }
    }
    for (int l = (int) level_begin.size() - 2; l >= 0; --l) {
#pragma omp parallel for schedule(dynamic, 256)
        for (int o = level_begin[l]; o < level_begin[l + 1]; ++o) {
            T sx = 0, sy = 0, sz = 0, sm = 0;
            if (num_children[o] == 0) {
                for (int p = points_begin[o]; p <= points_end[o]; ++p) {
                    T mp = m[p];
                    sx = std::fma(mp, x[p], sx);
                    sy = std::fma(mp, y[p], sy);
                    sz = std::fma(mp, z[p], sz);
                    sm += mp;
                }
            } else {
                for (int c = first_child[o]; c < first_child[o] + num_children[o]; ++c) {
                    sx += wx[c]; sy += wy[c]; sz += wz[c]; sm += mass[c];
                }
            }
            wx[o] = sx; wy[o] = sy; wz[o] = sz; mass[o] = sm;
        }
    }
#pragma omp parallel for schedule(static)
    for (int o = 0; o < n_oct; ++o) {
        comx[o] = wx[o] / mass[o];
        comy[o] = wy[o] / mass[o];
        comz[o] = wz[o] / mass[o];
    }
}

// Bodies (bx,by,bz; n of them, grouped by `group`) against the tree whose points are
// (x,y,z,m).  orc_traverse passes the same arrays for both; orc_traverse_from passes the
// bodies of one domain and the tree built over records imported from other domains.
template <typename T>
void traverse_core(int n, int group, const T *bx, const T *by, const T *bz, const T *x, const T *y,
                   const T *z, const T *m, const int32_t *first_child,
                   const int32_t *num_children, const int32_t *points_begin,
                   const int32_t *points_end, const T *size, const T *comx, const T *comy,
                   const T *comz, const T *mass, T theta, T eps2, T gravity, T domain, T *ax,
                   T *ay, T *az, uint32_t *list_len, int accumulate)
{
    const int ngroups = (n + group - 1) / group;
#pragma omp parallel
    {
        std::vector<int> stack;
        std::vector<T> fx(group), fy(group), fz(group);
#pragma omp for schedule(dynamic, 16)
        for (int g = 0; g < ngroups; ++g) {
            const int b0 = g * group, b1 = std::min(n, b0 + group);
            std::fill(fx.begin(), fx.end(), T(0));
            std::fill(fy.begin(), fy.end(), T(0));
            std::fill(fz.begin(), fz.end(), T(0));
            uint32_t cnt = 0;
            stack.clear();
            stack.push_back(0);
            while (!stack.empty()) {
                int node = stack.back();
                stack.pop_back();
                const int nc = num_children[node];
                if (nc == 0) {
                    for (int p = points_begin[node]; p <= points_end[node]; ++p) {
                        for (int b = b0; b < b1; ++b)
                            pair_force<T>(bx[b], by[b], bz[b], x[p], y[p], z[p], m[p], fx[b - b0],
                                          fy[b - b0], fz[b - b0], eps2);
                        ++cnt;
                    }
                    continue;
                }
                T dmin = domain * domain * T(3);
                for (int b = b0; b < b1; ++b) {
                    T d = dist_sq<T>(bx[b], by[b], bz[b], comx[node], comy[node], comz[node]);
                    if (d <= dmin) dmin = d;
                }
                const T sz = size[node];
                if (sz * sz / dmin < theta * theta) {
                    for (int b = b0; b < b1; ++b)
                        pair_force<T>(bx[b], by[b], bz[b], comx[node], comy[node], comz[node],
                                      mass[node], fx[b - b0], fy[b - b0], fz[b - b0], eps2);
                    ++cnt;
                } else {
                    for (int c = first_child[node] + nc - 1; c >= first_child[node]; --c)
                        stack.push_back(c);
                }
            }
            for (int b = b0; b < b1; ++b) {
                T rx = gravity * fx[b - b0], ry = gravity * fy[b - b0], rz = gravity * fz[b - b0];
                if (accumulate) { rx += ax[b]; ry += ay[b]; rz += az[b]; }
                ax[b] = rx; ay[b] = ry; az[b] = rz;
            }
            list_len[g] = cnt;
        }
    }
}

template <typename T>
inline void reflect(T &p, T &v, T domain)
{
    if (p < T(0) || p > domain) {
        v *= T(-1);
        p = std::max(T(0), std::min(domain, p));
    }
}

template <typename T>
void leapfrog_pos_impl(int n, T dt, T domain, T *x, T *y, T *z, const T *vx, const T *vy,
                       const T *vz, const T *ax, const T *ay, const T *az, T *vhx, T *vhy, T *vhz)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        T hx = std::fma(T(0.5) * ax[i], dt, vx[i]);
        T hy = std::fma(T(0.5) * ay[i], dt, vy[i]);
        T hz = std::fma(T(0.5) * az[i], dt, vz[i]);
        T px = std::fma(hx, dt, x[i]);
        T py = std::fma(hy, dt, y[i]);
        T pz = std::fma(hz, dt, z[i]);
        reflect(px, hx, domain);
        reflect(py, hy, domain);
        reflect(pz, hz, domain);
        x[i] = px; y[i] = py; z[i] = pz;
        vhx[i] = hx; vhy[i] = hy; vhz[i] = hz;
    }
}

template <typename T>
void leapfrog_vel_impl(int n, T dt, T damp, const T *vhx, const T *vhy, const T *vhz, const T *ax,
                       const T *ay, const T *az, T *vx, T *vy, T *vz)
{
    const T k = T(1) - damp;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        vx[i] = std::fma(T(0.5) * ax[i], dt, vhx[i]) * k;
        vy[i] = std::fma(T(0.5) * ay[i], dt, vhy[i]) * k;
        vz[i] = std::fma(T(0.5) * az[i], dt, vhz[i]) * k;
    }
}

// Whole-step simulation object: the sequencing of Simulation<T>::update()
// (src/cuda/simulation.cu:75-119,224-238) over the stage functions.  It is the
// "multithreaded host build of the same algorithm" timed as the CPU baseline, and the
// multi-step oracle for trajectory tests.
template <typename T>
struct OrcSimT {
    int n, bits, max_codes, group;
    float theta, dt, gravity, eps, domain, damp;   // Params stay float (include/simulation.hpp:21-26)
    std::vector<T> x, y, z, m, vx, vy, vz, vhx, vhy, vhz, ax, ay, az;
    std::vector<uint32_t> id;
    // tree
    std::vector<uint64_t> codes, sorted, unique;
    std::vector<int32_t> perm, first_pt, lfc;
    std::vector<int32_t> first_child, num_children, lb, le, depth, level, pb, pe;
    std::vector<float> size_f;
    std::vector<T> size, comx, comy, comz, mass;
    std::vector<uint32_t> list_len;
    int U = 0, M = 0, n_oct = 0;
    // fitted key domain (not in the reference): keys relative to the cube [klo, klo + ksize]^3
    int fit = 0;
    bool kvalid = false;
    float klo[3] = {0.f, 0.f, 0.f}, ksize = 0.f;
    double ms_build = 0, ms_traverse = 0, ms_integrate = 0;
    uint64_t interactions = 0;
};

double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// Fitted key domain.  Not in the reference, whose keys always span [0, domain_size]^3: with 10
// bits per axis that leaves a system that fills a small part of the domain a handful of cells
// across.  The cube is kept while it holds every particle and the particles' largest extent
// is at least 0.4 of its side; otherwise it is re-drawn around the bounding box: side =
// 1.25 x the largest extent (at least domain_size / 2^20), centred on the box.  Float
// arithmetic in exactly this order (the library does the same on the device).
template <typename T>
void key_fit(OrcSimT<T> *s)
{
    float lo[3], hi[3];
    const T *p[3] = {s->x.data(), s->y.data(), s->z.data()};
    for (int a = 0; a < 3; ++a) {
        T mn = p[a][0], mx = p[a][0];
        for (int i = 1; i < s->n; ++i) { mn = std::min(mn, p[a][i]); mx = std::max(mx, p[a][i]); }
        lo[a] = (float) mn; hi[a] = (float) mx;
    }
    float side = 0.f;
    bool inside = s->kvalid;
    for (int a = 0; a < 3; ++a) {
        side = std::max(side, hi[a] - lo[a]);
        inside = inside && lo[a] >= s->klo[a] && hi[a] <= s->klo[a] + s->ksize;
    }
    if (inside && side >= 0.4f * s->ksize) return;
    float size_new = 1.25f * side;
    const float floor_ = s->domain * (1.0f / 1048576.0f);
    if (!(size_new >= floor_)) size_new = floor_;
    for (int a = 0; a < 3; ++a) s->klo[a] = 0.5f * (lo[a] + hi[a]) - 0.5f * size_new;
    s->ksize = size_new;
    s->kvalid = true;
}

// update_points + update_octree (src/cuda/simulation.cu:75-102).
template <typename T>
void sim_build(OrcSimT<T> *s, bool permute_all)
{
    const int n = s->n;
    if (s->fit) {
        key_fit(s);
        morton_impl<T>(s->x.data(), s->y.data(), s->z.data(), n, s->ksize, s->bits, s->codes.data(), s->klo);
    } else {
        morton_impl<T>(s->x.data(), s->y.data(), s->z.data(), n, s->domain, s->bits, s->codes.data());
    }
    orc_sort(s->codes.data(), n, s->sorted.data(), s->perm.data());
    std::vector<T> tf;
    std::vector<uint32_t> ti;
    permute(s->x, s->perm, tf); permute(s->y, s->perm, tf); permute(s->z, s->perm, tf);
    permute(s->m, s->perm, tf);
    permute(s->vhx, s->perm, tf); permute(s->vhy, s->perm, tf); permute(s->vhz, s->perm, tf);
    permute(s->id, s->perm, ti);
    if (permute_all) {  // src/cuda/barnes_hut.cu:577-587 gathers vel and acc too
        permute(s->vx, s->perm, tf); permute(s->vy, s->perm, tf); permute(s->vz, s->perm, tf);
        permute(s->ax, s->perm, tf); permute(s->ay, s->perm, tf); permute(s->az, s->perm, tf);
    }
    s->U = orc_unique(s->sorted.data(), n, s->unique.data(), s->first_pt.data());
    s->M = orc_coarsen(s->unique.data(), s->U, s->bits, s->max_codes, s->lfc.data());
    s->n_oct = orc_build_octree(s->unique.data(), s->U, s->lfc.data(), s->M, s->first_pt.data(),
                                s->bits, s->fit ? s->ksize : s->domain, s->first_child.data(),
                                s->num_children.data(), s->lb.data(), s->le.data(),
                                s->depth.data(), s->level.data(), s->size_f.data(), s->pb.data(),
                                s->pe.data());
    // domain / 2^level: a power-of-two scaling of the float parameter, exact in T
    for (int o = 0; o < s->n_oct; ++o) s->size[o] = (T) s->size_f[o];
    com_impl<T>(s->n_oct, s->first_child.data(), s->num_children.data(), s->pb.data(),
                s->pe.data(), s->x.data(), s->y.data(), s->z.data(), s->m.data(), s->comx.data(),
                s->comy.data(), s->comz.data(), s->mass.data());
}

template <typename T>
OrcSimT<T> *sim_create(int n, int bits, int max_codes, int group, float theta, float dt,
                       float gravity, float eps, float domain, float damp, const T *x, const T *y,
                       const T *z, const T *vx, const T *vy, const T *vz, const T *m, int fit = 0)
{
    OrcSimT<T> *s = new OrcSimT<T>();
    s->fit = fit;
    s->n = n; s->bits = bits; s->max_codes = max_codes; s->group = group;
    s->theta = theta; s->dt = dt; s->gravity = gravity; s->eps = eps; s->domain = domain;
    s->damp = damp;
    auto cp = [&](std::vector<T> &d, const T *src) {
        d.assign(n, T(0));
        if (src) std::memcpy(d.data(), src, sizeof(T) * n);
    };
    cp(s->x, x); cp(s->y, y); cp(s->z, z); cp(s->vx, vx); cp(s->vy, vy); cp(s->vz, vz);
    cp(s->m, m);
    cp(s->vhx, nullptr); cp(s->vhy, nullptr); cp(s->vhz, nullptr);
    cp(s->ax, nullptr); cp(s->ay, nullptr); cp(s->az, nullptr);
    s->id.resize(n);
    std::iota(s->id.begin(), s->id.end(), 0u);
    s->codes.resize(n); s->sorted.resize(n); s->unique.resize(n);
    s->perm.resize(n); s->first_pt.resize(n + 1); s->lfc.resize(n + 1);
    const int cap = 2 * n;
    s->first_child.resize(cap); s->num_children.resize(cap); s->lb.resize(cap);
    s->le.resize(cap); s->depth.resize(cap); s->level.resize(cap); s->pb.resize(cap);
    s->pe.resize(cap); s->size_f.resize(cap); s->size.resize(cap); s->comx.resize(cap);
    s->comy.resize(cap); s->comz.resize(cap); s->mass.resize(cap);
    s->list_len.resize((n + group - 1) / group);
    // exec_post_spawn (src/cuda/simulation.cu:160-168): sort + tree, no forces.
    sim_build(s, true);
    return s;
}

// Simulation<T>::update (src/cuda/simulation.cu:224-238).
template <typename T>
void sim_step(OrcSimT<T> *s)
{
    double t0 = now_ms();
    leapfrog_pos_impl<T>(s->n, (T) s->dt, (T) s->domain, s->x.data(), s->y.data(), s->z.data(),
                         s->vx.data(), s->vy.data(), s->vz.data(), s->ax.data(), s->ay.data(),
                         s->az.data(), s->vhx.data(), s->vhy.data(), s->vhz.data());
    double t1 = now_ms();
    sim_build(s, false);
    double t2 = now_ms();
    const T eps = (T) s->eps;
    traverse_core<T>(s->n, s->group, s->x.data(), s->y.data(), s->z.data(), s->x.data(),
                     s->y.data(), s->z.data(), s->m.data(), s->first_child.data(),
                     s->num_children.data(), s->pb.data(), s->pe.data(), s->size.data(),
                     s->comx.data(), s->comy.data(), s->comz.data(), s->mass.data(), (T) s->theta,
                     eps * eps, (T) s->gravity, (T) s->domain, s->ax.data(), s->ay.data(),
                     s->az.data(), s->list_len.data(), 0);
    double t3 = now_ms();
    leapfrog_vel_impl<T>(s->n, (T) s->dt, (T) s->damp, s->vhx.data(), s->vhy.data(), s->vhz.data(),
                         s->ax.data(), s->ay.data(), s->az.data(), s->vx.data(), s->vy.data(),
                         s->vz.data());
    double t4 = now_ms();
    s->ms_integrate = (t1 - t0) + (t4 - t3);
    s->ms_build = t2 - t1;
    s->ms_traverse = t3 - t2;
    uint64_t inter = 0;
    const int ng = (int) s->list_len.size();
    for (int g = 0; g < ng; ++g) {
        int nb = std::min(s->group, s->n - g * s->group);
        inter += (uint64_t) s->list_len[g] * nb;
    }
    s->interactions = inter;
}

template <typename T>
void sim_get_state(OrcSimT<T> *s, T *x, T *y, T *z, T *vx, T *vy, T *vz, T *m, T *ax, T *ay,
                   T *az, uint32_t *id)
{
    auto cp = [&](T *d, const std::vector<T> &v) {
        if (d) std::memcpy(d, v.data(), sizeof(T) * s->n);
    };
    cp(x, s->x); cp(y, s->y); cp(z, s->z); cp(vx, s->vx); cp(vy, s->vy); cp(vz, s->vz);
    cp(m, s->m); cp(ax, s->ax); cp(ay, s->ay); cp(az, s->az);
    if (id) std::memcpy(id, s->id.data(), sizeof(uint32_t) * s->n);
}

}  // namespace

using OrcSim = OrcSimT<float>;
using OrcSimD = OrcSimT<double>;

extern "C" {

// ---------------------------------------------------------------------------
// Morton encode.  src/cuda/points.cu:16-38: q = (u64)(p / domain * (2^B-1))
// evaluated in float32 (divide, then multiply, truncate); x is the most
// significant bit of each triple: code = X*4 + Y*2 + Z.
// ---------------------------------------------------------------------------
void orc_morton(const float *x, const float *y, const float *z, int n, float domain, int bits,
                uint64_t *codes)
{
    morton_impl<float>(x, y, z, n, domain, bits, codes);
}
// The double instantiation: positions are double, `domain_size` stays a float and the
// division and the multiplication by 2^B-1 are done in double (src/cuda/points.cu:28-30
// with T = double).
void orc_morton_dp(const double *x, const double *y, const double *z, int n, float domain,
                   int bits, uint64_t *codes)
{
    morton_impl<double>(x, y, z, n, domain, bits, codes);
}

// ---------------------------------------------------------------------------
// Sort.  src/cuda/points.cu:73-84: (code, iota) pairs sorted ascending by code;
// ties pinned to original index order (stable).  perm[i] = pre-sort index of
// the i-th sorted body.
// ---------------------------------------------------------------------------
void orc_sort(const uint64_t *codes, int n, uint64_t *sorted, int32_t *perm)
{
    // A stable sort has one possible result, so the multithreaded sort (sorted chunks merged
    // pairwise, ties taken from the earlier chunk) gives the serial one's order.
    std::vector<int32_t> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    __gnu_parallel::stable_sort(idx.begin(), idx.end(),
                                [&](int32_t a, int32_t b) { return codes[a] < codes[b]; });
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        perm[i] = idx[i];
        sorted[i] = codes[idx[i]];
    }
}

// ---------------------------------------------------------------------------
// Unique codes + first point index.  src/cuda/points.cu:116-142 (run-length
// encode + exclusive scan of the run lengths); first_pt[U] = n.
// Returns U.  unique has room for n entries, first_pt for n+1.
// ---------------------------------------------------------------------------
int orc_unique(const uint64_t *sorted, int n, uint64_t *unique, int32_t *first_pt)
{
    const int u = compact_indices(n, [&](int i) { return i == 0 || sorted[i] != sorted[i - 1]; },
                                  first_pt);
#pragma omp parallel for schedule(static)
    for (int j = 0; j < u; ++j) unique[j] = sorted[first_pt[j]];
    first_pt[u] = n;
    return u;
}

// ---------------------------------------------------------------------------
// Leaf coarsening.  src/cuda/btree.cu:124-187 (reduce_leaf_depth, merge_leaves)
// and :420-455 (generate_leaves: flagged partition; the rejected tail leaves
// leaf_first_code_idx[M] == U, which compute_octree_nodes_points_range relies
// on, src/cuda/octree.cu:103).  Returns M; lfc has room for U+1 entries.
// ---------------------------------------------------------------------------
int orc_coarsen(const uint64_t *unique, int U, int bits, int max_codes, int32_t *lfc)
{
    Keys k{unique, U, 3 * bits};
    if (U <= max_codes) {  // reference: undefined (search_lr never terminates)
        lfc[0] = 0;
        lfc[1] = U;
        return 1;
    }
    std::vector<int> leaf_depth(U);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < U; ++i)
        leaf_depth[i] = std::max(lcp(k, i, i - 1) / 3, lcp(k, i, i + 1) / 3) + 1;

    std::vector<char> flag(U);
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < U; ++idx) {
        int left_old = 0, left = 0, right = 0, dd = 0;
        int num = 1;  // leaf_first_in is iota (src/cuda/btree.cu:440): counts are codes
        while (num <= max_codes) {
            dd++;
            left_old = left;
            left = search_lr(k, leaf_depth.data(), -1, idx, dd);
            right = search_lr(k, leaf_depth.data(), 1, idx, dd);
            num = (idx + right + 1) - (idx - left);
        }
        flag[idx] = left_old == 0;
    }
    const int m = compact_indices(U, [&](int i) { return flag[i] != 0; }, lfc);
    lfc[m] = U;
    return m;
}

// ---------------------------------------------------------------------------
// Radix tree -> BFS order -> octree arrays.
//   src/cuda/btree.cu:190-304 (build_radix_tree), :306-336,:479-494 (depth by
//   pointer jumping), :497-559 (stable sort by depth, gathers, child pointer
//   fix-up), :561-572 (octree map = exclusive scan of edge_delta);
//   src/cuda/octree.cu:13-83 (build_octree), :85-107 (points ranges).
// All output arrays have room for 2M-1 entries.  first_child of a leaf is
// written as -1 (the reference leaves it stale).  Returns n_oct.
// ---------------------------------------------------------------------------
int orc_build_octree(const uint64_t *unique, int U, const int32_t *lfc, int M,
                     const int32_t *first_pt, int bits, float domain, int32_t *first_child,
                     int32_t *num_children, int32_t *leaves_begin, int32_t *leaves_end,
                     int32_t *depth, int32_t *level, float *size, int32_t *points_begin,
                     int32_t *points_end)
{
    (void) U;
    if (M == 1) {  // single root leaf (our definition, see header)
        first_child[0] = -1; num_children[0] = 0; leaves_begin[0] = 0; leaves_end[0] = 0;
        depth[0] = 0; level[0] = 0; size[0] = domain;
        points_begin[0] = first_pt[lfc[0]];
        points_end[0] = first_pt[lfc[1]] - 1;
        return 1;
    }
    // src/cuda/btree.cu:464-468: first code of each leaf.
    std::vector<key_t_> lk(M);
#pragma omp parallel for schedule(static)
    for (int j = 0; j < M; ++j) lk[j] = unique[lfc[j]];
    Keys k{lk.data(), M, 3 * bits};

    const int nbt = 2 * M - 1;
    Btree t;
    t.resize(nbt);

    // src/cuda/btree.cu:230-235: root.
    t.parent[1] = 0; t.depth[1] = 0; t.edge_delta[1] = 1;
    // one iteration per internal node: it writes its own record and those of its two children,
    // which no other node has as a child
#pragma omp parallel for schedule(dynamic, 1024)
    for (int idx = 0; idx < M - 1; ++idx) {
        int d = sign(lcp(k, idx, idx + 1) - lcp(k, idx, idx - 1));
        int lcp_min = lcp(k, idx, idx - d);
        int max_length = 2;
        while (lcp(k, idx, idx + max_length * d) > lcp_min) max_length <<= 1;
        int length = 0;
        int step = max_length;
        do {
            step >>= 1;
            if (lcp(k, idx, idx + (length + step) * d) > lcp_min) length += step;
        } while (step > 1);
        int last = idx + length * d;
        int node_lcp = lcp_safe(k, idx, last);
        t.lcp[2 * idx + 1] = node_lcp;
        int split = find_split(k, idx, last, node_lcp, d);
        int min_leaf = std::min(idx, last), max_leaf = std::max(idx, last);
        t.lb[2 * idx + 1] = min_leaf;
        t.le[2 * idx + 1] = max_leaf;
        set_node_child(k, t, t.left, idx, split, min_leaf, node_lcp);
        set_node_child(k, t, t.right, idx, split + 1, max_leaf, node_lcp);
    }

    // src/cuda/btree.cu:306-336 x (ceil(log2(nbt))+3) rounds: pointer jumping.
    {
        int rounds = 3;
        for (int n = 1; n < nbt; n <<= 1) rounds++;
        std::vector<int> p_in = t.parent, d_in = t.depth, p_out(nbt), d_out(nbt);
        for (int r = 0; r < rounds; ++r) {
#pragma omp parallel for schedule(static)
            for (int i = 0; i < nbt; ++i) {
                int cp = p_in[i], cd = d_in[i];
                bool flag = cp && cd > 0;
                d_out[i] = cd + d_in[cp] * (int) flag;
                p_out[i] = flag ? p_in[cp] : cp;
            }
            p_in.swap(p_out);
            d_in.swap(d_out);
        }
        t.depth = d_in;
    }

    // src/cuda/btree.cu:497-559: stable sort by depth, permute, fix children.
    std::vector<int> order(nbt), inv(nbt);
    std::iota(order.begin(), order.end(), 0);
    __gnu_parallel::stable_sort(order.begin(), order.end(),
                                [&](int a, int b) { return t.depth[a] < t.depth[b]; });
#pragma omp parallel for schedule(static)
    for (int i = 0; i < nbt; ++i) inv[order[i]] = i;
    Btree s;
    s.resize(nbt);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < nbt; ++i) {
        int o = order[i];
        s.depth[i] = t.depth[o];
        s.lcp[i] = t.lcp[o];
        s.edge_delta[i] = t.edge_delta[o];
        s.lb[i] = t.lb[o];
        s.le[i] = t.le[o];
        int l = t.left[o], r = t.right[o];
        if (l != 0 || r != 0) { l = inv[l]; r = inv[r]; }
        s.left[i] = l;
        s.right[i] = r;
    }
    // src/cuda/btree.cu:561-572: octree map.
    std::vector<int> map(nbt);
    int acc = 0;
    for (int i = 0; i < nbt; ++i) { map[i] = acc; acc += s.edge_delta[i]; }
    const int n_oct = map[nbt - 1] + s.edge_delta[nbt - 1];

    // src/cuda/octree.cu:26-83 (build_octree) + :13-24 (traverse).
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < nbt; ++idx) {
        if (!s.edge_delta[idx]) continue;
        int o = map[idx];
        num_children[o] = 0;
        first_child[o] = -1;
        int lvl = s.lcp[idx] / 3;
        level[o] = lvl;
        size[o] = domain / (float) (1 << lvl);
        leaves_begin[o] = s.lb[idx];
        leaves_end[o] = s.le[idx];
        depth[o] = s.depth[idx];
        if (s.lb[idx] == s.le[idx]) continue;
        int fb = idx;
        do { fb = s.left[fb]; } while (!s.edge_delta[fb]);
        int lbn = idx;
        do { lbn = s.right[lbn]; } while (!s.edge_delta[lbn]);
        first_child[o] = map[fb];
        num_children[o] = map[lbn] - map[fb] + 1;
    }
    // src/cuda/octree.cu:85-107.
#pragma omp parallel for schedule(static)
    for (int o = 0; o < n_oct; ++o) {
        points_begin[o] = first_pt[lfc[leaves_begin[o]]];
        points_end[o] = first_pt[lfc[leaves_end[o] + 1]] - 1;
    }
    return n_oct;
}

// ---------------------------------------------------------------------------
// Centre of mass.  src/cuda/octree.cu:109-217,292-346: leaves sum m*pos over
// their points, internal nodes sum their children deepest level first, then
// every node divides by its mass.  Order fixed as stated in the file header.
// Nodes are in depth-major order, so a reverse index sweep is "deepest first".
// ---------------------------------------------------------------------------
void orc_com(int n_oct, const int32_t *first_child, const int32_t *num_children,
             const int32_t *points_begin, const int32_t *points_end, const float *x,
             const float *y, const float *z, const float *m, float *comx, float *comy,
             float *comz, float *mass)
{
    com_impl<float>(n_oct, first_child, num_children, points_begin, points_end, x, y, z, m, comx,
                    comy, comz, mass);
}
void orc_com_dp(int n_oct, const int32_t *first_child, const int32_t *num_children,
                const int32_t *points_begin, const int32_t *points_end, const double *x,
                const double *y, const double *z, const double *m, double *comx, double *comy,
                double *comz, double *mass)
{
    com_impl<double>(n_oct, first_child, num_children, points_begin, points_end, x, y, z, m, comx,
                     comy, comz, mass);
}

// ---------------------------------------------------------------------------
// Group traversal.  src/cuda/barnes_hut.cu:199-537: one group = G consecutive
// sorted bodies; per candidate node d2min = min over the group's bodies of
// |x_b - com|^2 (:37-61, initial value 3*domain^2); a node with no children is
// always evaluated directly against every point of its range, self included
// (:144-197,:428-474); otherwise size^2/d2min < theta^2 -> approximate by
// (com, mass) (:31-35,:95-142), else open (:63-93,:404-426).  acc = G_grav * f
// (:533-536).  The interaction SET is order independent; this walk is depth
// first.  list_len[g] = number of (com|point) sources the group interacted
// with; interactions = sum_g list_len[g] * bodies_in_group(g).
// ---------------------------------------------------------------------------
void orc_traverse(int n, int group, const float *x, const float *y, const float *z,
                  const float *m, int n_oct, const int32_t *first_child,
                  const int32_t *num_children, const int32_t *points_begin,
                  const int32_t *points_end, const float *size, const float *comx,
                  const float *comy, const float *comz, const float *mass, float theta,
                  float eps2, float gravity, float domain, float *ax, float *ay, float *az,
                  uint32_t *list_len)
{
    (void) n_oct;
    traverse_core<float>(n, group, x, y, z, x, y, z, m, first_child, num_children, points_begin,
                         points_end, size, comx, comy, comz, mass, theta, eps2, gravity, domain, ax,
                         ay, az, list_len, 0);
}
// T = double: theta, softening^2, gravity and the domain size are the float parameters
// widened (the reference keeps Params in float for both instantiations,
// include/simulation.hpp:21-26); node sizes are domain / 2^level, exact in either type.
void orc_traverse_dp(int n, int group, const double *x, const double *y, const double *z,
                     const double *m, int n_oct, const int32_t *first_child,
                     const int32_t *num_children, const int32_t *points_begin,
                     const int32_t *points_end, const double *size, const double *comx,
                     const double *comy, const double *comz, const double *mass, double theta,
                     double eps2, double gravity, double domain, double *ax, double *ay,
                     double *az, uint32_t *list_len)
{
    (void) n_oct;
    traverse_core<double>(n, group, x, y, z, x, y, z, m, first_child, num_children, points_begin,
                          points_end, size, comx, comy, comz, mass, theta, eps2, gravity, domain,
                          ax, ay, az, list_len, 0);
}

// ---------------------------------------------------------------------------
// Multi-GPU locally-essential-tree exchange.  NOT a restatement of the reference (it is
// single-GPU): these two functions are an independent, top-down host formulation of what
// libnbody_mi355.so's nbody_traverse_from / nbody_let_export compute, used to check them.
// ---------------------------------------------------------------------------
void orc_traverse_from(int nb, int group, const float *bx, const float *by, const float *bz,
                       const float *x, const float *y, const float *z, const float *m,
                       const int32_t *first_child, const int32_t *num_children,
                       const int32_t *points_begin, const int32_t *points_end,
                       const float *size, const float *comx, const float *comy,
                       const float *comz, const float *mass, float theta, float eps2,
                       float gravity, float domain, float *ax, float *ay, float *az,
                       uint32_t *list_len, int accumulate)
{
    traverse_core<float>(nb, group, bx, by, bz, x, y, z, m, first_child, num_children,
                         points_begin, points_end, size, comx, comy, comz, mass, theta, eps2,
                         gravity, domain, ax, ay, az, list_len, accumulate);
}

// Records (x,y,z,m) a remote domain whose particles lie in the union of `nboxes` boxes (lo xyz,
// hi xyz each) needs from this tree: depth-first from the root; a node with children whose
// size^2 / d^2 < theta^2, d the smallest distance from any of the boxes to the node's centre
// of mass, contributes (com, mass); a node
// without children contributes its points; anything else is opened.  Returns the number of
// records; writes at most `cap` of them.
int orc_let_export(const int32_t *first_child, const int32_t *num_children,
                   const int32_t *points_begin, const int32_t *points_end, const float *size,
                   const float *comx, const float *comy, const float *comz, const float *mass,
                   const float *x, const float *y, const float *z, const float *m,
                   const float *boxes, int nboxes, float theta, float *out, int cap)
{
    int cnt = 0;
    auto emit = [&](float a, float b, float c, float d) {
        if (cnt < cap) { out[4 * cnt] = a; out[4 * cnt + 1] = b; out[4 * cnt + 2] = c; out[4 * cnt + 3] = d; }
        ++cnt;
    };
    std::vector<int> stack{0};
    while (!stack.empty()) {
        const int o = stack.back();
        stack.pop_back();
        if (num_children[o] == 0) {
            for (int p = points_begin[o]; p <= points_end[o]; ++p) emit(x[p], y[p], z[p], m[p]);
            continue;
        }
        float d2 = 3.0e38f;
        for (int b = 0; b < nboxes; ++b) {
            const float *box = boxes + 6 * b;
            const float dx = std::max(0.0f, std::max(box[0] - comx[o], comx[o] - box[3]));
            const float dy = std::max(0.0f, std::max(box[1] - comy[o], comy[o] - box[4]));
            const float dz = std::max(0.0f, std::max(box[2] - comz[o], comz[o] - box[5]));
            d2 = std::min(d2, fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
        }
        if (size[o] * size[o] / d2 < theta * theta) emit(comx[o], comy[o], comz[o], mass[o]);
        else
            for (int c = first_child[o]; c < first_child[o] + num_children[o]; ++c) stack.push_back(c);
    }
    return cnt;
}

// ---------------------------------------------------------------------------
// Leapfrog.  src/cuda/physics_common.cu:69-147: v_half = v + 0.5*a*dt;
// x += v_half*dt; reflective walls flip the velocity component and clamp the
// coordinate; v = (v_half + 0.5*a*dt) * (1 - damp).
// ---------------------------------------------------------------------------
void orc_leapfrog_pos(int n, float dt, float domain, float *x, float *y, float *z,
                      const float *vx, const float *vy, const float *vz, const float *ax,
                      const float *ay, const float *az, float *vhx, float *vhy, float *vhz)
{
    leapfrog_pos_impl<float>(n, dt, domain, x, y, z, vx, vy, vz, ax, ay, az, vhx, vhy, vhz);
}

void orc_leapfrog_vel(int n, float dt, float damp, const float *vhx, const float *vhy,
                      const float *vhz, const float *ax, const float *ay, const float *az,
                      float *vx, float *vy, float *vz)
{
    leapfrog_vel_impl<float>(n, dt, damp, vhx, vhy, vhz, ax, ay, az, vx, vy, vz);
}

// ---------------------------------------------------------------------------
// Float64 direct sum.  src/cuda/validator.cu:134-174 (compute_ap_forces, self
// pair included, contributes exactly 0) evaluated in double as the accuracy
// reference; targets [t0,t1) against all n sources.
// ---------------------------------------------------------------------------
void orc_direct_f64(int n, int t0, int t1, const float *x, const float *y, const float *z,
                    const float *m, double eps2, double gravity, double *ax, double *ay,
                    double *az)
{
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = t0; i < t1; ++i) {
        double fx = 0, fy = 0, fz = 0;
        const double px = x[i], py = y[i], pz = z[i];
        for (int j = 0; j < n; ++j) {
            double dx = x[j] - px, dy = y[j] - py, dz = z[j] - pz;
            double s = dx * dx + dy * dy + dz * dz + eps2;
            double inv = m[j] / (s * std::sqrt(s));
            fx += dx * inv; fy += dy * inv; fz += dz * inv;
        }
        ax[i - t0] = gravity * fx; ay[i - t0] = gravity * fy; az[i - t0] = gravity * fz;
    }
}

// Float32 direct sum in the reference's own arithmetic (validator.cu:134-174),
// sources in index order; used to check the GPU direct-sum kernel.
void orc_direct_f32(int n, int t0, int t1, const float *x, const float *y, const float *z,
                    const float *m, float eps2, float gravity, float *ax, float *ay, float *az)
{
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = t0; i < t1; ++i) {
        float fx = 0, fy = 0, fz = 0;
        for (int j = 0; j < n; ++j)
            pair_force(x[i], y[i], z[i], x[j], y[j], z[j], m[j], fx, fy, fz, eps2);
        ax[i - t0] = gravity * fx; ay[i - t0] = gravity * fy; az[i - t0] = gravity * fz;
    }
}

// ---------------------------------------------------------------------------
// Total energy.  src/cuda/validator.cu:19-36,65-132: sum_i m_i (0.5 v_i^2 -
// 0.5 G sum_{j!=i} m_j / sqrt(d^2 + eps^2)), in double.
// ---------------------------------------------------------------------------
double orc_energy_f64(int n, const float *x, const float *y, const float *z, const float *vx,
                      const float *vy, const float *vz, const float *m, double eps2,
                      double gravity)
{
    double total = 0;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : total)
    for (int i = 0; i < n; ++i) {
        double e = 0.5 * ((double) vx[i] * vx[i] + (double) vy[i] * vy[i] + (double) vz[i] * vz[i]);
        double pot = 0;
        for (int j = 0; j < n; ++j) {
            if (j == i) continue;
            double dx = (double) x[j] - x[i], dy = (double) y[j] - y[i], dz = (double) z[j] - z[i];
            pot += m[j] / std::sqrt(dx * dx + dy * dy + dz * dz + eps2);
        }
        e -= 0.5 * gravity * pot;
        total += e * m[i];
    }
    return total;
}

// Mean and max of |a - a_ref| / |a_ref| (src/cuda/validator.cu:38-63) over k bodies.
void orc_acc_error(int k, const float *ax, const float *ay, const float *az, const double *rx,
                   const double *ry, const double *rz, double *mean_out, double *max_out)
{
    double sum = 0, mx = 0;
    for (int i = 0; i < k; ++i) {
        double dx = ax[i] - rx[i], dy = ay[i] - ry[i], dz = az[i] - rz[i];
        double e = std::sqrt(dx * dx + dy * dy + dz * dz) /
                   std::sqrt(rx[i] * rx[i] + ry[i] * ry[i] + rz[i] * rz[i]);
        sum += e;
        mx = std::max(mx, e);
    }
    *mean_out = sum / k;
    *max_out = mx;
}

// ---------------------------------------------------------------------------
// Whole-step simulation object: the sequencing of Simulation<T>::update()
// (src/cuda/simulation.cu:75-119,224-238) over the stage functions above.  It
// is the "multithreaded host build of the same algorithm" timed as the CPU
// baseline, and the multi-step oracle for trajectory tests.
// ---------------------------------------------------------------------------
OrcSim *orc_sim_create(int n, int bits, int max_codes, int group, float theta, float dt,
                       float gravity, float eps, float domain, float damp, const float *x,
                       const float *y, const float *z, const float *vx, const float *vy,
                       const float *vz, const float *m)
{
    return sim_create<float>(n, bits, max_codes, group, theta, dt, gravity, eps, domain, damp, x, y,
                             z, vx, vy, vz, m);
}
// The same with the fitted key domain (key_fit above).
OrcSim *orc_sim_create_fit(int n, int bits, int max_codes, int group, float theta, float dt,
                           float gravity, float eps, float domain, float damp, const float *x,
                           const float *y, const float *z, const float *vx, const float *vy,
                           const float *vz, const float *m)
{
    return sim_create<float>(n, bits, max_codes, group, theta, dt, gravity, eps, domain, damp, x, y,
                             z, vx, vy, vz, m, 1);
}
// klo xyz, ksize of the current key cube (zeros without the fitted key domain).
void orc_sim_get_key_domain(OrcSim *s, float *out)
{
    out[0] = s->klo[0]; out[1] = s->klo[1]; out[2] = s->klo[2]; out[3] = s->ksize;
}
// Tree arrays of the last build (n_oct entries each; pointers may be null) and the sorted codes.
void orc_sim_get_tree(OrcSim *s, uint64_t *sorted, int32_t *first_child, int32_t *num_children,
                      int32_t *points_begin, int32_t *points_end, float *size, float *comx,
                      float *comy, float *comz, float *mass, uint32_t *list_len)
{
    const size_t no = (size_t) s->n_oct;
    if (sorted) std::memcpy(sorted, s->sorted.data(), sizeof(uint64_t) * s->n);
    if (first_child) std::memcpy(first_child, s->first_child.data(), 4 * no);
    if (num_children) std::memcpy(num_children, s->num_children.data(), 4 * no);
    if (points_begin) std::memcpy(points_begin, s->pb.data(), 4 * no);
    if (points_end) std::memcpy(points_end, s->pe.data(), 4 * no);
    if (size) std::memcpy(size, s->size_f.data(), 4 * no);
    if (comx) std::memcpy(comx, s->comx.data(), 4 * no);
    if (comy) std::memcpy(comy, s->comy.data(), 4 * no);
    if (comz) std::memcpy(comz, s->comz.data(), 4 * no);
    if (mass) std::memcpy(mass, s->mass.data(), 4 * no);
    if (list_len) std::memcpy(list_len, s->list_len.data(), 4 * s->list_len.size());
}
void orc_sim_destroy(OrcSim *s) { delete s; }
void orc_sim_step(OrcSim *s) { sim_step(s); }
void orc_sim_get_state(OrcSim *s, float *x, float *y, float *z, float *vx, float *vy, float *vz,
                       float *m, float *ax, float *ay, float *az, uint32_t *id)
{
    sim_get_state(s, x, y, z, vx, vy, vz, m, ax, ay, az, id);
}
void orc_sim_get_counts(OrcSim *s, int32_t *U, int32_t *M, int32_t *n_oct,
                        uint64_t *interactions)
{
    *U = s->U; *M = s->M; *n_oct = s->n_oct; *interactions = s->interactions;
}
void orc_sim_get_timings(OrcSim *s, double *ms_integrate, double *ms_build, double *ms_traverse)
{
    *ms_integrate = s->ms_integrate; *ms_build = s->ms_build; *ms_traverse = s->ms_traverse;
}

// Simulation<double> (src/cuda/simulation.cu:256): the same object with every float array
// and every floating-point operation in double.
OrcSimD *orc_sim_create_dp(int n, int bits, int max_codes, int group, float theta, float dt,
                           float gravity, float eps, float domain, float damp, const double *x,
                           const double *y, const double *z, const double *vx, const double *vy,
                           const double *vz, const double *m)
{
    return sim_create<double>(n, bits, max_codes, group, theta, dt, gravity, eps, domain, damp, x,
                              y, z, vx, vy, vz, m);
}
void orc_sim_destroy_dp(OrcSimD *s) { delete s; }
void orc_sim_step_dp(OrcSimD *s) { sim_step(s); }
void orc_sim_get_state_dp(OrcSimD *s, double *x, double *y, double *z, double *vx, double *vy,
                          double *vz, double *m, double *ax, double *ay, double *az, uint32_t *id)
{
    sim_get_state(s, x, y, z, vx, vy, vz, m, ax, ay, az, id);
}
void orc_sim_get_counts_dp(OrcSimD *s, int32_t *U, int32_t *M, int32_t *n_oct,
                           uint64_t *interactions)
{
    *U = s->U; *M = s->M; *n_oct = s->n_oct; *interactions = s->interactions;
}
// Tree products of the last build: integer arrays as in orc_build_octree, centres of mass in double.
void orc_sim_get_tree_dp(OrcSimD *s, uint64_t *sorted, int32_t *first_child, int32_t *num_children,
                         int32_t *points_begin, int32_t *points_end, double *comx, double *comy,
                         double *comz, double *mass, uint32_t *list_len)
{
    if (sorted) std::memcpy(sorted, s->sorted.data(), sizeof(uint64_t) * s->n);
    const size_t no = (size_t) s->n_oct;
    if (first_child) std::memcpy(first_child, s->first_child.data(), 4 * no);
    if (num_children) std::memcpy(num_children, s->num_children.data(), 4 * no);
    if (points_begin) std::memcpy(points_begin, s->pb.data(), 4 * no);
    if (points_end) std::memcpy(points_end, s->pe.data(), 4 * no);
    if (comx) std::memcpy(comx, s->comx.data(), 8 * no);
    if (comy) std::memcpy(comy, s->comy.data(), 8 * no);
    if (comz) std::memcpy(comz, s->comz.data(), 8 * no);
    if (mass) std::memcpy(mass, s->mass.data(), 8 * no);
    if (list_len) std::memcpy(list_len, s->list_len.data(), 4 * s->list_len.size());
}

// All-pairs accelerations of double-precision particles, in double (src/cuda/validator.cu:
// 134-174 with T = double).
void orc_direct_dp(int n, int t0, int t1, const double *x, const double *y, const double *z,
                   const double *m, double eps2, double gravity, double *ax, double *ay,
                   double *az)
{
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = t0; i < t1; ++i) {
        double fx = 0, fy = 0, fz = 0;
        for (int j = 0; j < n; ++j)
            pair_force<double>(x[i], y[i], z[i], x[j], y[j], z[j], m[j], fx, fy, fz, eps2);
        ax[i - t0] = gravity * fx; ay[i - t0] = gravity * fy; az[i - t0] = gravity * fz;
    }
}

// Threads the OpenMP regions of this file use from now on (tests compare a one-thread build
// with a many-thread one).
void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : 1);
#else
    (void) n;
#endif
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

}  // extern "C"

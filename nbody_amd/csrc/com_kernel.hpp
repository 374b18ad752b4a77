// com_kernel.hpp -- centre-of-mass aggregation, one launch, shared by the float and the
// double handles (R = float / double, R4 = Float4 / Double4).
//
// Reference stages restated: compute_octree_leaves_weighted_pos, the per-depth
// compute_octree_nodes_weighted_pos launches and compute_octree_nodes_barycenter
// (src/cuda/octree.cu:109-217,292-346).
//
// Summation order is fixed, whatever the scheduling, so that the host checker reproduces the
// result bit for bit: a leaf sums its particles in index order (fused multiply-adds), a node
// its children in index order.
//
// Two regimes.  The bottom of the tree is done WITHOUT any device-wide hand-off: an inner node
// with at most COM_LOCAL particles whose parent has more is a "local root" (k_emit lists
// them), and one wave computes its whole subtree level by level from the deepest -- in the
// breadth-first numbering every level of a subtree is a contiguous range of node ids, found
// from the first and last inner node of the level above; lanes take the nodes of a level, and
// sums pass between lanes of the wave through the CU's own cache (workgroup-scope fences).
// Only above the local roots does the atomic-counter climb run: a thread that has finished a
// node bumps the parent's arrival counter and stops unless it was the last child to arrive, in
// which case it sums the parent's children and goes on.  A level of that climb is three
// device-scope round trips past the XCD's L2 (write-through stores, counter, children) --
// several microseconds -- which is why it is kept to the top log8(N / COM_LOCAL) levels.
// Hand-off in the climb: sums are stored write-through (agent-scope atomic stores), the storing
// wave drains its stores (s_waitcnt vmcnt(0)) before the agent-scope counter add, and the last
// arriver reads the sums with agent-scope atomic loads -- only after its add has returned.
// The sums never sit in a cache the reader does not see, so no cache maintenance is needed;
// an acquire-release counter add instead (the compiler then writes back and invalidates the
// XCD's L2 around every add) was measured at 0.30 ms against 0.064 ms for this kernel at
// 2,097,152 particles.
// Leaves that hang directly off a node above the local roots start the climb themselves.
#pragma once
#include "dev_utils.hpp"
#include "nbody_internal.hpp"

// The hand-off of the climb (write-through stores, s_waitcnt vmcnt(0), relaxed agent-scope
// counter add, agent-scope loads) is written for the cache hierarchy of gfx942 / gfx950, where
// agent-scope atomic stores and loads go past the XCD's L2.  It is not a statement in the HIP
// memory model: on another target the counter add would have to be acquire-release
// (__ATOMIC_ACQ_REL) and the sums ordinary stores and loads.  Refuse to build elsewhere.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "com_kernel.hpp: the centre-of-mass hand-off is written for gfx942/gfx950 (see the note above)"
#endif

namespace nb {

// most particles under a local root: what one wave stages in LDS (8 KB)
template <typename R> constexpr int com_local() { return 8192 / (4 * (int) sizeof(R)); }
constexpr int COM_ROOTS_TICKET = 8; // word of the per-build zeroed tickets that counts them
constexpr int COM_MAX_LEVELS = 24;  // levels of a subtree (a 63-bit key has 21)
// Nodes of a local subtree whose links a wave keeps in LDS, one packed word each (a subtree with
// more nodes -- tiny leaves -- reads its links from memory instead).
constexpr int COM_META_CAP = 384;
#ifndef NB_COM_GRID
#define NB_COM_GRID 2048
#endif
constexpr int COM_GRID = NB_COM_GRID;   // workgroups of k_com_t (four waves each; they loop over the local roots)

// Every dependent memory round trip of this kernel is paid by a whole wave, so the loads of a
// step are issued together and waited for once: a subtree's particles go to registers in one
// batch, a level's links (child count, first child, particle range) are read while the level
// is being found and kept in LDS, the (at most eight) children of a node are loaded before the
// first is added.  The order of the additions is untouched.
template <typename R, typename R4>
__global__ __launch_bounds__(256) void k_com_t(const Counts *counts,
                                               const int32_t *__restrict__ o_num_children,
                                               const int32_t *__restrict__ o_first_child,
                                               const int32_t *__restrict__ o_pb,
                                               const int32_t *__restrict__ o_pe,
                                               const int32_t *__restrict__ o_parent,
                                               const R4 *__restrict__ posm, R4 *wsum,
                                               R4 *__restrict__ nodeA, uint32_t *counter,
                                               const uint4 *local_roots,
                                               const uint32_t *n_local_roots, uint32_t *err)
{
    constexpr int COM_LOCAL = com_local<R>();
    constexpr int STAGE = COM_LOCAL / 64;
    static_assert(STAGE == 4 || STAGE == 8, "staging is written out for 256 or 512 particles");
    __shared__ int lv_lo[4][COM_MAX_LEVELS], lv_hi[4][COM_MAX_LEVELS], lv_off[4][COM_MAX_LEVELS];
    __shared__ uint32_t meta[4][COM_META_CAP];
    __shared__ R4 pts[4][COM_LOCAL];
    auto st_wt = [](R *p, R v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto ld_wt = [](const R *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    // particles pb..pe of a leaf, from memory, in index order
    auto leaf_sum = [&](int pb, int pe, R &sx, R &sy, R &sz, R &sm) {
        sx = sy = sz = sm = R(0);
        for (int p = pb; p <= pe; p += 4) {
            R4 q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) q[u] = posm[min(p + u, pe)];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (p + u <= pe) {
                    sx = fma(q[u].w, q[u].x, sx);
                    sy = fma(q[u].w, q[u].y, sy);
                    sz = fma(q[u].w, q[u].z, sz);
                    sm += q[u].w;
                }
        }
    };
    // children fc..fc+nc-1 whose sums this wave has stored itself, in index order
    auto child_sum = [&](int fc, int nc, R &sx, R &sy, R &sz, R &sm) {
        R4 c[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = wsum[fc + min(j, nc - 1)];
        sx = sy = sz = sm = R(0);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < nc) { sx += c[j].x; sy += c[j].y; sz += c[j].z; sm += c[j].w; }
    };
    // sums of a node above the local roots whose last child has just arrived (the caller's
    // counter add has returned): children in index order, all requested before the first is added
    auto sum_arrived = [&](int node, int fc, int nc, R &sx, R &sy, R &sz, R &sm) {
        // leave the counter at 0 for the next build
        __hip_atomic_store(&counter[node], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        R c[8][4];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const R *cs = reinterpret_cast<const R *>(&wsum[fc + min(j, nc - 1)]);
#pragma unroll
            for (int q = 0; q < 4; ++q) c[j][q] = ld_wt(cs + q);
        }
        sx = sy = sz = sm = R(0);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < nc) { sx += c[j][0]; sy += c[j][1]; sz += c[j][2]; sm += c[j][3]; }
    };
    // from node cur, whose sums the calling lane holds, upwards while this lane arrives last
    auto climb = [&](int cur, R sx, R sy, R sz, R sm) {
        int guard = 0;
        while (true) {
            R *ws = reinterpret_cast<R *>(&wsum[cur]);
            st_wt(ws + 0, sx); st_wt(ws + 1, sy); st_wt(ws + 2, sz); st_wt(ws + 3, sm);
            nodeA[cur] = R4{sx / sm, sy / sm, sz / sm, sm};
            const int parent = o_parent[cur];
            if (parent < 0) break;
            if (++guard > 64) {   // deeper than any key allows: the parent chain is corrupt
                atomicOr(err, (uint32_t) DEV_ERR_RANGE);
                break;
            }
            const int nc = o_num_children[parent], fc = o_first_child[parent];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint32_t arrived = __hip_atomic_fetch_add(&counter[parent], 1u, __ATOMIC_RELAXED,
                                                            __HIP_MEMORY_SCOPE_AGENT);
            if ((int) arrived != nc - 1) break;
            sum_arrived(parent, fc, nc, sx, sy, sz, sm);
            cur = parent;
        }
    };

    const int n_oct = counts->n_oct;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // Local roots, one wave each, listed by k_emit as (node, first particle, particles, parent).
    // Two things are carried from one root of a wave to the next so that their round trips are
    // spent working: the next record is requested while the current subtree is summed, and the
    // arrival of a finished root at its parent (drain the stores, bump the counter, look at the
    // count) is issued when the next subtree's first loads have come back and looked at when its
    // levels have been found.
    const int n_roots = (int) __hip_atomic_load(n_local_roots, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int stride = gridDim.x * 4;
    int e = blockIdx.x * 4 + w;
    uint4 rec = local_roots[min(e, max(n_roots - 1, 0))];
    int pend_parent = -1, pend_nc = 0, pend_fc = 0;   // lane 0: arrival still to be made / looked at
    uint32_t pend_arrived = 0;
    bool pend_issued = false;
    auto pend_issue = [&]() {
        if (lane == 0 && pend_parent >= 0) {
            pend_nc = o_num_children[pend_parent];
            pend_fc = o_first_child[pend_parent];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the root's sums have reached memory
            pend_arrived = __hip_atomic_fetch_add(&counter[pend_parent], 1u, __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_AGENT);
        }
        pend_issued = true;
    };
    auto pend_finish = [&]() {
        if (lane == 0 && pend_parent >= 0 && (int) pend_arrived == pend_nc - 1) {
            R sx, sy, sz, sm;
            sum_arrived(pend_parent, pend_fc, pend_nc, sx, sy, sz, sm);
            climb(pend_parent, sx, sy, sz, sm);
        }
        pend_parent = -1;
        pend_issued = false;
    };
    for (; e < n_roots; e += stride) {
        const int root = (int) rec.x, pb0 = (int) rec.y, cnt0 = (int) rec.z, root_parent = (int) rec.w;
        rec = local_roots[min(e + stride, n_roots - 1)];
        // the subtree's particles (one contiguous range) are requested now and go to LDS once
        // its levels have been found
        // (named values, not an array: the array went to scratch memory)
        auto ld_stage = [&](int u) { return posm[pb0 + min(u * 64 + lane, cnt0 - 1)]; };
        const R4 st0 = ld_stage(0), st1 = ld_stage(1), st2 = ld_stage(2), st3 = ld_stage(3);
        R4 st4{}, st5{}, st6{}, st7{};
        if constexpr (STAGE == 8) { st4 = ld_stage(4); st5 = ld_stage(5); st6 = ld_stage(6); st7 = ld_stage(7); }
        // Levels: in the breadth-first numbering the nodes of a level of the subtree are one id
        // range, from the first child of the level's first inner node to the last child of its
        // last.  A node's links are packed on the way: leaf = 1<<31 | (first particle - pb0)<<10 |
        // (particles - 1); inner = (first child - first id of the next level)<<4 | children.
        int K = 0, lo = root, hi = root + 1, off = 0;
        bool cached = true;
        while (true) {
            lv_lo[w][K] = lo;
            lv_hi[w][K] = hi;
            lv_off[w][K] = off;
            const bool fits = off + (hi - lo) <= COM_META_CAP;
            int first = -1, next_lo = 0, next_hi = 0;
            for (int base = lo; base < hi; base += 64) {
                const int i = base + lane;
                const bool in = i < hi;
                int nc = 0, fc = 0, pb = 0, pe = 0;
                if (in) { nc = o_num_children[i]; fc = o_first_child[i]; pb = o_pb[i]; pe = o_pe[i]; }
                const unsigned long long inner = __ballot(in && nc != 0);
                if (inner) {
                    const int fl = __ffsll((long long) inner) - 1, ll = 63 - __clzll((long long) inner);
                    if (first < 0) { first = base + fl; next_lo = __shfl(fc, fl, 64); }
                    next_hi = __shfl(fc, ll, 64) + __shfl(nc, ll, 64);
                }
                if (fits && in)
                    meta[w][off + i - lo] = nc ? (((uint32_t) (fc - next_lo) << 4) | (uint32_t) nc)
                                               : (0x80000000u | ((uint32_t) (pb - pb0) << 10) | (uint32_t) (pe - pb));
                // the previous root's stores are long done by now: its arrival goes out behind
                // this level's loads
                if (!pend_issued) pend_issue();
            }
            cached = cached && fits;
            off += hi - lo;
            if (first < 0) break;
            if (K + 1 >= COM_MAX_LEVELS) {   // deeper than any key allows
                if (lane == 0) atomicOr(err, (uint32_t) DEV_ERR_RANGE);
                break;
            }
            lo = next_lo;
            hi = next_hi;
            ++K;
        }
        pend_finish();
        auto st_stage = [&](int u, const R4 &v) { if (u * 64 + lane < cnt0) pts[w][u * 64 + lane] = v; };
        st_stage(0, st0); st_stage(1, st1); st_stage(2, st2); st_stage(3, st3);
        if constexpr (STAGE == 8) { st_stage(4, st4); st_stage(5, st5); st_stage(6, st6); st_stage(7, st7); }
        // Everything below is read by lanes of the wave that wrote it (LDS here, sums through
        // memory further down): a wave's memory operations are performed in order, so program
        // order is all the fences have to keep.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int k = K; k >= 1; --k) {
            const int l0 = lv_lo[w][k], l1 = lv_hi[w][k], o0 = lv_off[w][k];
            const int child_lo = k < K ? lv_lo[w][k + 1] : 0;
            for (int i = l0 + lane; i < l1; i += 64) {
                bool leaf;
                int a, b;   // leaf: particles a..b of pts; inner: children a..a+b-1
                if (cached) {
                    const uint32_t m = meta[w][o0 + i - l0];
                    leaf = (m >> 31) != 0u;
                    if (leaf) { a = (int) ((m >> 10) & 0x3ffu); b = a + (int) (m & 0x3ffu); }
                    else { a = child_lo + (int) ((m & 0x7fffffffu) >> 4); b = (int) (m & 15u); }
                } else {
                    const int nc = o_num_children[i];
                    leaf = nc == 0;
                    if (leaf) { a = o_pb[i] - pb0; b = o_pe[i] - pb0; }
                    else { a = o_first_child[i]; b = nc; }
                }
                R sx, sy, sz, sm;
                if (leaf) {
                    sx = sy = sz = sm = R(0);
                    for (int p = a; p <= b; p += 4) {
                        R4 q[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) q[u] = pts[w][min(p + u, b)];
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (p + u <= b) {
                                sx = fma(q[u].w, q[u].x, sx);
                                sy = fma(q[u].w, q[u].y, sy);
                                sz = fma(q[u].w, q[u].z, sz);
                                sm += q[u].w;
                            }
                    }
                } else {
                    child_sum(a, b, sx, sy, sz, sm);
                }
                wsum[i] = R4{sx, sy, sz, sm};
                nodeA[i] = R4{sx / sm, sy / sm, sz / sm, sm};
            }
            // this level's sums, stored by some lanes, are read by others one level up
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (lane == 0) {
            int fc, nc;
            if (cached) { const uint32_t m = meta[w][0]; fc = lv_lo[w][1] + (int) (m >> 4); nc = (int) (m & 15u); }
            else { fc = o_first_child[root]; nc = o_num_children[root]; }
            R sx, sy, sz, sm;
            child_sum(fc, nc, sx, sy, sz, sm);
            // the root's sums go out write-through; its arrival at the parent is made from the
            // next iteration (or below, after the last)
            R *ws = reinterpret_cast<R *>(&wsum[root]);
            st_wt(ws + 0, sx); st_wt(ws + 1, sy); st_wt(ws + 2, sz); st_wt(ws + 3, sm);
            nodeA[root] = R4{sx / sm, sy / sm, sz / sm, sm};
            pend_parent = root_parent;
        }
        __builtin_amdgcn_wave_barrier();   // pts[] and meta[] are reused by the next local root
    }
    if (!pend_issued) pend_issue();
    pend_finish();
    // leaves directly under a node above the local roots
    for (int o = blockIdx.x * 256 + threadIdx.x; o < n_oct; o += gridDim.x * 256) {
        if (o_num_children[o] != 0) continue;
        const int par = o_parent[o];
        if (par >= 0 && o_pe[par] - o_pb[par] + 1 <= COM_LOCAL) continue;   // inside a local subtree
        R sx, sy, sz, sm;
        leaf_sum(o_pb[o], o_pe[o], sx, sy, sz, sm);
        climb(o, sx, sy, sz, sm);
    }
}

}  // namespace nb

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

template <typename R, typename R4>
__global__ __launch_bounds__(256) void k_com_t(const Counts *counts,
                                               const int32_t *__restrict__ o_num_children,
                                               const int32_t *__restrict__ o_first_child,
                                               const int32_t *__restrict__ o_pb,
                                               const int32_t *__restrict__ o_pe,
                                               const int32_t *__restrict__ o_parent,
                                               const R4 *__restrict__ posm, R4 *wsum,
                                               R4 *__restrict__ nodeA, uint32_t *counter,
                                               const uint32_t *local_roots,
                                               const uint32_t *n_local_roots, uint32_t *err)
{
    constexpr int COM_LOCAL = com_local<R>();
    __shared__ int lv_lo[4][COM_MAX_LEVELS], lv_hi[4][COM_MAX_LEVELS];
    __shared__ R4 pts[4][COM_LOCAL];
    auto st_wt = [](R *p, R v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto ld_wt = [](const R *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto leaf_sum = [&](int o, R &sx, R &sy, R &sz, R &sm) {
        sx = sy = sz = sm = R(0);
        const int pb = o_pb[o], pe = o_pe[o];
        for (int p = pb; p <= pe; ++p) {
            const R4 q = posm[p];
            sx = fma(q.w, q.x, sx);
            sy = fma(q.w, q.y, sy);
            sz = fma(q.w, q.z, sz);
            sm += q.w;
        }
    };
    auto child_sum = [&](int i, R &sx, R &sy, R &sz, R &sm) {   // children written by this wave
        const int fc = o_first_child[i], nc = o_num_children[i];
        sx = sy = sz = sm = R(0);
        for (int c = fc; c < fc + nc; ++c) {
            const R4 w = wsum[c];
            sx += w.x; sy += w.y; sz += w.z; sm += w.w;
        }
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
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint32_t arrived = __hip_atomic_fetch_add(&counter[parent], 1u, __ATOMIC_RELAXED,
                                                            __HIP_MEMORY_SCOPE_AGENT);
            const int nc = o_num_children[parent];
            if ((int) arrived != nc - 1) break;
            // every child has arrived: leave the counter at 0 for the next build
            __hip_atomic_store(&counter[parent], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int fc = o_first_child[parent];
            sx = sy = sz = sm = R(0);
            for (int c = fc; c < fc + nc; ++c) {
                const R *cs = reinterpret_cast<const R *>(&wsum[c]);
                sx += ld_wt(cs + 0);
                sy += ld_wt(cs + 1);
                sz += ld_wt(cs + 2);
                sm += ld_wt(cs + 3);
            }
            cur = parent;
        }
    };

    const int n_oct = counts->n_oct;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // local roots, one wave each
    const int n_roots = (int) __hip_atomic_load(n_local_roots, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int e = blockIdx.x * 4 + w; e < n_roots; e += gridDim.x * 4) {
        const int root = (int) local_roots[e];
        // the subtree's particles (one contiguous range) go to LDS while its levels are found
        const int pb0 = o_pb[root], cnt0 = o_pe[root] - pb0 + 1;
        for (int j = lane; j < cnt0; j += 64) pts[w][j] = posm[pb0 + j];
        int K = 0, lo = root, hi = root + 1;
        while (true) {
            lv_lo[w][K] = lo;
            lv_hi[w][K] = hi;
            int first = -1, last = -1;
            for (int base = lo; base < hi; base += 64) {
                const int i = base + lane;
                const unsigned long long inner = __ballot(i < hi && o_num_children[i] != 0);
                if (inner) {
                    if (first < 0) first = base + __ffsll((long long) inner) - 1;
                    last = base + 63 - __clzll((long long) inner);
                }
            }
            if (first < 0) break;
            if (K + 1 >= COM_MAX_LEVELS) {   // deeper than any key allows
                if (lane == 0) atomicOr(err, (uint32_t) DEV_ERR_RANGE);
                break;
            }
            lo = o_first_child[first];
            hi = o_first_child[last] + o_num_children[last];
            ++K;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // pts[] written by other lanes
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        for (int k = K; k >= 1; --k) {
            const int l0 = lv_lo[w][k], l1 = lv_hi[w][k];
            for (int i = l0 + lane; i < l1; i += 64) {
                R sx, sy, sz, sm;
                if (o_num_children[i] == 0) {
                    sx = sy = sz = sm = R(0);
                    const int pe = o_pe[i] - pb0;
                    for (int p = o_pb[i] - pb0; p <= pe; ++p) {
                        const R4 q = pts[w][p];
                        sx = fma(q.w, q.x, sx);
                        sy = fma(q.w, q.y, sy);
                        sz = fma(q.w, q.z, sz);
                        sm += q.w;
                    }
                } else {
                    child_sum(i, sx, sy, sz, sm);
                }
                wsum[i] = R4{sx, sy, sz, sm};
                nodeA[i] = R4{sx / sm, sy / sm, sz / sm, sm};
            }
            // this level's sums, stored by some lanes, are read by others one level up
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        if (lane == 0) {
            R sx, sy, sz, sm;
            child_sum(root, sx, sy, sz, sm);
            climb(root, sx, sy, sz, sm);
        }
        __builtin_amdgcn_wave_barrier();   // pts[] is reused by the next local root
    }
    // leaves directly under a node above the local roots
    for (int o = blockIdx.x * 256 + threadIdx.x; o < n_oct; o += gridDim.x * 256) {
        if (o_num_children[o] != 0) continue;
        const int par = o_parent[o];
        if (par >= 0 && o_pe[par] - o_pb[par] + 1 <= COM_LOCAL) continue;   // inside a local subtree
        R sx, sy, sz, sm;
        leaf_sum(o, sx, sy, sz, sm);
        climb(o, sx, sy, sz, sm);
    }
}

}  // namespace nb

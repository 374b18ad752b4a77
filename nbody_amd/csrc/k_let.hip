// k_let.hip -- kernels of the multi-GPU locally-essential-tree exchange: bounding box of a
// handle's particles, export of the sources a remote domain needs from this handle's tree,
// and unpacking of received (x,y,z,m) records into a source-only handle.
//
// The reference is single-GPU and has none of this; the scheme follows the hybrid
// distributed-octree idea its README cites (Cazalbou et al., PASC'24): every GPU owns a
// domain, builds its own tree, and ships to every other domain only the part of that tree the
// other domain's bodies can ever touch.
#include "dev_utils.hpp"
#include "nbody_internal.hpp"

namespace nb {

// positions are >= 0 (reflecting walls), so float order == unsigned order of the bit patterns
__global__ __launch_bounds__(256) void k_bbox(int n, const Float4 *__restrict__ posm,
                                              uint32_t *__restrict__ box /* lo xyz, hi xyz */)
{
    __shared__ float s[6][4];
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {0.f, 0.f, 0.f};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const Float4 p = posm[i];
        lo[0] = fminf(lo[0], p.x); lo[1] = fminf(lo[1], p.y); lo[2] = fminf(lo[2], p.z);
        hi[0] = fmaxf(hi[0], p.x); hi[1] = fmaxf(hi[1], p.y); hi[2] = fmaxf(hi[2], p.z);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], o, 64));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], o, 64));
        }
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 3; ++k) { s[k][w] = lo[k]; s[3 + k][w] = hi[k]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        float v = s[k][0];
        for (int i = 1; i < 4; ++i) v = k < 3 ? fminf(v, s[k][i]) : fmaxf(v, s[k][i]);
        if (k < 3) atomicMin(&box[k], __float_as_uint(v)); else atomicMax(&box[k], __float_as_uint(v));
    }
}

// (x,y,z,w) of either precision as floats; bounds of a coordinate as floats that enclose it
__device__ __forceinline__ Float4 narrow4(const Float4 v) { return v; }
__device__ __forceinline__ Float4 narrow4(const Double4 v) { return Float4{(float) v.x, (float) v.y, (float) v.z, (float) v.w}; }
__device__ __forceinline__ float bound_lo(float v) { return v; }
__device__ __forceinline__ float bound_hi(float v) { return v; }
__device__ __forceinline__ float bound_lo(double v) { return __double2float_rd(v); }
__device__ __forceinline__ float bound_hi(double v) { return __double2float_ru(v); }

__device__ __forceinline__ float box_dist_sq(const float *__restrict__ bx, float cx, float cy,
                                             float cz)
{
    const float dx = fmaxf(0.f, fmaxf(bx[0] - cx, cx - bx[3]));
    const float dy = fmaxf(0.f, fmaxf(bx[1] - cy, cy - bx[4]));
    const float dz = fmaxf(0.f, fmaxf(bx[2] - cz, cz - bx[5]));
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

// ---------------------------------------------------------------------------
// Top of the tree ("frontier") of a handle: at most T nodes that together hold every
// particle exactly once -- what a domain tells the other domains about itself.  The nodes
// of the breadth-first prefix [0, K) are expanded, K the largest prefix whose expansion
// leaves at most T nodes: the frontier is every child of an expanded node that is not itself
// expanded, plus the leaves among the expanded ones.  Records are ordered by first particle
// so that they tile [0, n).  A record is LET_REC floats:
//   lo xyz, hi xyz (tight box of the node's particles, filled by k_frontier_bbox),
//   centre of mass xyz, mass, cell size, particle count (as float).
// Unused records are empty boxes (lo = 3e38, hi = 0), which are farther than anything.
// One workgroup of 256 threads; T <= LET_MAX_T = 32.
// ---------------------------------------------------------------------------
template <typename R4>
__global__ __launch_bounds__(256) void k_frontier_select(
    const Counts *counts, int n, int T, const NodeB *__restrict__ nodeB,
    const R4 *__restrict__ nodeA, const int32_t *__restrict__ o_pb,
    const int32_t *__restrict__ o_pe, float *__restrict__ rec, int32_t *__restrict__ seg)
{
    __shared__ uint32_t s_scan[5];
    __shared__ int s_size[LET_MAX_T + 1];   // frontier size after expanding the prefix [0, K)
    __shared__ int s_K, s_C, s_F;
    __shared__ int s_node[LET_MAX_T], s_first[LET_MAX_T];
    const int tid = threadIdx.x;
    const int n_oct = counts->n_oct;
    const int L = min(n_oct, T);
    // growth of the frontier when node t is expanded: children - 1 (a leaf stays as it is)
    uint32_t inc = 0;
    if (tid < L) {
        const NodeB b = nodeB[tid];
        inc = (b.meta & NODE_LEAF) ? 0u : (uint32_t) (b.count - 1);
    }
    uint32_t tot;
    const uint32_t excl = block_excl_scan_u32<4>(inc, s_scan, &tot);
    if (tid <= L) s_size[tid] = 1 + (int) excl;   // size after expanding [0, tid)
    if (tid == 0) { s_K = 0; s_C = 1; s_F = 0; }
    __syncthreads();
    if (tid <= L && s_size[tid] <= T) atomicMax(&s_K, tid);   // sizes do not decrease with K
    __syncthreads();
    const int K = s_K;
    // nodes [0, C) are the root and all children of expanded nodes (breadth-first numbering)
    if (tid < K) {
        const NodeB b = nodeB[tid];
        if (!(b.meta & NODE_LEAF)) atomicMax(&s_C, b.first + b.count);
    }
    __syncthreads();
    const int C = s_C;
    for (int base = 0; base < C; base += 256) {   // C <= 1 + 8 T: two rounds at most
        const int o = base + tid;
        bool in = false;
        if (o < C) in = o >= K || (nodeB[o].meta & NODE_LEAF);
        uint32_t t2;
        const uint32_t ex = block_excl_scan_u32<4>(in ? 1u : 0u, s_scan, &t2);
        const int at = s_F + (int) ex;
        if (in && at < LET_MAX_T) { s_node[at] = o; s_first[at] = o_pb[o]; }
        __syncthreads();
        if (tid == 0) s_F += (int) t2;
        __syncthreads();
    }
    const int F = min(s_F, T);   // == s_F by construction
    if (tid < T) {
        float *r = rec + (size_t) tid * LET_REC;
        if (tid < F) {
            // slot = rank of this node's first particle among the frontier nodes
            int slot = 0;
            for (int j = 0; j < F; ++j) slot += s_first[j] < s_first[tid];
            r = rec + (size_t) slot * LET_REC;
            const int o = s_node[tid];
            const Float4 A = narrow4(nodeA[o]);
            r[0] = r[1] = r[2] = 3.0e38f;
            r[3] = r[4] = r[5] = 0.f;
            r[6] = A.x; r[7] = A.y; r[8] = A.z; r[9] = A.w;
            r[10] = nodeB[o].size;
            r[11] = (float) (o_pe[o] - o_pb[o] + 1);
            seg[slot] = s_first[tid];
        } else {
            r[0] = r[1] = r[2] = 3.0e38f;
            for (int k = 3; k < LET_REC; ++k) r[k] = 0.f;
        }
    }
    if (tid == 0) { seg[F] = n; seg[LET_MAX_T + 1] = F; }
}

// Tight boxes of the frontier nodes: one pass over the particles, each finds its node by
// binary search over the <= T first-particle indices; a wave whose 64 particles share a node
// (nearly all do) reduces first and issues one atomic per bound.  Positions are >= 0, so the
// order of the floats is the order of their bit patterns.
template <typename R4>
__global__ __launch_bounds__(256) void k_frontier_bbox(int n, const R4 *__restrict__ posm,
                                                       const int32_t *__restrict__ seg,
                                                       float *__restrict__ rec)
{
    __shared__ int s_seg[LET_MAX_T + 1];
    __shared__ uint32_t s_box[LET_MAX_T][6];
    const int tid = threadIdx.x;
    const int F = seg[LET_MAX_T + 1];
    if (tid <= F) s_seg[tid] = seg[tid];
    for (int j = tid; j < LET_MAX_T * 6; j += 256) (&s_box[0][0])[j] = (j % 6) < 3 ? 0x7f7fffffu : 0u;
    __syncthreads();
    // grid-stride over whole 256-particle chunks: a few hundred workgroups, each of which ends
    // with one global atomic per bound of every node it met
    for (int base = blockIdx.x * 256; base < n; base += gridDim.x * 256) {
        const int i = base + tid;
        const bool valid = i < n;
        int sg = 0;
        uint32_t v[3] = {0, 0, 0}, u[3] = {0, 0, 0};   // lower and upper float bounds of the coordinates
        if (valid) {
            const R4 p = posm[i];
            v[0] = __float_as_uint(bound_lo(p.x)); v[1] = __float_as_uint(bound_lo(p.y)); v[2] = __float_as_uint(bound_lo(p.z));
            u[0] = __float_as_uint(bound_hi(p.x)); u[1] = __float_as_uint(bound_hi(p.y)); u[2] = __float_as_uint(bound_hi(p.z));
            int lo = 0, hi = F;   // largest sg with s_seg[sg] <= i
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (s_seg[mid] <= i) lo = mid; else hi = mid;
            }
            sg = lo;
        }
        const int sg0 = __builtin_amdgcn_readfirstlane(sg);
        if (__all(valid && sg == sg0)) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                uint32_t mn = v[k], mx = u[k];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    mn = min(mn, (uint32_t) __shfl_xor((int) mn, o, 64));
                    mx = max(mx, (uint32_t) __shfl_xor((int) mx, o, 64));
                }
                if ((tid & 63) == 0) { atomicMin(&s_box[sg0][k], mn); atomicMax(&s_box[sg0][3 + k], mx); }
            }
        } else if (valid) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { atomicMin(&s_box[sg][k], v[k]); atomicMax(&s_box[sg][3 + k], u[k]); }
        }
    }
    __syncthreads();
    for (int j = tid; j < F * 6; j += 256) {
        const int sl = j / 6, k = j % 6;
        const uint32_t b = s_box[sl][k];
        uint32_t *dst = reinterpret_cast<uint32_t *>(rec + (size_t) sl * LET_REC + k);
        if (k < 3) { if (b != 0x7f7fffffu) atomicMin(dst, b); }
        else if (s_box[sl][k - 3] != 0x7f7fffffu) atomicMax(dst, b);
    }
}

// ---------------------------------------------------------------------------
// Export of the locally-essential tree.  A remote domain d is described by its frontier
// boxes; the opening criterion of this tree's node a with respect to d is the traversal's
// own  size^2 / dist^2 < theta^2  with dist = the smallest distance from any of d's boxes to
// a's centre of mass -- never larger than the distance from any body of d (every body lies in
// one of the boxes), so whatever passes here would pass for every group of d as well.
// k_let_mask evaluates the criterion once per (inner node, domain) into a bit mask; the
// enclosing box of the domain is tried first (it passes => every box passes).
// k_let_emit, one thread per (node, domain): a node is REACHED iff none of its ancestors
// passes; a reached node exports (com, mass) if it passes itself, all its particles if it is
// a leaf, nothing otherwise (its children decide).  No stack, no queue; output slots come
// from one atomic per workgroup and domain.
// ---------------------------------------------------------------------------
template <typename R4>
__global__ __launch_bounds__(256) void k_let_mask(const Counts *counts,
                                                  const R4 *__restrict__ nodeA,
                                                  const NodeB *__restrict__ nodeB,
                                                  const float *__restrict__ frontiers, int ndom,
                                                  int T, int skip, float theta,
                                                  uint32_t *__restrict__ mask)
{
    extern __shared__ float s_geo[];   // [ndom][T][6] boxes, then [ndom][6] enclosing boxes
    float *s_hull = s_geo + (size_t) ndom * T * 6;
    const int tid = threadIdx.x;
    for (int j = tid; j < ndom * T * 6; j += 256)
        s_geo[j] = frontiers[(size_t) (j / 6) * LET_REC + (j % 6)];
    __syncthreads();
    for (int j = tid; j < ndom * 6; j += 256) {
        const int d = j / 6, k = j % 6;
        float v = s_geo[(size_t) d * T * 6 + k];
        for (int b = 1; b < T; ++b) {
            const float w = s_geo[((size_t) d * T + b) * 6 + k];
            const bool empty = s_geo[((size_t) d * T + b) * 6] > 1.0e38f;
            if (!empty) v = k < 3 ? fminf(v, w) : fmaxf(v, w);
        }
        s_hull[j] = v;
    }
    __syncthreads();
    const int n_oct = counts->n_oct;
    const float th2 = theta * theta;
    for (int o = blockIdx.x * 256 + tid; o < n_oct; o += gridDim.x * 256) {
        const NodeB B = nodeB[o];
        uint32_t bits = 0;
        if (!(B.meta & NODE_LEAF)) {
            const Float4 A = narrow4(nodeA[o]);   // (the criterion in float for double handles too)
            const float sz2 = B.size * B.size;
            for (int d = 0; d < ndom; ++d) {
                if (d == skip) continue;
                bool pass = __fdiv_rn(sz2, box_dist_sq(s_hull + 6 * d, A.x, A.y, A.z)) < th2;
                if (!pass) {
                    float dmin = 3.0e38f;
                    const float *g = s_geo + (size_t) d * T * 6;
                    for (int b = 0; b < T; ++b) dmin = fminf(dmin, box_dist_sq(g + 6 * b, A.x, A.y, A.z));
                    pass = __fdiv_rn(sz2, dmin) < th2;
                }
                bits |= (pass ? 1u : 0u) << d;
            }
        }
        mask[o] = bits;
    }
}

__global__ __launch_bounds__(256) void k_let_emit(
    const Counts *counts, const Float4 *__restrict__ nodeA, const NodeB *__restrict__ nodeB,
    const int32_t *__restrict__ o_parent, const Float4 *__restrict__ posm,
    const uint32_t *__restrict__ mask, int skip, Float4 *__restrict__ out, int cap,
    uint32_t *__restrict__ out_count)
{
    __shared__ uint32_t s_scan[5];
    __shared__ uint32_t s_base;
    const int n_oct = counts->n_oct;
    const int dom = blockIdx.y;
    if (dom == skip) return;
    const int o = blockIdx.x * 256 + threadIdx.x;
    uint32_t cnt = 0;
    bool whole = false;
    NodeB me{};
    if (o < n_oct) {
        bool reached = true;
        for (int a = o_parent[o]; a >= 0; a = o_parent[a])
            if ((mask[a] >> dom) & 1u) { reached = false; break; }
        if (reached) {
            me = nodeB[o];
            if (me.meta & NODE_LEAF) cnt = (uint32_t) me.count;
            else if ((mask[o] >> dom) & 1u) { whole = true; cnt = 1; }
        }
    }
    uint32_t tot;
    const uint32_t excl = block_excl_scan_u32<4>(cnt, s_scan, &tot);
    if (threadIdx.x == 0) s_base = tot ? atomicAdd(&out_count[dom], tot) : 0u;
    __syncthreads();
    if (cnt == 0) return;
    const size_t pos = (size_t) s_base + excl;
    if (pos + cnt > (size_t) cap) return;  // over capacity: the caller sees out_count > cap
    Float4 *dst = out + (size_t) dom * cap + pos;
    if (whole) dst[0] = nodeA[o];
    else
        for (int k = 0; k < me.count; ++k) dst[k] = posm[me.first + k];
}

// ---------------------------------------------------------------------------
// Export of the locally-essential tree AS A TREE.  What a remote domain d needs from this tree
// is the set of REACHED nodes (no ancestor passes d's opening criterion): a pruned copy of the
// tree whose leaves are this tree's reached leaves (with their particles) and the reached inner
// nodes that pass (one (com, mass) record each, the node "as a particle").  The receiver can
// hang that copy beside its own tree and walk both in one traversal -- no second sort, build
// or centre-of-mass pass over the received records.
//
// Numbering.  Nodes are numbered breadth-first with the children of a node contiguous and the
// child blocks of a level in the order of their parents.  The reached nodes are the root and the
// children of the reached inner nodes that do not pass ("kept" nodes), so, in the same order,
//   index of node o in the copy            R(o) = number of reached nodes before o
//   index of the first child of a kept o   1 + S(o),  S(o) = sum of the child counts of the kept
//                                                      nodes before o
//   index of the first particle record     P(o) = particle records of the reached nodes before o
// are three exclusive prefix sums in node order: k_let_tree_count (per workgroup and domain),
// k_let_tree_scan (over the workgroups), k_let_tree_emit.  Per domain the output is one segment:
// node records (32 bytes: com xyz, mass | size, first, count, meta -- the traversal's own two
// records, `first` relative to the copy), then the particle records (16 bytes).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_let_reach(const Counts *counts,
                                                   const int32_t *__restrict__ o_parent,
                                                   const uint32_t *__restrict__ mask,
                                                   uint32_t *__restrict__ reach)
{
    const int n_oct = counts->n_oct;
    for (int o = blockIdx.x * 256 + threadIdx.x; o < n_oct; o += gridDim.x * 256) {
        uint32_t blocked = 0;
        for (int a = o_parent[o]; a >= 0; a = o_parent[a]) blocked |= mask[a];
        reach[o] = ~blocked;
    }
}

// what node o contributes to domain `dom`'s copy: x = 1 | child count << 12 (kept nodes), y =
// particle records
__device__ __forceinline__ uint2 let_tree_contrib(int o, int n_oct, int dom,
                                                  const uint32_t *__restrict__ reach,
                                                  const uint32_t *__restrict__ mask,
                                                  const NodeB *__restrict__ nodeB, NodeB *me,
                                                  int *kind)
{
    *kind = 0;
    if (o >= n_oct || !((reach[o] >> dom) & 1u)) return uint2{0u, 0u};
    *me = nodeB[o];
    if (me->meta & NODE_LEAF) { *kind = 1; return uint2{1u, (uint32_t) me->count}; }
    if ((mask[o] >> dom) & 1u) { *kind = 2; return uint2{1u, 1u}; }
    *kind = 3;
    return uint2{1u | ((uint32_t) me->count << 12), 0u};
}

__global__ __launch_bounds__(256) void k_let_tree_count(const Counts *counts,
                                                        const NodeB *__restrict__ nodeB,
                                                        const uint32_t *__restrict__ mask,
                                                        const uint32_t *__restrict__ reach,
                                                        int skip, int nblk,
                                                        uint32_t *__restrict__ bsum /* [dom][nblk][3] */)
{
    __shared__ uint32_t s_acc[3];
    const int dom = blockIdx.y;
    if (dom == skip) return;
    const int n_oct = counts->n_oct;
    if ((int) blockIdx.x * 256 >= n_oct) return;
    if (threadIdx.x < 3) s_acc[threadIdx.x] = 0;
    __syncthreads();
    NodeB me;
    int kind;
    const uint2 c = let_tree_contrib(blockIdx.x * 256 + threadIdx.x, n_oct, dom, reach, mask, nodeB, &me, &kind);
    const uint32_t a = wave_sum_u32(c.x & 0xfffu), k = wave_sum_u32(c.x >> 12), p = wave_sum_u32(c.y);
    if ((threadIdx.x & 63) == 0) { atomicAdd(&s_acc[0], a); atomicAdd(&s_acc[1], k); atomicAdd(&s_acc[2], p); }
    __syncthreads();
    if (threadIdx.x < 3) bsum[((size_t) dom * nblk + blockIdx.x) * 3 + threadIdx.x] = s_acc[threadIdx.x];
}

// One workgroup per domain: exclusive scan of the per-workgroup sums in place, totals to
// out_count[2 dom] (nodes) and [2 dom + 1] (particle records).
__global__ __launch_bounds__(256) void k_let_tree_scan(const Counts *counts, int skip, int nblk,
                                                       uint32_t *__restrict__ bsum,
                                                       uint32_t *__restrict__ out_count)
{
    __shared__ uint32_t s_scan[5];
    const int dom = blockIdx.x;
    if (dom == skip) { if (threadIdx.x < 2) out_count[2 * dom + threadIdx.x] = 0; return; }
    const int used = min(nblk, (counts->n_oct + 255) / 256);
    uint32_t carry[3] = {0, 0, 0};
    for (int base = 0; base < used; base += 256) {
        const int b = base + (int) threadIdx.x;
        uint32_t *q = bsum + ((size_t) dom * nblk + b) * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const uint32_t v = b < used ? q[k] : 0u;
            uint32_t tot;
            const uint32_t ex = block_excl_scan_u32<4>(v, s_scan, &tot);
            if (b < used) q[k] = carry[k] + ex;
            carry[k] += tot;
        }
    }
    if (threadIdx.x == 0) { out_count[2 * dom] = carry[0]; out_count[2 * dom + 1] = carry[2]; }
}

template <typename R4>
__global__ __launch_bounds__(256) void k_let_tree_emit(
    const Counts *counts, const R4 *__restrict__ nodeA, const NodeB *__restrict__ nodeB,
    const R4 *__restrict__ posm, const uint32_t *__restrict__ mask,
    const uint32_t *__restrict__ reach, int skip, int nblk, const uint32_t *__restrict__ bsum,
    const uint32_t *__restrict__ out_count, unsigned char *__restrict__ out, size_t cap_bytes)
{
    __shared__ uint32_t s_scan[5];
    const int dom = blockIdx.y;
    if (dom == skip) return;
    const int n_oct = counts->n_oct;
    if ((int) blockIdx.x * 256 >= n_oct) return;
    constexpr size_t NODE_BYTES = sizeof(R4) + sizeof(NodeB), PART_BYTES = sizeof(R4);
    const uint32_t tot_nodes = out_count[2 * dom], tot_parts = out_count[2 * dom + 1];
    // over capacity: nothing is written, the caller sees the counts and comes back with more room
    if ((size_t) tot_nodes * NODE_BYTES + (size_t) tot_parts * PART_BYTES > cap_bytes) return;
    const int o = blockIdx.x * 256 + threadIdx.x;
    NodeB me;
    int kind;
    const uint2 c = let_tree_contrib(o, n_oct, dom, reach, mask, nodeB, &me, &kind);
    uint32_t t0, t1;
    const uint32_t ex0 = block_excl_scan_u32<4>(c.x, s_scan, &t0);   // nodes | kept children << 12
    const uint32_t ex1 = block_excl_scan_u32<4>(c.y, s_scan, &t1);
    if (kind == 0) return;
    const uint32_t *b3 = bsum + ((size_t) dom * nblk + blockIdx.x) * 3;
    const uint32_t R = b3[0] + (ex0 & 0xfffu), S = b3[1] + (ex0 >> 12), P = b3[2] + ex1;
    unsigned char *seg = out + (size_t) dom * cap_bytes;
    R4 *nA = reinterpret_cast<R4 *>(seg + (size_t) R * NODE_BYTES);
    NodeB *nB = reinterpret_cast<NodeB *>(seg + (size_t) R * NODE_BYTES + sizeof(R4));
    R4 *parts = reinterpret_cast<R4 *>(seg + (size_t) tot_nodes * NODE_BYTES);
    const R4 A = nodeA[o];
    *nA = A;
    if (kind == 3) {
        // kept: its children follow.  Marked as "leaf child" (taken without a test) are the
        // children that pass for the whole domain; real leaves are left to the receiver's own
        // test, so that its far groups take them as one source instead of particle by particle.
        uint32_t lm = 0;
        for (int k = 0; k < me.count; ++k) lm |= ((mask[me.first + k] >> dom) & 1u) << k;
        *nB = NodeB{me.size, (int32_t) (1u + S), me.count, lm};
    } else if (kind == 2) {   // passes: the node as one particle
        *nB = NodeB{me.size, (int32_t) P, 1, NODE_LEAF};
        parts[P] = A;
    } else {
        *nB = NodeB{me.size, (int32_t) P, me.count, NODE_LEAF};
        for (int k = 0; k < me.count; ++k) parts[P + k] = posm[me.first + k];
    }
}

// Received copies -> the handle's own node and particle arrays, behind its own tree: copy r
// (nn[r] node records, then np[r] particle records, at byte offset off[r] of `recv`) gets the
// node indices [node_base[r], +nn[r]) and the particle indices [part_base[r], +np[r]); `first`
// of every node is rebased accordingly.  desc = 5 ints per copy: off/16, nn, np, node_base, part_base;
// item_prefix[r] = nodes + particles of the copies before r.
template <typename R4>
__global__ __launch_bounds__(256) void k_let_graft(int ncopies, const int32_t *__restrict__ desc,
                                                   const int32_t *__restrict__ item_prefix,
                                                   const unsigned char *__restrict__ recv,
                                                   R4 *__restrict__ nodeA, NodeB *__restrict__ nodeB,
                                                   R4 *__restrict__ posm, float *__restrict__ posm2)
{
    constexpr size_t NODE_BYTES = sizeof(R4) + sizeof(NodeB), PART_BYTES = sizeof(R4);
    const int total = item_prefix[ncopies];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        int lo = 0, hi = ncopies;   // largest r with item_prefix[r] <= i
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (item_prefix[mid] <= i) lo = mid; else hi = mid;
        }
        const int32_t *d = desc + 5 * lo;
        const int j = i - item_prefix[lo];
        const unsigned char *src = recv + (size_t) d[0] * 16;
        if (j < d[1]) {
            const R4 A = *reinterpret_cast<const R4 *>(src + (size_t) j * NODE_BYTES);
            NodeB B = *reinterpret_cast<const NodeB *>(src + (size_t) j * NODE_BYTES + sizeof(R4));
            B.first += (B.meta & NODE_LEAF) ? d[4] : d[3];
            nodeA[d[3] + j] = A;
            nodeB[d[3] + j] = B;
        } else {
            const int k = j - d[1];
            const R4 p = *reinterpret_cast<const R4 *>(src + (size_t) d[1] * NODE_BYTES + (size_t) k * PART_BYTES);
            const size_t q = (size_t) d[4] + k;
            posm[q] = p;
            if (posm2) {   // float handles: the pair-interleaved copy the traversal streams
                float *r = posm2 + (q >> 1) * 8 + (q & 1);
                r[0] = (float) p.x; r[2] = (float) p.y; r[4] = (float) p.z; r[6] = (float) p.w;
            }
        }
    }
}

// (x,y,z,m) records -> the SoA arrays of a source-only handle.
__global__ __launch_bounds__(256) void k_unpack_sources(int n, const Float4 *__restrict__ src,
                                                        float *__restrict__ x, float *__restrict__ y,
                                                        float *__restrict__ z, float *__restrict__ m,
                                                        uint32_t *__restrict__ id)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const Float4 p = src[i];
    x[i] = p.x; y[i] = p.y; z[i] = p.z; m[i] = p.w;
    id[i] = (uint32_t) i;
}

// ---------------------------------------------------------------------------
// Re-balancing of a Morton-range decomposition: positions of query keys in the handle's sorted
// keys (histograms of key prefixes and the cut positions of the splitters are differences of
// these), and the 44-byte particle record that migrates between domains:
// x, y, z, vx, vy, vz, m, ax, ay, az, id.
// ---------------------------------------------------------------------------
template <typename KeyT>
__global__ __launch_bounds__(256) void k_key_lower_bound(const KeyT *__restrict__ sorted, int n,
                                                         const uint64_t *__restrict__ queries, int nq,
                                                         int key_bits, int32_t *__restrict__ out)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= nq) return;
    const uint64_t q = queries[j];
    int lo = 0, hi = n;   // first index whose key is >= q; a query of 2^key_bits is "past the end"
    if ((q >> key_bits) != 0) lo = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((uint64_t) sorted[mid] < q) lo = mid + 1; else hi = mid;
    }
    out[j] = lo;
}

constexpr int MIG_WORDS = 11;   // of the handle's precision: 44 bytes (float), 88 (double)

__device__ __forceinline__ float id_as_word(uint32_t id, float) { return __uint_as_float(id); }
__device__ __forceinline__ double id_as_word(uint32_t id, double) { return __longlong_as_double((long long) id); }
__device__ __forceinline__ uint32_t word_as_id(float w) { return __float_as_uint(w); }
__device__ __forceinline__ uint32_t word_as_id(double w) { return (uint32_t) __double_as_longlong(w); }

template <typename R>
__global__ __launch_bounds__(256) void k_pack_particles(
    int n, const R *__restrict__ x, const R *__restrict__ y, const R *__restrict__ z,
    const R *__restrict__ vx, const R *__restrict__ vy, const R *__restrict__ vz,
    const R *__restrict__ m, const R *__restrict__ ax, const R *__restrict__ ay,
    const R *__restrict__ az, const uint32_t *__restrict__ id, R *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    R *r = out + (size_t) i * MIG_WORDS;
    r[0] = x[i]; r[1] = y[i]; r[2] = z[i]; r[3] = vx[i]; r[4] = vy[i]; r[5] = vz[i]; r[6] = m[i];
    r[7] = ax[i]; r[8] = ay[i]; r[9] = az[i]; r[10] = id_as_word(id[i], R(0));
}

// The accelerations ride in the half-step velocity arrays (dead between two steps) so that
// the permute of the following sort carries them along.
template <typename R>
__global__ __launch_bounds__(256) void k_unpack_particles(
    int n, const R *__restrict__ in, R *__restrict__ x, R *__restrict__ y,
    R *__restrict__ z, R *__restrict__ vx, R *__restrict__ vy, R *__restrict__ vz,
    R *__restrict__ m, R *__restrict__ hx, R *__restrict__ hy, R *__restrict__ hz,
    uint32_t *__restrict__ id)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const R *r = in + (size_t) i * MIG_WORDS;
    x[i] = r[0]; y[i] = r[1]; z[i] = r[2]; vx[i] = r[3]; vy[i] = r[4]; vz[i] = r[5]; m[i] = r[6];
    hx[i] = r[7]; hy[i] = r[8]; hz[i] = r[9]; id[i] = word_as_id(r[10]);
}

hipError_t launch_key_lower_bound(Sim *s, const uint64_t *queries_dev, int nq, int32_t *out_dev)
{
    if (nq <= 0) return hipSuccess;
    const int kb = s->key_bytes == 4 ? 30 : 63;
    if (s->key_bytes == 4)
        hipLaunchKernelGGL(k_key_lower_bound<uint32_t>, dim3((nq + 255) / 256), dim3(256), 0, s->stream,
                           (const uint32_t *) s->keys[s->sorted_buf], s->n, queries_dev, nq, kb, out_dev);
    else
        hipLaunchKernelGGL(k_key_lower_bound<uint64_t>, dim3((nq + 255) / 256), dim3(256), 0, s->stream,
                           (const uint64_t *) s->keys[s->sorted_buf], s->n, queries_dev, nq, kb, out_dev);
    return hipGetLastError();
}

hipError_t launch_pack_particles(Sim *s, void *out)
{
    const int c = s->cur;
    if (s->n < 1) return hipSuccess;
    const dim3 grid((s->n + 255) / 256);
    if (s->fp64) {
        State64 &d = s->d;
        hipLaunchKernelGGL(k_pack_particles<double>, grid, dim3(256), 0, s->stream, s->n, d.x[c], d.y[c], d.z[c],
                           d.vx[c], d.vy[c], d.vz[c], d.m[c], d.ax, d.ay, d.az, s->id[c], (double *) out);
    } else {
        hipLaunchKernelGGL(k_pack_particles<float>, grid, dim3(256), 0, s->stream, s->n, s->x[c], s->y[c], s->z[c],
                           s->vx[c], s->vy[c], s->vz[c], s->m[c], s->ax, s->ay, s->az, s->id[c], (float *) out);
    }
    return hipGetLastError();
}

hipError_t launch_unpack_particles(Sim *s, const void *in, int n)
{
    const int c = s->cur;
    if (n < 1) return hipSuccess;
    const dim3 grid((n + 255) / 256);
    if (s->fp64) {
        State64 &d = s->d;
        hipLaunchKernelGGL(k_unpack_particles<double>, grid, dim3(256), 0, s->stream, n, (const double *) in, d.x[c],
                           d.y[c], d.z[c], d.vx[c], d.vy[c], d.vz[c], d.m[c], d.vhx[c], d.vhy[c], d.vhz[c], s->id[c]);
    } else {
        hipLaunchKernelGGL(k_unpack_particles<float>, grid, dim3(256), 0, s->stream, n, (const float *) in, s->x[c],
                           s->y[c], s->z[c], s->vx[c], s->vy[c], s->vz[c], s->m[c], s->vhx[c], s->vhy[c],
                           s->vhz[c], s->id[c]);
    }
    return hipGetLastError();
}

hipError_t launch_bbox(Sim *s, uint32_t *box_dev)
{
    hipLaunchKernelGGL(k_bbox, dim3(std::min((s->n + 255) / 256, 1024)), dim3(256), 0, s->stream,
                       s->n, s->posm, box_dev);
    return hipGetLastError();
}

// records_dev: T * LET_REC floats.  seg: LET_MAX_T + 2 ints of scratch.
hipError_t launch_frontier(Sim *s, float *records_dev, int T)
{
    const int grid = std::min((s->n + 255) / 256, 1024);
    if (s->fp64) {
        hipLaunchKernelGGL(k_frontier_select<Double4>, dim3(1), dim3(256), 0, s->stream, s->counts, s->n, T,
                           s->nodeB, s->d.nodeA, s->o_pb, s->o_pe, records_dev, s->let_seg);
        hipLaunchKernelGGL(k_frontier_bbox<Double4>, dim3(grid), dim3(256), 0, s->stream, s->n, s->d.posm,
                           s->let_seg, records_dev);
    } else {
        hipLaunchKernelGGL(k_frontier_select<Float4>, dim3(1), dim3(256), 0, s->stream, s->counts, s->n, T,
                           s->nodeB, s->nodeA, s->o_pb, s->o_pe, records_dev, s->let_seg);
        hipLaunchKernelGGL(k_frontier_bbox<Float4>, dim3(grid), dim3(256), 0, s->stream, s->n, s->posm,
                           s->let_seg, records_dev);
    }
    return hipGetLastError();
}

// frontiers_dev: ndom * T records (ndom <= LET_MASK_DOMAINS); out: [ndom][cap] records;
// out_count[ndom] is zeroed here.  Domain `skip` (the caller's own, -1 = none) gets nothing.
hipError_t launch_let_export(Sim *s, const float *frontiers_dev, int ndom, int T, int skip,
                             Float4 *out, int cap, uint32_t *out_count)
{
    hipError_t e = hipMemsetAsync(out_count, 0, (size_t) ndom * sizeof(uint32_t), s->stream);
    if (e != hipSuccess) return e;
    // grids from the host-known upper bound of the node count (2n); surplus threads exit
    const int gx = (2 * s->n + 255) / 256;
    const size_t lds = ((size_t) ndom * T * 6 + (size_t) ndom * 6) * sizeof(float);
    hipLaunchKernelGGL(k_let_mask<Float4>, dim3(std::min(gx, 2048)), dim3(256), lds, s->stream, s->counts,
                       s->nodeA, s->nodeB, frontiers_dev, ndom, T, skip, s->p.theta, s->let_mask);
    hipLaunchKernelGGL(k_let_emit, dim3(gx, ndom), dim3(256), 0, s->stream, s->counts, s->nodeA,
                       s->nodeB, s->o_parent, s->posm, s->let_mask, skip, out, cap, out_count);
    return hipGetLastError();
}

// Pruned copies of the tree, one per remote domain (see k_let_tree_emit): out = [ndom] segments
// of cap_bytes; out_count[2 d], [2 d + 1] = node and particle records of domain d's copy.
hipError_t launch_let_export_tree(Sim *s, const float *frontiers_dev, int ndom, int T, int skip,
                                  unsigned char *out, size_t cap_bytes, uint32_t *out_count)
{
    const int nblk = (2 * s->n + 255) / 256;
    const size_t lds = ((size_t) ndom * T * 6 + (size_t) ndom * 6) * sizeof(float);
    if (s->fp64)
        hipLaunchKernelGGL(k_let_mask<Double4>, dim3(std::min(nblk, 2048)), dim3(256), lds, s->stream, s->counts,
                           s->d.nodeA, s->nodeB, frontiers_dev, ndom, T, skip, s->p.theta, s->let_mask);
    else
        hipLaunchKernelGGL(k_let_mask<Float4>, dim3(std::min(nblk, 2048)), dim3(256), lds, s->stream, s->counts,
                           s->nodeA, s->nodeB, frontiers_dev, ndom, T, skip, s->p.theta, s->let_mask);
    hipLaunchKernelGGL(k_let_reach, dim3(std::min(nblk, 2048)), dim3(256), 0, s->stream, s->counts,
                       s->o_parent, s->let_mask, s->let_reach);
    hipLaunchKernelGGL(k_let_tree_count, dim3(nblk, ndom), dim3(256), 0, s->stream, s->counts, s->nodeB,
                       s->let_mask, s->let_reach, skip, nblk, s->let_bsum);
    hipLaunchKernelGGL(k_let_tree_scan, dim3(ndom), dim3(256), 0, s->stream, s->counts, skip, nblk,
                       s->let_bsum, out_count);
    if (s->fp64)
        hipLaunchKernelGGL(k_let_tree_emit<Double4>, dim3(nblk, ndom), dim3(256), 0, s->stream, s->counts, s->d.nodeA,
                           s->nodeB, s->d.posm, s->let_mask, s->let_reach, skip, nblk, s->let_bsum, out_count, out,
                           cap_bytes);
    else
        hipLaunchKernelGGL(k_let_tree_emit<Float4>, dim3(nblk, ndom), dim3(256), 0, s->stream, s->counts, s->nodeA,
                           s->nodeB, s->posm, s->let_mask, s->let_reach, skip, nblk, s->let_bsum, out_count, out,
                           cap_bytes);
    return hipGetLastError();
}

hipError_t launch_let_graft(Sim *s, int ncopies, const int32_t *desc_dev, const int32_t *prefix_dev,
                            int total_items, const unsigned char *recv)
{
    if (total_items < 1) return hipSuccess;
    const int grid = std::min((total_items + 255) / 256, 4096);
    if (s->fp64)
        hipLaunchKernelGGL(k_let_graft<Double4>, dim3(grid), dim3(256), 0, s->stream, ncopies, desc_dev, prefix_dev,
                           recv, s->d.nodeA, s->nodeB, s->d.posm, (float *) nullptr);
    else
        hipLaunchKernelGGL(k_let_graft<Float4>, dim3(grid), dim3(256), 0, s->stream, ncopies, desc_dev, prefix_dev,
                           recv, s->nodeA, s->nodeB, s->posm, s->posm2);
    return hipGetLastError();
}

hipError_t launch_unpack_sources(Sim *s, const Float4 *src, int n)
{
    const int c = s->cur;
    hipLaunchKernelGGL(k_unpack_sources, dim3((n + 255) / 256), dim3(256), 0, s->stream, n, src,
                       s->x[c], s->y[c], s->z[c], s->m[c], s->id[c]);
    return hipGetLastError();
}

}  // namespace nb

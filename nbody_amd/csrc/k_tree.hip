// k_tree.hip -- unique codes, leaf coarsening, Karras radix tree, compressed-octree depth,
// breadth-first node numbering, octree emission and centre-of-mass aggregation.
//
// The OUTPUT contract is the reference's (integer arrays bit-for-bit); the way there is
// not.  Reference stages restated:
//   Points::compute_unique_codes / compute_codes_first_point_idx  src/cuda/points.cu:116-142
//   reduce_leaf_depth, merge_leaves, Btree::generate_leaves       src/cuda/btree.cu:124-187,420-455
//   build_radix_tree, set_node_child, find_split                  src/cuda/btree.cu:37-58,90-122,190-304
//   compute_nodes_depth x ~25 launches                            src/cuda/btree.cu:306-336,479-494
//   sort_to_bfs_order, correct_child_pointers, compute_octree_map src/cuda/btree.cu:339-359,497-572
//   build_octree, compute_octree_nodes_points_range               src/cuda/octree.cu:13-107
//   leaf / per-depth / normalise COM kernels                      src/cuda/octree.cu:109-217,292-346
// No count ever travels to the host during a build: kernels read U, M, n_bt, n_oct from
// device memory and run grid-stride or ticketed tiles under host-known upper bounds.
#include "com_kernel.hpp"
#include "dev_utils.hpp"
#include "nbody_internal.hpp"

namespace nb {

void launch_rank_pass(const uint32_t *keys, uint32_t *keys_out, uint32_t *vals_out,
                      uint32_t *inv_out, const int32_t *n_dev, int n_max, const uint32_t *hist,
                      uint32_t *status, bool late_test, uint32_t *err, hipStream_t st);

constexpr int DEPTH_ROW = 32;      // words of a workgroup's depth counts: a 63-bit key has 22 depths
constexpr int DEPTH_GRID = 2048;   // workgroups of k_depth

template <typename KeyT> struct KeyBits;
template <> struct KeyBits<uint32_t> { static constexpr int KB = 30; };
template <> struct KeyBits<uint64_t> { static constexpr int KB = 63; };

// Length of the common prefix of two keys, counted from the key's top bit (bit KB-1);
// KB for equal keys.  Same value as the reference's `__clzll(a ^ b) - 1` at KB = 63.
__device__ __forceinline__ int lcp_keys(uint32_t a, uint32_t b)
{
    const uint32_t x = a ^ b;
    return x ? __clz((int) x) - 2 : 30;
}
__device__ __forceinline__ int lcp_keys(uint64_t a, uint64_t b)
{
    const uint64_t x = a ^ b;
    return x ? __clzll((long long) x) - 1 : 63;
}

// ---------------------------------------------------------------------------
// Unique codes: head flags -> tile scan with look-back -> compaction.
// unique[u], first_pt[u] (index of the first particle carrying the code), first_pt[U] = n.
// ---------------------------------------------------------------------------
// Striped: thread t of a tile takes the keys t, t + 256, ... so that every load and (where all
// flags are set, the usual case) every store has consecutive lanes on consecutive addresses,
// with nothing staged in LDS.  Order of the output = order of the input: item-major, then
// wave, then lane; the per-(item, wave) flag counts are scanned by one wave.
template <int ITEMS>
__device__ __forceinline__ uint32_t striped_offsets(const unsigned long long (&bal)[ITEMS], uint32_t *s_cnt,
                                                    uint32_t *s_tot)
{
    // s_cnt[ITEMS * 4], filled here with the exclusive prefix of popcount(bal[j]) of wave w at
    // [j * 4 + w]; returns nothing a caller needs besides *s_tot = flags in the tile.
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) s_cnt[j * 4 + w] = (uint32_t) __popcll(bal[j]);
    }
    __syncthreads();
    if (w == 0) {
        static_assert(ITEMS * 4 <= 64 || ITEMS * 4 == 128, "one or two values per lane");
        if (ITEMS * 4 <= 64) {
            const uint32_t v = lane < ITEMS * 4 ? s_cnt[lane] : 0u;
            const uint32_t inc = wave_incl_scan_u32(v);
            if (lane < ITEMS * 4) s_cnt[lane] = inc - v;
            if (lane == 63) *s_tot = inc;
        } else {
            const uint32_t v0 = s_cnt[2 * lane], v1 = s_cnt[2 * lane + 1];
            const uint32_t inc = wave_incl_scan_u32(v0 + v1);
            s_cnt[2 * lane] = inc - v0 - v1;
            s_cnt[2 * lane + 1] = inc - v1;
            if (lane == 63) *s_tot = inc;
        }
    }
    __syncthreads();
    return *s_tot;
}

// HELP as in k_sort_pass: false = tiles numbered by ticket, the look-back simply waits; true =
// tile = blockIdx, and a workgroup whose look-back gives a wait up does the lowest tile that has
// published nothing, then its own again (help_spins / late_test: see there).
__device__ __forceinline__ void late_start(int late_test, int own)
{
    if (late_test && (own & 3) == 1) {
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
        while (__builtin_amdgcn_s_memrealtime() - t0 < 30000ull) __builtin_amdgcn_s_sleep(32);
    }
}

template <typename KeyT, int ITEMS, bool HELP>
__global__ __launch_bounds__(SCAN_THREADS) void k_unique(const KeyT *__restrict__ sorted, int n,
                                                         KeyT *__restrict__ unique,
                                                         int32_t *__restrict__ first_pt,
                                                         Counts *counts, uint32_t *status,
                                                         uint32_t *group_status, uint32_t *ticket,
                                                         uint32_t help_spins, int late_test)
{
    __shared__ uint32_t s_cnt[ITEMS * 4];
    __shared__ uint32_t s_bcast[3];
    __shared__ int s_stuck, s_min;
    __shared__ KeyT s_last[ITEMS * 4 + 1];
    static_assert(SCAN_THREADS == 256, "four waves per tile row");
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    constexpr int TILE = SCAN_THREADS * ITEMS;
    const int ntiles = (n + TILE - 1) / TILE;
    int own = (int) blockIdx.x;
    if constexpr (!HELP) {
        if (tid == 0) s_min = (int) atomicAdd(ticket, 1u);
        __syncthreads();
        own = s_min;
    }
    if (own >= ntiles) return;
    if constexpr (HELP) late_start(late_test, own);
    int tile = own;
    uint32_t rounds = 0;
    while (true) {
        const int tile_base = tile * TILE;
        // Every key is loaded once, all loads in flight together.  A key's predecessor is the
        // neighbouring lane's key; lane 0 of a wave takes it from the last lane of the wave before
        // (same item), wave 0 from the last wave of the item before, through s_last -- only the
        // key before the tile comes from memory.
        KeyT k[ITEMS];
        unsigned long long bal[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) k[j] = sorted[min(tile_base + j * SCAN_THREADS + tid, n - 1)];
        if (tid == 0) s_last[0] = sorted[max(tile_base - 1, 0)];
        if (lane == 63) {
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) s_last[1 + j * 4 + w] = k[j];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int gi = tile_base + j * SCAN_THREADS + tid;
            KeyT prev = (KeyT) __shfl_up((unsigned long long) k[j], 1, 64);
            if (lane == 0) prev = s_last[j * 4 + w];
            bal[j] = __ballot(gi < n && (gi == 0 || k[j] != prev));
        }
        const uint32_t tot = striped_offsets<ITEMS>(bal, s_cnt, &s_bcast[2]);
        if (tid < 64) {
            bool stuck = false;
            const uint32_t te = lb2_exclusive_wave<HELP>(status, group_status, tile, tot, help_spins, &stuck, &counts->err);
            if (tid == 0) { s_bcast[1] = te; s_stuck = stuck ? 1 : 0; }
        }
        __syncthreads();
        if (HELP && s_stuck) {
            const int lowest = lb2_lowest_unpublished(status, 1, 0, tile, &s_min);
            if (++rounds > LB_HELP_ROUNDS) {
                if (tid == 0) atomicOr(&counts->err, (uint32_t) DEV_ERR_SPIN);
                return;
            }
            tile = lowest;
            continue;
        }
        const uint32_t tile_excl = s_bcast[1];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            if ((bal[j] >> lane) & 1ull) {
                const uint32_t pos = tile_excl + s_cnt[j * 4 + w] + (uint32_t) mbcnt(bal[j]);
                unique[pos] = k[j];
                first_pt[pos] = tile_base + j * SCAN_THREADS + tid;
            }
        }
        if (tile == ntiles - 1 && tid == 0) {
            const int U = (int) (tile_excl + tot);
            counts->U = U;
            first_pt[U] = n;
        }
        if (!HELP || tile == own) return;
        tile = own;
        __syncthreads();   // (s_cnt / s_bcast / s_last are rewritten by the next round)
    }
}

// ---------------------------------------------------------------------------
// Leaf coarsening + compaction.  A leaf is the largest octree cell around a code that
// holds at most max_codes unique codes (never larger than the cell in which the code is
// alone with respect to its neighbours); a code is kept iff it is the first code of its
// leaf.  The reference finds a code's leaf by climbing level by level (merge_leaves);
// the same flag follows from ONE cell: let C be the deepest cell that code i shares with
// its predecessor (level lcp(i, i-1) / 3).  Cell populations only grow towards the root,
// so i's leaf lies strictly below C -- and then starts at i -- exactly when C holds more
// than max_codes codes; otherwise the leaf is C or an ancestor and i is not its first
// code.  C's population is only needed up to max_codes + 1: find C's first code with a
// bounded binary search to the left, then test the code max_codes places after it.
// Outputs lfc[0..M) (index of each leaf's first code), lfc[M] = U (the sentinel the
// reference gets from cub::DevicePartition's reversed rear), leaf_key[0..M), M, n_bt.
// ---------------------------------------------------------------------------
// `at(j)` returns unique code j (from LDS for indices near the tile, see k_leaves).
template <typename KeyT, typename At>
__device__ __forceinline__ bool leaf_head_flag(At at, int i, int U, int max_codes)
{
    constexpr int KB = KeyBits<KeyT>::KB;
    if (i == 0) return true;
    const KeyT c = at(i);
    const int level = lcp_keys(c, at(i - 1)) / 3;
    const int sh = KB - 3 * level;
    const KeyT pre = c >> sh;
    // first code of C is i - left, left in [1, max_codes]; beyond that C is too full anyway
    int lo = 1, hi = min(i, max_codes);
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((at(i - mid) >> sh) == pre) lo = mid; else hi = mid - 1;
    }
    const int probe = i - lo + max_codes;  // the (max_codes+1)-th code counted from C's start
    return probe < U && (at(probe) >> sh) == pre;
}

static_assert(SCAN_TILE >= 1024, "the look-back rows are sized for tiles of at least 1024 (max_scan_tiles)");
constexpr int LEAF_HALO = 64;   // codes staged on either side of a tile (covers max_codes <= 64)

template <typename KeyT, int LEAF_ITEMS>
__global__ __launch_bounds__(SCAN_THREADS) void k_leaves(const KeyT *__restrict__ unique,
                                                         int max_codes, int32_t *__restrict__ lfc,
                                                         KeyT *__restrict__ leaf_key,
                                                         Counts *counts, uint32_t *status,
                                                         uint32_t *group_status, uint32_t help_spins,
                                                         int late_test)
{
    constexpr int LEAF_TILE = SCAN_THREADS * LEAF_ITEMS;
    static_assert(LEAF_TILE >= 1024, "the look-back rows are sized for tiles of at least 1024 (max_scan_tiles)");
    __shared__ uint32_t s_cnt[LEAF_ITEMS * 4];
    __shared__ uint32_t s_bcast[3];
    __shared__ int s_stuck, s_min;
    __shared__ KeyT s_keys[LEAF_TILE + 2 * LEAF_HALO];
    const int tid = threadIdx.x;
    const int U = counts->U;
    const int ntiles = (U + LEAF_TILE - 1) / LEAF_TILE;
    const int own = (int) blockIdx.x;
    if (own >= ntiles) return;
    late_start(late_test, own);
    const bool single = U <= max_codes;  // whole set fits one leaf (undefined in the reference)
    int tile = own;
    uint32_t rounds = 0;
    while (true) {
        // The flag of code i looks at codes i - max_codes .. i + max_codes - 1 in a chain of dependent
        // reads: stage the tile and a halo in LDS so that the chain runs at LDS latency.
        const int s_lo = tile * LEAF_TILE - LEAF_HALO;
        for (int j = tid; j < LEAF_TILE + 2 * LEAF_HALO; j += SCAN_THREADS) {
            const int g = s_lo + j;
            if (g >= 0 && g < U) s_keys[j] = unique[g];
        }
        __syncthreads();
        const bool staged = max_codes <= LEAF_HALO;
        auto at = [&](int j) -> KeyT {
            const int r = j - s_lo;
            return (staged && r >= 0 && r < LEAF_TILE + 2 * LEAF_HALO) ? s_keys[r] : unique[j];
        };

        // striped over the tile (thread t takes codes t, t + 256, ...): the lanes of a wave then
        // probe consecutive LDS words at every step of the search
        unsigned long long bal[LEAF_ITEMS];
#pragma unroll
        for (int j = 0; j < LEAF_ITEMS; ++j) {
            const int i = tile * LEAF_TILE + j * SCAN_THREADS + tid;
            bool f = false;
            if (i < U) f = single ? (i == 0) : leaf_head_flag<KeyT>(at, i, U, max_codes);
            bal[j] = __ballot(f);
        }
        const uint32_t tot = striped_offsets<LEAF_ITEMS>(bal, s_cnt, &s_bcast[2]);
        if (tid < 64) {
            bool stuck = false;
            const uint32_t te = lb2_exclusive_wave<true>(status, group_status, tile, tot, help_spins, &stuck, &counts->err);
            if (tid == 0) { s_bcast[1] = te; s_stuck = stuck ? 1 : 0; }
        }
        __syncthreads();
        if (s_stuck) {
            const int lowest = lb2_lowest_unpublished(status, 1, 0, tile, &s_min);
            if (++rounds > LB_HELP_ROUNDS) {
                if (tid == 0) atomicOr(&counts->err, (uint32_t) DEV_ERR_SPIN);
                return;
            }
            tile = lowest;
            continue;
        }
        const uint32_t tile_excl = s_bcast[1];
        const int lane = tid & 63, w = tid >> 6;
#pragma unroll
        for (int j = 0; j < LEAF_ITEMS; ++j) {
            if ((bal[j] >> lane) & 1ull) {
                const int i = tile * LEAF_TILE + j * SCAN_THREADS + tid;
                const uint32_t pos = tile_excl + s_cnt[j * 4 + w] + (uint32_t) mbcnt(bal[j]);
                lfc[pos] = i;
                leaf_key[pos] = at(i);
            }
        }
        if (tile == ntiles - 1 && tid == 0) {
            const int M = (int) (tile_excl + tot);
            counts->M = M;
            counts->n_bt = 2 * M - 1;
            lfc[M] = U;
        }
        if (tile == own) return;
        tile = own;
        __syncthreads();   // (s_keys / s_cnt / s_bcast are rewritten by the next round)
    }
}

// ---------------------------------------------------------------------------
// Karras radix tree over the M leaf keys.  Node ids: leaf j -> 2j, internal i -> 2i+1,
// root = 1.  An internal node writes its own range/level/children and, for each child,
// rt_pd[child] = parent*2 + edge_delta, where edge_delta = 1 iff the child starts a new
// octree level below its radix parent (always 1 for leaves).
// ---------------------------------------------------------------------------
template <typename KeyT>
__global__ __launch_bounds__(256) void k_radix(const KeyT *__restrict__ lk, const Counts *counts,
                                               int32_t *__restrict__ rt_pd,
                                               int32_t *__restrict__ rt_level,
                                               int32_t *__restrict__ rt_lb,
                                               int32_t *__restrict__ rt_le,
                                               int32_t *__restrict__ rt_left,
                                               int32_t *__restrict__ rt_right)
{
    constexpr int KB = KeyBits<KeyT>::KB;
    const int M = counts->M;
    const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
    if (M == 1) {
        if (gtid == 0) { rt_pd[0] = 1; rt_level[0] = 0; }
        return;
    }
    for (int i = gtid; i < M - 1; i += gridDim.x * blockDim.x) {
        const KeyT ki = lk[i];
        auto lcp_at = [&](int j) -> int {
            return (j < 0 || j > M - 1) ? -3 : lcp_keys(ki, lk[j]);
        };
        const int d = lcp_at(i + 1) > lcp_at(i - 1) ? 1 : -1;
        const int lcp_min = lcp_at(i - d);
        int l_max = 2;
        while (lcp_at(i + l_max * d) > lcp_min) l_max <<= 1;
        int length = 0;
        for (int step = l_max >> 1; step >= 1; step >>= 1)
            if (lcp_at(i + (length + step) * d) > lcp_min) length += step;
        const int last = i + length * d;
        const int node_lcp = lcp_keys(ki, lk[last]);
        // split: largest s with lcp(i, i + s*d) > node_lcp
        int s = 0, t = length;
        do {
            t = (t + 1) >> 1;
            if (lcp_at(i + (s + t) * d) > node_lcp) s += t;
        } while (t > 1);
        const int gamma = i + s * d + min(d, 0);
        const int lo = min(i, last), hi = max(i, last);
        const int self = 2 * i + 1;
        const int node_level = node_lcp / 3;

        const bool l_leaf = gamma == lo;
        const int l_id = 2 * gamma + (l_leaf ? 0 : 1);
        const int l_lcp = l_leaf ? KB : lcp_keys(lk[gamma], lk[lo]);
        const int l_delta = min(1, l_lcp / 3 - node_level + (l_leaf ? 1 : 0));
        const bool r_leaf = gamma + 1 == hi;
        const int r_id = 2 * (gamma + 1) + (r_leaf ? 0 : 1);
        const int r_lcp = r_leaf ? KB : lcp_keys(lk[gamma + 1], lk[hi]);
        const int r_delta = min(1, r_lcp / 3 - node_level + (r_leaf ? 1 : 0));

        rt_pd[l_id] = self * 2 + l_delta;
        rt_pd[r_id] = self * 2 + r_delta;
        if (l_leaf) rt_level[l_id] = node_level + 1;
        if (r_leaf) rt_level[r_id] = node_level + 1;
        rt_level[self] = node_level;
        rt_lb[i] = lo;
        rt_le[i] = hi;
        rt_left[i] = l_id;
        rt_right[i] = r_id;
        if (i == 0) rt_pd[1] = 1;  // root: no parent, counts as an octree node
    }
}

// ---------------------------------------------------------------------------
// Compressed-octree depth of every radix node that is an octree node (edge_delta = 1):
// number of edge_delta = 1 edges on its path to the root; one upward walk per node in a
// single launch.  Nodes with edge_delta = 0 get key 255 and sort behind everything.
// Also histograms the keys for the ranking pass.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_depth(const Counts *counts,
                                               const int32_t *__restrict__ rt_pd,
                                               uint32_t *__restrict__ depth_key,
                                               uint32_t *__restrict__ dhist,
                                               uint32_t *__restrict__ rows, uint32_t *ticket)
{
    // The walks are chains of dependent loads: beyond a few million nodes, as many waves as the
    // chip holds (rows != null).  A workgroup's counts then go to its own row (depths 0..DEPTH_ROW-2 and the "not an octree node" key 255 in
    // the last word) and the workgroup that finishes last adds the rows up: per-bin global
    // atomics from a thousand workgroups on two dozen addresses would serialise.
    __shared__ uint32_t h[MAX_DEPTH_BINS];
    __shared__ uint32_t s_last;
    h[threadIdx.x] = 0;
    __syncthreads();
    const int n_bt = counts->n_bt;
    const int root = n_bt == 1 ? 0 : 1;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < n_bt; r += gridDim.x * 256) {
        const int pd = rt_pd[r];
        uint32_t key = 255u;
        if (pd & 1) {
            int dsum = 0;
            int cur = r;
            int guard = 0;
            while (cur != root && guard < 256) {
                const int p = rt_pd[cur];
                dsum += p & 1;
                cur = p >> 1;
                ++guard;
            }
            key = (uint32_t) min(dsum, DEPTH_ROW - 2);
        }
        depth_key[r] = key;
        atomicAdd(&h[key], 1u);
    }
    __syncthreads();
    if (!rows) {   // few workgroups: one global atomic per workgroup and non-empty bin
        const uint32_t c = h[threadIdx.x];
        if (c) atomicAdd(&dhist[threadIdx.x], c);
        return;
    }
    if (threadIdx.x < DEPTH_ROW) {
        const uint32_t c = threadIdx.x == DEPTH_ROW - 1 ? h[255] : h[threadIdx.x];
        __hip_atomic_store(&rows[(size_t) blockIdx.x * DEPTH_ROW + threadIdx.x], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
        s_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    // DEPTH_ROW bins x 8 partial sums per bin
    const int bin = threadIdx.x % DEPTH_ROW, part = threadIdx.x / DEPTH_ROW;
    uint32_t acc = 0;
    constexpr int PARTS = 256 / DEPTH_ROW;
    for (int b0 = part; b0 < (int) gridDim.x; b0 += 8 * PARTS) {   // eight loads in flight
        uint32_t v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int b = b0 + q * PARTS;
            v[q] = b < (int) gridDim.x ? __hip_atomic_load(&rows[(size_t) b * DEPTH_ROW + bin], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += v[q];
    }
    h[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < DEPTH_ROW) {
        uint32_t t = 0;
        for (int q = 0; q < 256 / DEPTH_ROW; ++q) t += h[q * DEPTH_ROW + threadIdx.x];
        dhist[threadIdx.x == DEPTH_ROW - 1 ? 255 : threadIdx.x] = t;
    }
}

// ---------------------------------------------------------------------------
// Octree emission.  Octree id o = breadth-first rank (depth-major, Morton-minor) of radix
// node bfs[o]; children of a node are contiguous.  first_child/num_children come from the
// left-most / right-most chains through edge_delta = 0 nodes.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_emit(
    Counts *counts, float domain, const float *__restrict__ keybox, const uint32_t *__restrict__ dhist,
    const uint32_t *__restrict__ bfs, const uint32_t *__restrict__ octid,
    const int32_t *__restrict__ rt_pd, const int32_t *__restrict__ rt_level,
    const int32_t *__restrict__ rt_lb, const int32_t *__restrict__ rt_le,
    const int32_t *__restrict__ rt_left, const int32_t *__restrict__ rt_right,
    const int32_t *__restrict__ lfc, const int32_t *__restrict__ first_pt,
    int32_t *__restrict__ o_first_child, int32_t *__restrict__ o_num_children,
    int32_t *__restrict__ o_lb, int32_t *__restrict__ o_le, int32_t *__restrict__ o_level,
    float *__restrict__ o_size, int32_t *__restrict__ o_pb, int32_t *__restrict__ o_pe,
    int32_t *__restrict__ o_parent, NodeB *__restrict__ nodeB, int com_local,
    uint4 *__restrict__ local_roots, uint32_t *n_local_roots)
{
    const int n_bt = counts->n_bt;
    const int n_oct = n_bt - (int) dhist[255];
    const int root = n_bt == 1 ? 0 : 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) counts->n_oct = n_oct;
    if (keybox) domain = keybox[3];   // cells subdivide the cube the keys were taken relative to
    for (int o = blockIdx.x * 256 + threadIdx.x; o < n_oct; o += gridDim.x * 256) {
        const int r = (int) bfs[o];
        const bool leaf = (r & 1) == 0;
        const int idx = r >> 1;
        const int level = rt_level[r];
        int lb, le, fc = -1, nc = 0;
        if (leaf) {
            lb = le = idx;
        } else {
            lb = rt_lb[idx];
            le = rt_le[idx];
            int f = rt_left[idx];
            while (!(rt_pd[f] & 1)) f = rt_left[f >> 1];
            int l = rt_right[idx];
            while (!(rt_pd[l] & 1)) l = rt_right[l >> 1];
            fc = (int) octid[f];
            nc = (int) octid[l] - fc + 1;
        }
        int parent = -1;
        if (r != root) {
            int p = rt_pd[r] >> 1;
            while (!(rt_pd[p] & 1)) p = rt_pd[p] >> 1;
            parent = (int) octid[p];
        }
        const int pb = first_pt[lfc[lb]];
        const int pe = first_pt[lfc[le + 1]] - 1;
        const float size = domain / (float) (1 << level);
        o_first_child[o] = fc;
        o_num_children[o] = nc;
        o_lb[o] = lb;
        o_le[o] = le;
        o_level[o] = level;
        o_size[o] = size;
        o_pb[o] = pb;
        o_pe[o] = pe;
        o_parent[o] = parent;
        uint32_t meta = NODE_LEAF;
        if (!leaf) {
            meta = 0;
            for (int c = 0; c < nc; ++c)
                if ((bfs[fc + c] & 1u) == 0u) meta |= 1u << c;
        }
        nodeB[o] = leaf ? NodeB{size, pb, pe - pb + 1, meta} : NodeB{size, fc, nc, meta};
        // the centre-of-mass pass gives one wave to every inner node with at most com_local
        // particles whose parent has more
        if (!leaf && pe - pb + 1 <= com_local) {
            bool big_parent = parent < 0;
            if (!big_parent) {
                int p = rt_pd[r] >> 1;
                while (!(rt_pd[p] & 1)) p = rt_pd[p] >> 1;
                const int plb = rt_lb[p >> 1], ple = rt_le[p >> 1];
                big_parent = first_pt[lfc[ple + 1]] - first_pt[lfc[plb]] > com_local;
            }
            // (node, first particle, particles, parent): all the wave needs to get going
            if (big_parent)
                local_roots[atomicAdd(n_local_roots, 1u)] = uint4{(uint32_t) o, (uint32_t) pb, (uint32_t) (pe - pb + 1), (uint32_t) parent};
        }
    }
}

// ---------------------------------------------------------------------------
template <typename KeyT>
static hipError_t build_impl(Sim *s)
{
    const int n = s->n;
    hipStream_t st = s->stream;
    // tickets, look-back words, depth histogram start at zero: the sort that precedes a build
    // has cleared them together with its own scratch
    hipError_t e = s->build_scratch_zeroed ? hipSuccess : hipMemsetAsync(s->zero_block, 0, s->zero_bytes, st);
    if (e != hipSuccess) return e;
    s->build_scratch_zeroed = false;
    const KeyT *sorted = (const KeyT *) s->keys[s->sorted_buf];
    // (test aid: late workgroups and an early give-up exercise the helping path of the look-back)
    const int late = (s->p.flags & NBODY_FLAG_DEBUG_LATE_TILES) ? 1 : 0;
    const uint32_t help = late ? LB_TEST_SPINS : LB_HELP_SPINS;
    // Tiles of 8192 keys from a million keys on (fewer tiles to number and to look back over),
    // of 4096 below so that there are enough of them; numbered by ticket up to 4 M keys, by
    // blockIdx with helping beyond (and in the test mode).
    {
        uint32_t *st0 = s->scan_status, *gst0 = s->scan_status + s->max_scan_tiles, *tk = s->tickets + 0;
#define NB_UNIQUE(IT, HELP)                                                                             \
        hipLaunchKernelGGL((k_unique<KeyT, IT, HELP>), dim3((n + SCAN_THREADS * IT - 1) / (SCAN_THREADS * IT)), \
                           dim3(SCAN_THREADS), 0, st, sorted, n, (KeyT *) s->unique, s->first_pt, s->counts, \
                           st0, gst0, tk, help, late)
#ifndef NB_UNIQ_BIG_ITEMS
#define NB_UNIQ_BIG_ITEMS (2 * SCAN_ITEMS)
#endif
#ifndef NB_UNIQ_HELP_FROM
#define NB_UNIQ_HELP_FROM ((1 << 22) + 1)
#endif
        const bool big = n >= (1 << 20), by_help = late || n >= NB_UNIQ_HELP_FROM;
        if (big) { if (by_help) NB_UNIQUE(NB_UNIQ_BIG_ITEMS, true); else NB_UNIQUE(NB_UNIQ_BIG_ITEMS, false); }
        else     { if (by_help) NB_UNIQUE(SCAN_ITEMS, true); else NB_UNIQUE(SCAN_ITEMS, false); }
#undef NB_UNIQUE
    }
    // tiles of 2048 codes, of 1024 below a million particles (524 k: 17.4 -> 14.5 us; at 2 M the
    // smaller tile costs 14 us more)
#define NB_LEAVES(IT)                                                                                  \
    hipLaunchKernelGGL((k_leaves<KeyT, IT>), dim3((n + SCAN_THREADS * IT - 1) / (SCAN_THREADS * IT)),    \
                       dim3(SCAN_THREADS), 0, st, (const KeyT *) s->unique, s->p.max_num_codes_per_leaf, \
                       s->lfc, (KeyT *) s->leaf_key, s->counts,                                          \
                       s->scan_status + 2 * (size_t) s->max_scan_tiles,                                  \
                       s->scan_status + 3 * (size_t) s->max_scan_tiles, help, late)
    if (n < (1 << 20)) NB_LEAVES(4); else NB_LEAVES(8);
#undef NB_LEAVES
    const int node_grid = std::min((n + 255) / 256, 2048);
    hipLaunchKernelGGL(k_radix<KeyT>, dim3(node_grid), dim3(256), 0, st,
                       (const KeyT *) s->leaf_key, s->counts, s->rt_pd, s->rt_level, s->rt_lb,
                       s->rt_le, s->rt_left, s->rt_right);
    // (every workgroup ends with one same-address ticket atomic, ~15 ns each one after the
    // other: a large grid pays only where the walks dominate)
    const bool many = n > (1 << 22);
    const int depth_grid = std::min((2 * n + 255) / 256, many ? DEPTH_GRID : 512);
    hipLaunchKernelGGL(k_depth, dim3(depth_grid), dim3(256), 0, st, s->counts, s->rt_pd,
                       s->depth_key, s->dhist, many ? s->depth_rows : (uint32_t *) nullptr, s->tickets + 9);
    launch_rank_pass(s->depth_key, s->o_depth, s->bfs, s->octid, &s->counts->n_bt, 2 * n,
                     s->dhist, s->rank_status, late != 0, &s->counts->err, st);
    hipLaunchKernelGGL(k_emit, dim3(node_grid), dim3(256), 0, st, s->counts, s->p.domain_size,
                       (const float *) s->keybox, s->dhist, s->bfs, s->octid, s->rt_pd, s->rt_level, s->rt_lb, s->rt_le,
                       s->rt_left, s->rt_right, s->lfc, s->first_pt, s->o_first_child,
                       s->o_num_children, s->o_lb, s->o_le, s->o_level, s->o_size, s->o_pb,
                       s->o_pe, s->o_parent, s->nodeB, s->fp64 ? com_local<double>() : com_local<float>(),
                       reinterpret_cast<uint4 *>(s->depth_key), s->tickets + COM_ROOTS_TICKET);
    return hipGetLastError();
}

hipError_t launch_build(Sim *s)
{
    return s->key_bytes == 4 ? build_impl<uint32_t>(s) : build_impl<uint64_t>(s);
}

hipError_t launch_com(Sim *s)
{
    const int grid = std::min((2 * s->n + 255) / 256, COM_GRID);
    hipLaunchKernelGGL((k_com_t<float, Float4>), dim3(grid), dim3(256), 0, s->stream, s->counts, s->o_num_children,
                       s->o_first_child, s->o_pb, s->o_pe, s->o_parent, s->posm, s->o_wsum,
                       s->nodeA, s->com_counter, reinterpret_cast<const uint4 *>(s->depth_key),
                       s->tickets + COM_ROOTS_TICKET, &s->counts->err);
    return hipGetLastError();
}

}  // namespace nb

// k_sort.hip -- LDS-staged onesweep LSD radix sort of (Morton key, particle index) pairs.
//
// Replaces the reference's cub::DeviceMergeSort::SortPairs + thrust::sequence
// (src/cuda/points.cu:73-84).  Ascending by key, STABLE (ties keep index order), which is
// the tie order this build pins; the reference's merge sort leaves it unspecified.
//
// Structure (8-bit digits, P = ceil(key_bits/8) passes):
//   k_sort_hist : one read of the keys, all P digit histograms at once (LDS atomics, then
//                 one global atomic per non-empty bin per workgroup).
//   k_sort_pass : one launch per digit.  A workgroup takes a tile (its own index, or a ticket
//                 in launches of few tiles: see the kernel), ranks
//                 its keys with wave64 ballot matching, publishes the tile's 256 digit
//                 counts, reorders the tile in LDS by digit, resolves the counts' global
//                 offsets by the two-level look-back over the earlier tiles (dev_utils.hpp)
//                 and writes each digit run to its final place with consecutive lanes on
//                 consecutive addresses.  Keys and indices are read once and written
//                 once per pass: 2*(sizeof(KeyT)+4) bytes per element per pass.
//   63-bit keys: the passes sort on the upper 40 bits only (5 passes instead of 8); elements
//                 whose upper 40 bits agree -- a cell of domain / 2^13 -- are then ordered on
//                 the remaining 24 bits where they lie, by k_tiefix: one thread per run of equal
//                 upper bits, ranked in registers (runs are 2-3 long in practice); a run longer
//                 than TIE_SHORT is taken by the whole wave of its head (stable 8-bit LSD passes
//                 within the run).  The result is the stable sort on the whole key.
#include "dev_utils.hpp"
#include "nbody_internal.hpp"

#ifndef NB_HIST_GRID
#define NB_HIST_GRID 256
#endif
#ifndef NB_HIST_THREADS
#define NB_HIST_THREADS 1024
#endif
#ifndef NB_HIST_KEYS
#define NB_HIST_KEYS 8
#endif
namespace nb {

// digits P0 .. NPASS-1 (ghist keeps one row of 256 per digit, rows below P0 stay zero)
template <typename KeyT, int NPASS, int P0>
__global__ __launch_bounds__(NB_HIST_THREADS) void k_sort_hist(const KeyT *__restrict__ keys, int n_host,
                                                               uint32_t *__restrict__ ghist,
                                                               const int32_t *n_dev)
{
    constexpr int HT = NB_HIST_THREADS;
    const int n = n_dev ? *n_dev : n_host;   // (device-side count: the moved elements of a coherent sort)
    __shared__ uint32_t h[NPASS * 256];
    for (int j = threadIdx.x + P0 * 256; j < NPASS * 256; j += HT) h[j] = 0;
    __syncthreads();
    // HK consecutive keys per thread and round, in few but large workgroups: the hot bins take one
    // global atomic per workgroup, so their number stays at one per CU, and 16 waves per CU keep
    // enough loads in flight (256-thread workgroups of 16 keys per thread: 17.5 us at 2 M keys,
    // 84 us at 16.7 M; 1024 threads of 8 keys: 12 and 40).  The keys arrive in the
    // previous step's order, so their upper digits repeat from one key to the next: a wave
    // whose HK x 64 keys agree in a digit adds them with one LDS atomic, and a thread's equal
    // neighbours share one otherwise.
    constexpr int HK = NB_HIST_KEYS;
    const int stride = gridDim.x * HT * HK;
    const int lane = threadIdx.x & 63;
    for (int i0 = (blockIdx.x * HT + threadIdx.x) * HK; i0 < n; i0 += stride) {
        KeyT k[HK];
        const int cntv = min(HK, n - i0);
        if (cntv == HK) {
#pragma unroll
            for (int j = 0; j < HK; ++j) k[j] = keys[i0 + j];
        } else {
#pragma unroll
            for (int j = 0; j < HK; ++j) k[j] = keys[min(i0 + j, n - 1)];
        }
        const bool full = __all(cntv == HK);   // (lanes past n have left the loop)
#pragma unroll
        for (int p = P0; p < NPASS; ++p) {
            int d = (int) ((k[0] >> (8 * p)) & 255);
            const int d0 = __builtin_amdgcn_readfirstlane(d);
            bool same = d == d0;
#pragma unroll
            for (int j = 1; j < HK; ++j) same = same && (int) ((k[j] >> (8 * p)) & 255) == d0;
            if (full && __all(same)) {
                const unsigned long long live = __ballot(true);
                if (lane == (int) __builtin_ctzll(live)) atomicAdd(&h[p * 256 + d0], (uint32_t) HK * (uint32_t) __popcll(live));
                continue;
            }
            int run = 1;
#pragma unroll
            for (int j = 1; j < HK; ++j) {
                const int dj = (int) ((k[j] >> (8 * p)) & 255);
                if (j < cntv) {
                    if (dj == d) ++run;
                    else { atomicAdd(&h[p * 256 + d], (uint32_t) run); d = dj; run = 1; }
                }
            }
            atomicAdd(&h[p * 256 + d], (uint32_t) run);
        }
    }
    __syncthreads();
    for (int j = threadIdx.x + P0 * 256; j < NPASS * 256; j += HT) {
        const uint32_t c = h[j];
        if (c) atomicAdd(&ghist[j], c);
    }
}

// ---------------------------------------------------------------------------
// Ordering of the elements whose upper key bits (above `low_bits`) agree, after the digit
// passes have sorted on those upper bits.  Within such a run the elements are in index order
// (the passes are stable and start from the identity), so a stable sort of the run on the
// whole key yields the stable sort of everything.  A run only ever holds keys with the same
// upper bits, whatever its sort is doing, so neighbouring runs can be found and sorted at the
// same time.
// ---------------------------------------------------------------------------
constexpr int TIE_SHORT = 16;   // longest run one thread orders (in registers); longer ones take a wave

constexpr int TIE_REG = TIE_SHORT;

template <typename KeyT>
__device__ __forceinline__ bool tiefix_at(KeyT *keys, uint32_t *vals, int n, int low_bits, int i)
{
    const bool in = i < n;
    const int ci = in ? i : n - 1;
    // a run starts where a key agrees (in its upper bits) with its successor and not with its
    // predecessor; nearly always no key of the wave does and nothing is left to do
    const KeyT k0 = keys[ci];
    const KeyT kn = keys[min(ci + 1, n - 1)], kp = keys[max(ci - 1, 0)];
    const KeyT h = k0 >> low_bits;
    const bool head = in && i + 1 < n && (kn >> low_bits) == h && !(i > 0 && (kp >> low_bits) == h);
    if (!head) return false;
    // only the heads go on (a few lanes of a wave).  Runs of two or three first: they are nearly
    // all there is.
    const KeyT k2 = keys[min(i + 2, n - 1)], k3 = keys[min(i + 3, n - 1)];
    const bool t2 = i + 2 < n && (k2 >> low_bits) == h, t3 = t2 && i + 3 < n && (k3 >> low_bits) == h;
    if (!t2) {   // two elements
        if (kn < k0) {
            const uint32_t v0 = vals[i], v1 = vals[i + 1];
            keys[i] = kn; keys[i + 1] = k0;
            vals[i] = v1; vals[i + 1] = v0;
        }
        return false;
    }
    if (!t3) {   // three elements: stable ranks
        const uint32_t v0 = vals[i], v1 = vals[i + 1], v2 = vals[i + 2];
        const int r0 = (kn < k0) + (k2 < k0), r1 = (k0 <= kn) + (k2 < kn), r2 = (k0 <= k2) + (kn <= k2);
        if (r0 != 0 || r1 != 1) {
            keys[i + r0] = k0; vals[i + r0] = v0;
            keys[i + r1] = kn; vals[i + r1] = v1;
            keys[i + r2] = k2; vals[i + r2] = v2;
        }
        return false;
    }
    // longer: the TIE_REG - 1 keys after the head's own, fetched together
    KeyT k[TIE_REG];
    k[0] = k0;
    k[1] = kn;
#pragma unroll
    for (int a = 2; a < TIE_REG; ++a) k[a] = keys[min(i + a, n - 1)];
    int L = 2;
#pragma unroll
    for (int a = 2; a < TIE_REG; ++a)
        if (L == a && i + a < n && (k[a] >> low_bits) == h) L = a + 1;
    if (L == TIE_REG && i + L < n && (keys[i + L] >> low_bits) == h) return true;   // longer than a thread takes
    // rank every element of the run among the others (equal keys by position) and move those
    // that are out of place; all reads come before the writes
    uint32_t v[TIE_REG];
#pragma unroll
    for (int a = 0; a < TIE_REG; ++a) v[a] = a < L ? vals[i + a] : 0u;
#pragma unroll
    for (int a = 0; a < TIE_REG; ++a) {
        if (a < L) {
            int r = 0;
#pragma unroll
            for (int b = 0; b < TIE_REG; ++b)
                if (b < L) r += (k[b] < k[a] || (k[b] == k[a] && b < a)) ? 1 : 0;
            if (r != a) { keys[i + r] = k[a]; vals[i + r] = v[a]; }
        }
    }
    return false;
}

// One wave orders one long run: low_bits / 8 stable 8-bit LSD passes over the run, between the
// run's place in the sorted buffers (k0, v0) and the same place in the other buffer pair (k1, v1),
// whose contents are dead after the digit passes.  64 elements at a time, in order: lanes with
// the same digit rank themselves by ballot matching, the digit's running offset lives in LDS
// (base: 256 words of the wave's own).  All 64 lanes call it with the same arguments.
template <typename KeyT>
__device__ __forceinline__ void tiefix_long_run(KeyT *k0, uint32_t *v0, KeyT *k1, uint32_t *v1, int n,
                                                int low_bits, int start, uint32_t *base)
{
    const int lane = threadIdx.x & 63;
    const KeyT h = k0[start] >> low_bits;
    int L = 0;   // length of the run: first position whose upper bits differ
    while (true) {
        const int j = start + L + lane;
        const unsigned long long same = __ballot(j < n && (k0[j] >> low_bits) == h);
        if (same != ~0ull) { L += __ffsll((long long) ~same) - 1; break; }
        L += 64;
    }
    KeyT *ks = k0 + start, *kd = k1 + start;
    uint32_t *vs = v0 + start, *vd = v1 + start;
    const int passes = low_bits / 8;
    for (int p = 0; p < passes; ++p) {
        const int shift = 8 * p;
#pragma unroll
        for (int q = 0; q < 4; ++q) base[lane * 4 + q] = 0;
        __builtin_amdgcn_wave_barrier();
        for (int j = lane; j < L; j += 64) atomicAdd(&base[(uint32_t) (ks[j] >> shift) & 255u], 1u);
        __builtin_amdgcn_wave_barrier();
        // exclusive scan of the 256 counts: four bins per lane
        uint32_t c[4], sum = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { c[q] = base[lane * 4 + q]; sum += c[q]; }
        uint32_t off = wave_incl_scan_u32(sum) - sum;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 4; ++q) { base[lane * 4 + q] = off; off += c[q]; }
        __builtin_amdgcn_wave_barrier();
        for (int j0 = 0; j0 < L; j0 += 64) {
            const int j = j0 + lane;
            const bool valid = j < L;
            const KeyT kk = valid ? ks[j] : (KeyT) 0;
            const uint32_t vv = valid ? vs[j] : 0u;
            const uint32_t d = (uint32_t) (kk >> shift) & 255u;
            unsigned long long m = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const bool bit = (d >> b) & 1u;
                const unsigned long long bb = __ballot(bit);
                m &= bit ? bb : ~bb;
            }
            const uint32_t rnk = (uint32_t) mbcnt(m);
            const uint32_t at = base[d] + rnk;
            __builtin_amdgcn_wave_barrier();
            if (valid) {
                kd[at] = kk;
                vd[at] = vv;
                if (rnk == 0) base[d] = at + (uint32_t) __popcll(m);
            }
            __builtin_amdgcn_wave_barrier();
        }
        // what this pass stored is read by other lanes of the wave in the next one
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        KeyT *tk = ks; ks = kd; kd = tk;
        uint32_t *tv = vs; vs = vd; vd = tv;
    }
    if (passes & 1) {   // the run ended up in the other buffer pair
        for (int j = lane; j < L; j += 64) { kd[j] = ks[j]; vd[j] = vs[j]; }
    }
}

// Every thread looks at one element; a run's head orders the run if it is short; the long ones
// (dense cores, coincident particles: rare) are then taken one after the other by the whole wave
// of the head that found them.  While a wave reorders its long run the other threads that look at
// elements of that run see some key of the run -- always one with the run's upper bits, which is
// all they compare -- so nothing else depends on the order inside it.
// (grid-stride: the launch may have fewer workgroups than n / 256, e.g. when n is a device-side count)
template <typename KeyT>
__global__ __launch_bounds__(256) void k_tiefix(KeyT *keys, uint32_t *vals, KeyT *keys_other, uint32_t *vals_other,
                                                int n_host, int low_bits, const int32_t *n_dev)
{
    __shared__ uint32_t s_base[4][256];
    const int n = n_dev ? *n_dev : n_host;
    const int w = threadIdx.x >> 6;
    for (int b = (int) blockIdx.x; b * 256 < n; b += (int) gridDim.x) {
        const int i = b * 256 + (int) threadIdx.x;
        const bool is_long = tiefix_at<KeyT>(keys, vals, n, low_bits, i);
        unsigned long long lm = __ballot(is_long);
        while (lm) {
            const int f = (int) __builtin_ctzll(lm);
            lm &= lm - 1;
            tiefix_long_run<KeyT>(keys, vals, keys_other, vals_other, n, low_bits, b * 256 + (w << 6) + f, s_base[w]);
        }
    }
}

// One digit pass.  n is read from n_dev when it is non-null (device-side element count,
// used by the octree node ranking), else n_host.  IOTA: the value of element i is i.
// inv_out (nullable): inv_out[value] = final position.
// Two ways of numbering the tiles (dev_utils.hpp, "Forward progress"):
//   HELP = false: a tile draws its index from *ticket (launches of few tiles, where the queue
//     for the counter is short); its predecessors are running, the look-back simply waits.
//   HELP = true: tile = blockIdx; a workgroup whose look-back waits longer than help_spins polls
//     does the lowest tile that has published nothing before it starts its own again.
//     late_test (test aid): every fourth workgroup idles ~0.3 ms before it starts, so that its
//     tile is taken care of by a helper first and then done a second time.
template <typename KeyT, int ITEMS, bool IOTA, bool HELP>
__global__ __launch_bounds__(SORT_THREADS) void k_sort_pass(
    const KeyT *__restrict__ kin, const uint32_t *__restrict__ vin, KeyT *__restrict__ kout,
    uint32_t *__restrict__ vout, uint32_t *__restrict__ inv_out, const int32_t *n_dev,
    int n_host, int shift, const uint32_t *__restrict__ ghist, uint32_t *status,
    uint32_t *group_status, uint32_t *ticket, uint32_t help_spins, int late_test, uint32_t *err)
{
    // NW waves share a tile: its 256 digits are looked after by the first 256 threads
    constexpr int NW = SORT_WAVES, NT = SORT_THREADS;
    constexpr int TILE = NT * ITEMS;
    static_assert(NT >= 256, "one thread per digit");
    __shared__ uint32_t s_whist[NW][256];
    __shared__ uint32_t s_dstart[256];
    __shared__ uint32_t s_goff[256];
    __shared__ uint32_t s_scan[NW + 1];
    __shared__ int s_stuck, s_min;
    __shared__ KeyT s_keys[TILE];
    __shared__ uint32_t s_vals[TILE];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int n = n_dev ? *n_dev : n_host;
    const int ntiles = (n + TILE - 1) / TILE;
    int own = (int) blockIdx.x;
    if constexpr (!HELP) {
        if (tid == 0) s_min = (int) atomicAdd(ticket, 1u);
        __syncthreads();
        own = s_min;
    }
    if (own >= ntiles) return;
    if constexpr (HELP) {
        if (late_test && (own & 3) == 1) {
            const uint64_t t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
            while (__builtin_amdgcn_s_memrealtime() - t0 < 30000ull) __builtin_amdgcn_s_sleep(32);
        }
    }
    int tile = own;
    uint32_t rounds = 0;
    while (true) {
        for (int j = tid; j < NW * 256; j += NT) (&s_whist[0][0])[j] = 0;
        if (tid == 0) s_stuck = 0;
        __syncthreads();
        const int tile_base = tile * TILE;
        const int tile_count = min(TILE, n - tile_base);

        KeyT key[ITEMS];
        uint32_t val[ITEMS];
        uint32_t rank[ITEMS];
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const int li = w * 64 * ITEMS + k * 64 + lane;
            const bool valid = li < tile_count;
            key[k] = valid ? kin[tile_base + li] : ~(KeyT) 0;
            val[k] = IOTA ? (uint32_t) (tile_base + li) : (valid ? vin[tile_base + li] : 0u);
        }
        // Rank every key among the keys of its wave with the same digit (stable: earlier
        // rounds first, lower lanes first).
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const int li = w * 64 * ITEMS + k * 64 + lane;
            const bool valid = li < tile_count;
            const uint32_t d = (uint32_t) (key[k] >> shift) & 255u;
            unsigned long long m = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const bool bit = (d >> b) & 1u;
                const unsigned long long bb = __ballot(bit);
                m &= bit ? bb : ~bb;
            }
            const uint32_t r = (uint32_t) mbcnt(m);
            const uint32_t c = (uint32_t) __popcll(m);
            const uint32_t prev = s_whist[w][d];
            __builtin_amdgcn_wave_barrier();
            if (valid && r == 0) s_whist[w][d] = prev + c;
            __builtin_amdgcn_wave_barrier();
            rank[k] = prev + r;
        }
        __syncthreads();
        // Thread d owns digit d: per-wave counts -> exclusive prefix over the waves.
        const bool digit = tid < 256;
        uint32_t cnt = 0;
        if (digit) {
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) {
                const uint32_t c = s_whist[ww][tid];
                s_whist[ww][tid] = cnt;
                cnt += c;
            }
            // the tile's digit counts start their way to the other tiles now; the offsets they
            // resolve to are needed only by the write-out, after the reorder in LDS
            lb2_publish(status, 256, tid, tile, cnt);
        }
        uint32_t tot;
        const uint32_t dstart = block_excl_scan_u32<NW>(cnt, s_scan, &tot);
        const uint32_t gbase = block_excl_scan_u32<NW>(digit ? ghist[tid] : 0u, s_scan, &tot);
        if (digit) s_dstart[tid] = dstart;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const int li = w * 64 * ITEMS + k * 64 + lane;
            if (li < tile_count) {
                const uint32_t d = (uint32_t) (key[k] >> shift) & 255u;
                const uint32_t lp = s_dstart[d] + s_whist[w][d] + rank[k];
                s_keys[lp] = key[k];
                s_vals[lp] = val[k];
            }
        }
        bool stuck = false;
        if (digit) {
            const uint32_t excl = lb2_resolve<HELP>(status, group_status, 256, tid, tile, cnt, help_spins, &stuck, err);
            if (HELP && stuck) s_stuck = 1;
            s_goff[tid] = gbase + excl - dstart;  // final position = s_goff[d] + position in tile
        }
        __syncthreads();
        if (HELP && s_stuck) {
            // some predecessor's words did not come: if a tile below this one has published
            // nothing at all, do that one (its owner has not started), else just wait again
            const int lowest = lb2_lowest_unpublished(status, 256, 0, tile, &s_min);
            if (++rounds > LB_HELP_ROUNDS) {
                if (tid == 0) atomicOr(err, (uint32_t) DEV_ERR_SPIN);
                return;
            }
            tile = lowest;   // (== tile when every predecessor has published: same tile again)
            continue;
        }
        for (int j = tid; j < tile_count; j += NT) {
            const KeyT kk = s_keys[j];
            const uint32_t d = (uint32_t) (kk >> shift) & 255u;
            const uint32_t gpos = s_goff[d] + (uint32_t) j;
            const uint32_t v = s_vals[j];
            kout[gpos] = kk;
            vout[gpos] = v;
            if (inv_out) inv_out[v] = gpos;
        }
        if (!HELP) return;
        // a helped tile is done: now this workgroup's own, from the start; its own is done: on to
        // the next one it owns, when the launch has fewer workgroups than tiles (a workgroup takes
        // its tiles in ascending order, so whoever a tile waits for is at work on it or on a lower
        // one, which needs nothing from above)
        if (tile == own) {
            own += (int) gridDim.x;
            if (own >= ntiles) return;
            rounds = 0;
        }
        tile = own;
        __syncthreads();   // (the write-out has read s_keys / s_goff)
    }
}

// Explicit instantiation used by the octree node ranking pass (k_tree.hip).
void launch_rank_pass(const uint32_t *keys, uint32_t *keys_out, uint32_t *vals_out,
                      uint32_t *inv_out, const int32_t *n_dev, int n_max, const uint32_t *hist,
                      uint32_t *status, bool late_test, uint32_t *err, hipStream_t st)
{
    // status: one row of 256 words per tile, then one per group of tiles
    const int grid = (n_max + SORT_TILE - 1) / SORT_TILE;
    // (numbered by blockIdx: the grid is sized for the largest count, most of it leaves at once)
    hipLaunchKernelGGL((k_sort_pass<uint32_t, SORT_ITEMS, true, true>), dim3(grid), dim3(SORT_THREADS), 0, st,
                       keys, (const uint32_t *) nullptr, keys_out, vals_out, inv_out, n_dev, 0, 0,
                       hist, status, status + (size_t) grid * 256, (uint32_t *) nullptr,
                       late_test ? LB_TEST_SPINS : LB_HELP_SPINS, late_test ? 1 : 0, err);
}


// ---------------------------------------------------------------------------
// Sort of keys that are NEARLY in order already.  A step starts from the particle order the
// previous step's sort left behind, and with the reference's time steps (1e-6 for the sphere,
// src/renderer.cpp:314-318) a particle's new key rarely leaves the gap between its neighbours'
// old keys.  With O the previous step's sorted keys (O[i] = old key of the particle now at index
// i) and m_i = O[i] + (O[i+1] - O[i]) / 2, element i STAYS iff its new key lies in
// (m_{i-1}, m_i]: these intervals are disjoint and ascending, so the staying elements are in
// strictly ascending key order whatever the others do.  The others (MOVED: a key that crossed
// a cell boundary coarser than the distance to the neighbour) are compacted in index order,
// sorted by the ordinary digit passes with a device-side count, and merged back:
//   staying i : position = (staying elements before i) + (moved elements below (key_i, i))
//   moved j-th: position = j + (staying elements below its (key, index))
// which is the stable sort of everything, the same permutation the digit passes over all n
// elements produce.  Nothing here depends on how many elements moved; only the time does, and
// the host picks this path from the count a previous step reported (sort_impl).
// ---------------------------------------------------------------------------
constexpr int COH_ITEMS = 8;
constexpr int COH_TILE = 256 * COH_ITEMS;        // 2048 elements = 32 ballot words per tile
constexpr int COH_WORDS = COH_TILE / 64;
constexpr int COH_MERGE_EXTRA = 1024;             // workgroups of k_coh_merge that place the moved elements
constexpr int COH_SORT_GRID = 256;                // workgroups of a digit pass over the moved elements (they loop)
constexpr int COH_MAX_SHARE = 64;                 // the path pays up to about n / 52 moved (2 M keys): taken up to n / 64
constexpr int COH_TINY_SHARE = 1024;              // "hardly any": at most n / 1024 moved
constexpr int COH_TINY_STREAK = 3;                // that many such counts in a row, and ...
constexpr int COH_TINY_AGE = 256;                 // ... a count up to this many sorts old is still believed

template <typename KeyT>
__global__ __launch_bounds__(256) void k_coh_flags(const KeyT *__restrict__ knew, const KeyT *__restrict__ kold,
                                                   int n, unsigned long long *__restrict__ bits,
                                                   uint32_t *__restrict__ tile_cnt)
{
    __shared__ uint32_t s_red[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int tile = blockIdx.x;
    const int base = tile * COH_TILE;
    uint32_t cnt = 0;
#pragma unroll
    for (int j = 0; j < COH_ITEMS; ++j) {
        const int i = base + j * 256 + tid;
        bool stay = false;
        if (i < n) {
            const KeyT k = knew[i], oc = kold[i];
            const KeyT op = kold[max(i - 1, 0)], on = kold[min(i + 1, n - 1)];
            const bool above = i == 0 || k > op + ((oc - op) >> 1);
            const bool below = i == n - 1 || k <= oc + ((on - oc) >> 1);
            stay = above && below;
        }
        const unsigned long long b = __ballot(stay);
        if (lane == 0 && base + j * 256 + w * 64 < n) bits[(base >> 6) + j * 4 + w] = b;
        cnt += (uint32_t) __popcll(b);
    }
    if (lane == 0) s_red[w] = cnt;   // (cnt is wave-uniform)
    __syncthreads();
    if (tid == 0) tile_cnt[tile] = s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

// Exclusive prefix of the tiles' staying counts, the moved count (one workgroup).  ctl[0] = moved
// count, ctl[2] = how many counts in a row were at most n / COH_TINY_SHARE, ctl[3] = the n they
// belong to.  host4 (pinned, read by the host without synchronisation: a hint, never an input) =
// {moved, n, index of the sort call that counted, that streak}.
__global__ __launch_bounds__(1024) void k_coh_scan(int n, int ntiles, const uint32_t *__restrict__ tile_cnt,
                                                   uint32_t *__restrict__ tile_sbase, int32_t *ctl, int32_t *host4,
                                                   int call)
{
    __shared__ uint32_t s_scan[17];
    const int tid = threadIdx.x;
    uint32_t carry = 0;
    for (int t0 = 0; t0 < ntiles; t0 += 1024) {
        const int t = t0 + tid;
        const uint32_t c = t < ntiles ? tile_cnt[t] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan_u32<16>(c, s_scan, &tot);
        if (t < ntiles) tile_sbase[t] = carry + ex;
        carry += tot;
    }
    if (tid == 0) {
        const int moved = n - (int) carry;
        ctl[0] = moved;
        const bool tiny = (long long) moved * COH_TINY_SHARE <= (long long) n;
        const int streak = tiny ? (ctl[3] == n ? ctl[2] : 0) + 1 : 0;
        ctl[2] = streak;
        ctl[3] = n;
        __hip_atomic_store(&host4[3], streak, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&host4[2], call, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&host4[1], n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&host4[0], moved, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Staying elements before element `at` of tile `tile` (words = the tile's ballots in LDS or global).
__device__ __forceinline__ uint32_t coh_below(unsigned long long word, int lane)
{
    return (uint32_t) __popcll(word & ((1ull << lane) - 1ull));
}

// The moved elements, in index order, into (mk, mv).
template <typename KeyT>
__global__ __launch_bounds__(256) void k_coh_collect(const KeyT *__restrict__ knew, int n,
                                                     const unsigned long long *__restrict__ bits,
                                                     const uint32_t *__restrict__ tile_sbase,
                                                     KeyT *__restrict__ mk, uint32_t *__restrict__ mv)
{
    __shared__ unsigned long long s_w[COH_WORDS];
    __shared__ uint32_t s_pre[COH_WORDS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int tile = blockIdx.x, base = tile * COH_TILE;
    const int tile_n = min(COH_TILE, n - base);
    if (tid < COH_WORDS) s_w[tid] = tid * 64 < tile_n ? bits[(base >> 6) + tid] : 0ull;
    __syncthreads();
    if (tid == 0) {
        uint32_t a = 0;
        for (int q = 0; q < COH_WORDS; ++q) { s_pre[q] = a; a += (uint32_t) __popcll(s_w[q]); }
    }
    __syncthreads();
    const uint32_t sb = tile_sbase[tile];
#pragma unroll
    for (int j = 0; j < COH_ITEMS; ++j) {
        const int i = base + j * 256 + tid;
        const unsigned long long word = s_w[j * 4 + w];
        if (i < n && !((word >> lane) & 1ull)) {
            const uint32_t srank = sb + s_pre[j * 4 + w] + coh_below(word, lane);
            const uint32_t m = (uint32_t) i - srank;
            mk[m] = knew[i];
            mv[m] = (uint32_t) i;
        }
    }
}

// Final places.  Workgroups [0, ntiles): the staying elements of one tile; the other
// COH_MERGE_EXTRA workgroups: the moved elements, one wave each at a time.
template <typename KeyT>
__global__ __launch_bounds__(256) void k_coh_merge(const KeyT *__restrict__ knew, const KeyT *__restrict__ kold,
                                                   int n, int ntiles,
                                                   const unsigned long long *__restrict__ bits,
                                                   const uint32_t *__restrict__ tile_sbase,
                                                   const int32_t *__restrict__ ctl,
                                                   const KeyT *__restrict__ mk, const uint32_t *__restrict__ mv,
                                                   KeyT *__restrict__ kout, uint32_t *__restrict__ vout)
{
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int K = ctl[0];
    if ((int) blockIdx.x >= ntiles) {
        // ---- moved elements: j-th of the sorted ones -> j + staying elements below it ----
        const int wave0 = ((int) blockIdx.x - ntiles) * 4 + w, nwaves = ((int) gridDim.x - ntiles) * 4;
        for (int j = wave0; j < K; j += nwaves) {
            const KeyT km = mk[j];
            const uint32_t im = mv[j];
            // p = first index whose upper interval bound m_p is >= km (m_{n-1} = infinity): 64 probes a round
            int lo = 0, hi = n - 1;
            while (lo < hi) {
                const int chunk = (hi - lo) / 64 + 1;   // 64 * chunk > hi - lo: the last lane probes hi
                const int q = min(lo + (lane + 1) * chunk - 1, hi);
                bool ok = true;
                if (q < n - 1) {
                    const KeyT a = kold[q], b = kold[q + 1];
                    ok = a + ((b - a) >> 1) >= km;
                }
                const unsigned long long okm = __ballot(ok);   // not empty: the last lane probes hi
                const int f = (int) __builtin_ctzll(okm);
                const int qf = min(lo + (f + 1) * chunk - 1, hi);
                if (f > 0) lo = min(lo + f * chunk - 1, hi) + 1;
                hi = qf;
            }
            const int p = lo;
            const int t = p / COH_TILE, wi = (p % COH_TILE) >> 6;
            const unsigned long long word = lane <= wi ? bits[(size_t) t * COH_WORDS + lane] : 0ull;
            uint32_t c = lane < wi ? (uint32_t) __popcll(word) : (lane == wi ? coh_below(word, p & 63) : 0u);
            c = wave_sum_u32(c) + tile_sbase[t];
            const unsigned long long wp = __shfl(word, wi, 64);
            if ((wp >> (p & 63)) & 1ull) {
                const KeyT kp = knew[p];
                if (kp < km || (kp == km && (uint32_t) p < im)) ++c;
            }
            if (lane == 0) {
                kout[(uint32_t) j + c] = km;
                vout[(uint32_t) j + c] = im;
            }
        }
        return;
    }
    // ---- staying elements: staying rank + moved elements below (key, index) ----
    __shared__ unsigned long long s_w[COH_WORDS];
    __shared__ uint32_t s_pre[COH_WORDS];
    __shared__ int s_m[2];
    const int tile = blockIdx.x, base = tile * COH_TILE;
    const int tile_n = min(COH_TILE, n - base);
    if (tid < COH_WORDS) s_w[tid] = tid * 64 < tile_n ? bits[(base >> 6) + tid] : 0ull;
    if (tid == 64 || tid == 128) {
        // the moved keys that can fall among this tile's staying keys lie in [O[base-1], O[end]]
        int lo = 0, hi = K;
        if (tid == 64) {
            if (base > 0) { const KeyT b = kold[base - 1]; while (lo < hi) { const int mid = (lo + hi) >> 1; if (mk[mid] < b) lo = mid + 1; else hi = mid; } }
            s_m[0] = base > 0 ? lo : 0;
        } else {
            if (base + COH_TILE < n) { const KeyT b = kold[base + COH_TILE]; while (lo < hi) { const int mid = (lo + hi) >> 1; if (mk[mid] <= b) lo = mid + 1; else hi = mid; } }
            s_m[1] = base + COH_TILE < n ? lo : K;
        }
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t a = 0;
        for (int q = 0; q < COH_WORDS; ++q) { s_pre[q] = a; a += (uint32_t) __popcll(s_w[q]); }
    }
    __syncthreads();
    const uint32_t sb = tile_sbase[tile];
    const int mlo = s_m[0], mhi = s_m[1];
#pragma unroll
    for (int j = 0; j < COH_ITEMS; ++j) {
        const int i = base + j * 256 + tid;
        const unsigned long long word = s_w[j * 4 + w];
        if (i < n && ((word >> lane) & 1ull)) {
            const KeyT k = knew[i];
            int lo = mlo, hi = mhi;   // first moved element of [mlo, mhi) not below (k, i)
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const KeyT km = mk[mid];
                if (km < k || (km == k && mv[mid] < (uint32_t) i)) lo = mid + 1; else hi = mid;
            }
            const uint32_t pos = sb + s_pre[j * 4 + w] + coh_below(word, lane) + (uint32_t) lo;
            kout[pos] = k;
            vout[pos] = (uint32_t) i;
        }
    }
}

constexpr int SORT_TICKET_TILES = 256;

#ifndef NB_SORT_SKIP_DIGITS
#define NB_SORT_SKIP_DIGITS 3
#endif

// P0 = low digits left to the tie fix (0 = every digit gets a pass)
template <typename KeyT, int NPASS, int P0>
static hipError_t sort_impl(Sim *s)
{
    const int n = s->n;
    const int tiles = (n + SORT_TILE - 1) / SORT_TILE;   // <= s->sort_tiles (sized for cap)
    const int rows = sort_status_rows(tiles);            // tile rows + group rows of one pass
    const size_t zbytes = (8 * 256 + 16 + (size_t) (NPASS - P0) * rows * 256) * sizeof(uint32_t);
    // one memset for this sort's scratch and for the tree build's, which lies just before it --
    // unless the kick-drift kernel of this step has zeroed them on its way (scratch_zero_bytes)
    if (s->scratch_prezeroed != s->zero_bytes + zbytes) {
        hipError_t e = hipMemsetAsync(s->zero_block, 0, s->zero_bytes + zbytes, s->stream);
        if (e != hipSuccess) return e;
    }
    s->scratch_prezeroed = 0;
    s->build_scratch_zeroed = true;
    // layout of the zeroed block: ghist[8*256] | tickets[8] (+pad to 16) | status[P][rows][256]
    uint32_t *tickets = s->ghist + 8 * 256;
    uint32_t *status = tickets + 16;
    const int hgrid = std::min(std::max((n + NB_HIST_THREADS * NB_HIST_KEYS - 1) / (NB_HIST_THREADS * NB_HIST_KEYS), 1), NB_HIST_GRID);
    // (test aid: late workgroups and an early give-up exercise the helping path of the look-back)
    const int late = (s->p.flags & NBODY_FLAG_DEBUG_LATE_TILES) ? 1 : 0;
    const uint32_t help = late ? LB_TEST_SPINS : LB_HELP_SPINS;

    // Nearly sorted already?  (see k_coh_flags)  The previous sort's keys must be those of the
    // current particle order: same count, result in keys[1], new keys in keys[0].
    bool coherent = false, probe = false;
    if constexpr (sizeof(KeyT) == 8 && P0 > 0) {
        // (not while the stream is being captured into a graph: this path trades buffers between
        // calls and asks an event how far the device is, neither of which a replay repeats)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s->stream, &cap) != hipSuccess) cap = hipStreamCaptureStatusActive;
        if (s->coh_ctl && s->old_keys_n == n && s->sorted_buf == 1 && n >= 4 * COH_TILE &&
            cap == hipStreamCaptureStatusNone && !(s->p.flags & NBODY_FLAG_NO_COHERENT_SORT)) {
            if (s->p.flags & NBODY_FLAG_ALWAYS_COHERENT_SORT) {
                coherent = true;
            } else {
                // The moved count of the PREVIOUS sort, if that sort has finished by now (a caller
                // that queues many steps without waiting is ahead of the device, and a count from
                // many steps ago says little: a system starting from rest has none moved in its
                // first step and a quarter a hundred steps later) -- or an older count, up to
                // COH_TINY_AGE sorts old, when it and the counts before it found hardly anything
                // moved (a cold or slow system: queued steps keep the short sort).  Decides which
                // way is faster, never what comes out.
                const int hint = s->coh_host[0], hint_n = s->coh_host[1];
                const uint32_t hint_call = (uint32_t) s->coh_host[2];
                const int streak = s->coh_host[3];
                const uint32_t age = s->sort_calls - hint_call;
                const bool ok = hint_n == n && hint >= 0 && (long long) hint * COH_MAX_SHARE <= (long long) n;
                const bool fresh = age == 1 && s->hint_call + 1 == s->sort_calls && s->ev_hint &&
                                   hipEventQuery(s->ev_hint) == hipSuccess;
                const bool calm = streak >= COH_TINY_STREAK && (long long) hint * COH_TINY_SHARE <= (long long) n &&
                                  age >= 1 && age <= (uint32_t) COH_TINY_AGE;
                if (ok && (fresh || calm)) {
                    coherent = true;
                    s->probe_interval = 8;
                } else if (s->sort_calls <= 4u || s->sort_calls >= s->next_probe) {
                    // look now and then: the first four sorts, then at growing intervals (a system
                    // that never qualifies pays one counting pass in 64 sorts)
                    probe = true;
                    if (s->sort_calls > 4u) s->probe_interval = std::min(2 * s->probe_interval, 64u);
                    s->next_probe = s->sort_calls + s->probe_interval;
                }
            }
        }
    }
    ++s->sort_calls;
    const int ctiles = (n + COH_TILE - 1) / COH_TILE;
    unsigned long long *bits = (unsigned long long *) s->rt_le;      // n / 8 bytes of the n * 4
    uint32_t *tile_cnt = (uint32_t *) s->rt_left, *tile_sbase = tile_cnt + ctiles;
    if constexpr (sizeof(KeyT) == 8 && P0 > 0) {
    if (coherent || probe) {
        hipLaunchKernelGGL(k_coh_flags<KeyT>, dim3(ctiles), dim3(256), 0, s->stream, (const KeyT *) s->keys[0],
                           (const KeyT *) s->keys[1], n, bits, tile_cnt);
        hipLaunchKernelGGL(k_coh_scan, dim3(1), dim3(1024), 0, s->stream, n, ctiles, (const uint32_t *) tile_cnt,
                           tile_sbase, s->coh_ctl, s->coh_host, (int) (s->sort_calls - 1));
        if (s->ev_hint && hipEventRecord(s->ev_hint, s->stream) == hipSuccess) s->hint_call = s->sort_calls - 1;
    }
    if (coherent) {
        // moved elements: (unique, vals[0]) <-> (leaf_key, vals[1]); NPASS - P0 = 5 passes end in the second pair
        KeyT *kb[2] = {(KeyT *) s->unique, (KeyT *) s->leaf_key};
        uint32_t *vb[2] = {s->vals[0], s->vals[1]};
        const int32_t *n_moved = s->coh_ctl;
        hipLaunchKernelGGL(k_coh_collect<KeyT>, dim3(ctiles), dim3(256), 0, s->stream, (const KeyT *) s->keys[0], n,
                           (const unsigned long long *) bits, (const uint32_t *) tile_sbase, kb[0], vb[0]);
        hipLaunchKernelGGL((k_sort_hist<KeyT, NPASS, P0>), dim3(hgrid), dim3(NB_HIST_THREADS), 0, s->stream,
                           (const KeyT *) kb[0], 0, s->ghist, n_moved);
        int in = 0;
        for (int p = P0; p < NPASS; ++p) {
            uint32_t *st = status + (size_t) (p - P0) * rows * 256;
            hipLaunchKernelGGL((k_sort_pass<KeyT, SORT_ITEMS, false, true>), dim3(std::min(tiles, COH_SORT_GRID)), dim3(SORT_THREADS), 0,
                               s->stream, (const KeyT *) kb[in], (const uint32_t *) vb[in], kb[in ^ 1], vb[in ^ 1],
                               (uint32_t *) nullptr, n_moved, 0, 8 * p, s->ghist + p * 256, st,
                               st + (size_t) tiles * 256, tickets + p, help, late, &s->counts->err);
            in ^= 1;
        }
        hipLaunchKernelGGL(k_tiefix<KeyT>, dim3(std::min((n + 255) / 256, 1024)), dim3(256), 0, s->stream, kb[in], vb[in],
                           kb[in ^ 1], vb[in ^ 1], 0, 8 * P0, n_moved);
        // merged result -> the first pair, which then becomes keys[1] / vals[1]; the old sorted
        // keys' buffer serves as the unique-key buffer from here on
        hipLaunchKernelGGL(k_coh_merge<KeyT>, dim3(ctiles + COH_MERGE_EXTRA), dim3(256), 0, s->stream,
                           (const KeyT *) s->keys[0], (const KeyT *) s->keys[1], n, ctiles,
                           (const unsigned long long *) bits, (const uint32_t *) tile_sbase, n_moved,
                           (const KeyT *) kb[in], (const uint32_t *) vb[in], kb[in ^ 1], vb[in ^ 1]);
        static_assert(((NPASS - P0) & 1) == 1, "an odd number of passes ends in the second buffer pair");
        std::swap(s->keys[1], s->unique);
        std::swap(s->vals[0], s->vals[1]);
        s->sorted_buf = 1;
        s->old_keys_n = n;
        ++s->coh_sorts;
        return hipGetLastError();
    }
    }

    hipLaunchKernelGGL((k_sort_hist<KeyT, NPASS, P0>), dim3(hgrid), dim3(NB_HIST_THREADS), 0, s->stream,
                       (const KeyT *) s->keys[0], n, s->ghist, (const int32_t *) nullptr);
    // numbering by ticket up to a few hundred tiles, by blockIdx (with helping) beyond: a pass of
    // 128 tiles 14.4 against 15.6 us, of 512 tiles 25.5 against 22.4 us; the test mode always
    // takes the second
    const bool by_ticket = !late && tiles <= SORT_TICKET_TILES;
    int in = 0;
    for (int p = P0; p < NPASS; ++p) {
        const KeyT *kin = (const KeyT *) s->keys[in];
        KeyT *kout = (KeyT *) s->keys[in ^ 1];
        uint32_t *st = status + (size_t) (p - P0) * rows * 256;
        uint32_t *gst = st + (size_t) tiles * 256;
        const uint32_t *vin = p == P0 ? (const uint32_t *) nullptr : (const uint32_t *) s->vals[in];
#define NB_PASS(IOTA, HELP)                                                                          \
        hipLaunchKernelGGL((k_sort_pass<KeyT, SORT_ITEMS, IOTA, HELP>), dim3(tiles), dim3(SORT_THREADS), 0, \
                           s->stream, kin, vin, kout, s->vals[in ^ 1], (uint32_t *) nullptr,           \
                           (const int32_t *) nullptr, n, 8 * p, s->ghist + p * 256, st, gst,           \
                           tickets + p, help, late, &s->counts->err)
        if (p == P0) { if (by_ticket) NB_PASS(true, false); else NB_PASS(true, true); }
        else         { if (by_ticket) NB_PASS(false, false); else NB_PASS(false, true); }
#undef NB_PASS
        in ^= 1;
    }
    s->sorted_buf = in;
    s->old_keys_n = n;
    ++s->full_sorts;
    if (P0 > 0) {
        // runs of equal upper bits, ordered where they lie (the other buffer pair is scratch)
        KeyT *k0 = (KeyT *) s->keys[in], *k1 = (KeyT *) s->keys[in ^ 1];
        hipLaunchKernelGGL(k_tiefix<KeyT>, dim3((n + 255) / 256), dim3(256), 0, s->stream, k0, s->vals[in], k1,
                           s->vals[in ^ 1], n, 8 * P0, (const int32_t *) nullptr);
    }
    return hipGetLastError();
}

template <typename KeyT, int NPASS, int P0>
static size_t sort_zero_bytes(const Sim *s)
{
    const int tiles = (s->n + SORT_TILE - 1) / SORT_TILE;
    return s->zero_bytes + (8 * 256 + 16 + (size_t) (NPASS - P0) * sort_status_rows(tiles) * 256) * sizeof(uint32_t);
}

size_t scratch_zero_bytes(const Sim *s)
{
    return s->key_bytes == 4 ? sort_zero_bytes<uint32_t, 4, 0>(s) : sort_zero_bytes<uint64_t, 8, NB_SORT_SKIP_DIGITS>(s);
}

hipError_t launch_sort(Sim *s)
{
    return s->key_bytes == 4 ? sort_impl<uint32_t, 4, 0>(s) : sort_impl<uint64_t, 8, NB_SORT_SKIP_DIGITS>(s);
}

}  // namespace nb

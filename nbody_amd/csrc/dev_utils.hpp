// dev_utils.hpp -- wave64 / workgroup primitives shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nb {

constexpr int WAVE = 64;

// Device-side error word (one per handle).  Kernels only ever OR bits into it.
enum : uint32_t {
    DEV_ERR_SPIN = 1u,       // a look-back spin hit its bound
    DEV_ERR_STACK = 2u,      // traversal stack invariant violated
    DEV_ERR_RANGE = 4u       // an index left its array
};

__device__ __forceinline__ int lane_id() { return (int) __lane_id(); }

// Number of set bits of `mask` below the calling lane.
__device__ __forceinline__ int mbcnt(unsigned long long mask)
{
    return (int) __builtin_amdgcn_mbcnt_hi((unsigned) (mask >> 32),
                                           __builtin_amdgcn_mbcnt_lo((unsigned) mask, 0u));
}

// Morton bit spreading (reference: expand_bits, src/cuda/points.cu:3-12).
// 10-bit -> 30-bit spread (bit k -> bit 3k).
__device__ __forceinline__ uint32_t spread3_10(uint32_t u)
{
    u &= 0x3ffu;
    u = (u | (u << 16)) & 0x030000FFu;
    u = (u | (u << 8)) & 0x0300F00Fu;
    u = (u | (u << 4)) & 0x030C30C3u;
    u = (u | (u << 2)) & 0x09249249u;
    return u;
}

// 21-bit -> 63-bit spread (bit k -> bit 3k).
__device__ __forceinline__ uint64_t spread3_21(uint64_t u)
{
    u &= 0x1fffffull;
    u = (u | (u << 32)) & 0x001f00000000ffffull;
    u = (u | (u << 16)) & 0x001f0000ff0000ffull;
    u = (u | (u << 8)) & 0x100f00f00f00f00full;
    u = (u | (u << 4)) & 0x10c30c30c30c30c3ull;
    u = (u | (u << 2)) & 0x1249249249249249ull;
    return u;
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

// Inclusive prefix sum across the 64 lanes.
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v)
{
    const int l = lane_id();
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        uint32_t t = __shfl_up(v, o, WAVE);
        if (l >= o) v += t;
    }
    return v;
}

// Exclusive prefix sum over a workgroup of NW waves (blockDim.x == 64*NW).
// `scratch` is NW+1 words of LDS.  Returns the exclusive prefix; *total = workgroup sum.
// Contains two __syncthreads(); all threads must call it.
template <int NW>
__device__ __forceinline__ uint32_t block_excl_scan_u32(uint32_t v, uint32_t *scratch,
                                                        uint32_t *total)
{
    const int l = lane_id();
    const int w = (int) threadIdx.x >> 6;
    uint32_t inc = wave_incl_scan_u32(v);
    if (l == WAVE - 1) scratch[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        uint32_t s = scratch[i];
        if (i < w) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// ---------------------------------------------------------------------------
// Single-pass chained scan ("decoupled look-back") over tiles of one launch.
// A tile publishes one 32-bit word per scanned quantity:
//   bits 31..30: 0 = not ready, 1 = tile aggregate, 2 = inclusive prefix
//   bits 29..0 : value
// The word is the whole hand-off (flag and payload travel in one naturally aligned
// store), written and polled with relaxed agent-scope atomics, so no fence is needed.
//
// Forward progress.  A tile waits for words of LOWER tiles only, so the lowest unfinished tile
// never waits for anything that is not on its way -- provided somebody is working on it.  The
// usual guarantee is an atomic ticket: tile ids are drawn in the order workgroups start, so
// every predecessor of a running tile is itself running.  That counter serialises (atomics on
// one address complete ~15 ns apart on this chip: 8 us before the last of 512 tiles may begin).
// Kernels instantiated with HELP = true take tile = blockIdx instead and guarantee progress by
// HELPING: a workgroup whose wait exceeds `max_spins` polls gives the wait up (lb2_*<true> report
// "stuck"),
// looks for the lowest tile that has published nothing (lb2_lowest_unpublished) and, if there
// is one below its own, processes THAT tile -- all of it, as its owner would -- before it starts
// its own again.  Everything a tile publishes and writes is a pure function of the kernel's
// input, so a tile done twice (by a helper and, later, by its late owner) stores the same values
// twice; a status word may fall back from "inclusive prefix" to "aggregate" in between, which a
// reader handles like any aggregate.  With workgroups started in index order -- what the
// hardware does -- nobody ever helps; if they are not, the lowest unpublished tile is picked up
// by whoever waits for it, whatever the start order.  Every spin is bounded.  Launches of few
// tiles keep the ticket (HELP = false): the queue is short there, and the look-back that cannot
// give up is about a microsecond faster per launch.
// ---------------------------------------------------------------------------
constexpr uint32_t LB_AGG = 1u << 30;
constexpr uint32_t LB_PREFIX = 2u << 30;
constexpr uint32_t LB_MASK = (1u << 30) - 1u;

__device__ __forceinline__ void lb_store(uint32_t *p, uint32_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t lb_load(const uint32_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Poll one status word until its flag is set.  HELP: ... or max_spins polls have gone by, and a
// word with flag 0 is returned then (the caller gives the wait up).  Not HELP (tiles numbered by
// ticket: the owner of the word is running): waits on, and only a wait beyond any plausible
// length ends with DEV_ERR_SPIN and an empty prefix.  The two are separate instantiations
// because the mere possibility of giving up changes the code around every wait (measured: a
// digit pass of 128 tiles 15.6 against 14.4 us).
constexpr uint32_t LB_SPIN_LIMIT = 1u << 24;
template <bool HELP>
__device__ __forceinline__ uint32_t lb_poll(const uint32_t *p, uint32_t max_spins, uint32_t *err)
{
    uint32_t w = lb_load(p);
    uint32_t spins = 0;
    if constexpr (HELP) {
        while ((w >> 30) == 0u && ++spins <= max_spins) {
            __builtin_amdgcn_s_sleep(1);
            w = lb_load(p);
        }
    } else {
        while ((w >> 30) == 0u) {
            if (++spins > LB_SPIN_LIMIT) {
                atomicOr(err, (uint32_t) DEV_ERR_SPIN);
                return LB_PREFIX;  // give up: value 0, chain terminates
            }
            __builtin_amdgcn_s_sleep(1);
            w = lb_load(p);
        }
    }
    return w;
}
constexpr uint32_t LB_HELP_SPINS = 1u << 11;   // polls (a microsecond or two each) before a wait is given up
constexpr uint32_t LB_HELP_ROUNDS = 1u << 12;  // give-ups before a workgroup reports DEV_ERR_SPIN
constexpr uint32_t LB_TEST_SPINS = 8;          // the give-up threshold of the test mode with late workgroups

// Lowest tile below `tile` that has published nothing yet (word q of its status row), or `tile`.
// Called by all threads of a workgroup; s_min is one word of LDS.  Two __syncthreads().
__device__ __forceinline__ int lb2_lowest_unpublished(const uint32_t *tile_status, int stride, int q,
                                                      int tile, int *s_min)
{
    if (threadIdx.x == 0) *s_min = tile;
    __syncthreads();
    int lowest = tile;
    for (int t = (int) threadIdx.x; t < tile; t += (int) blockDim.x)
        if ((lb_load(&tile_status[(size_t) t * stride + q]) >> 30) == 0u) { lowest = t; break; }
    if (lowest < tile) atomicMin(s_min, lowest);
    __syncthreads();
    return *s_min;
}

// Two levels, for launches whose tiles are all resident at once (a few hundred to a few
// thousand tiles on 256 CUs).  With a one-level chain the aggregates of all tiles appear at
// about the same moment and tile t then reads its t predecessors 64 at a time -- t / 64 memory
// round trips for the last tiles.  Here tiles form groups of LBW_GROUP: a tile sums the
// aggregates of the earlier tiles of its group (one round trip), the last tile of a group
// publishes the group's total, and the chained look-back runs over one word per GROUP (64
// groups per round trip).  One scanned quantity, one wave: all 64 lanes must call it
// (wave-uniform arguments).  *stuck: a wait was given up (see above); the result is then void.
constexpr int LBW_GROUP = 32;

template <bool HELP>
__device__ __forceinline__ uint32_t lb2_exclusive_wave(uint32_t *tile_status, uint32_t *group_status,
                                                       int tile, uint32_t aggregate, uint32_t max_spins,
                                                       bool *stuck, uint32_t *err)
{
    const int l = lane_id();
    const int g = tile / LBW_GROUP, j = tile % LBW_GROUP;
    *stuck = false;
    if (l == 0) lb_store(&tile_status[tile], LB_AGG | aggregate);
    uint32_t v = LB_AGG;      // lanes without a predecessor: aggregate 0
    if (l < j) v = lb_poll<HELP>(&tile_status[g * LBW_GROUP + l], max_spins, err);
    if constexpr (HELP) {
        if (__any((v >> 30) == 0u)) { *stuck = true; return 0u; }
    }
    const uint32_t in_excl = wave_sum_u32(v & LB_MASK);
    const bool leader = j == LBW_GROUP - 1;
    const uint32_t group_total = in_excl + aggregate;
    if (g == 0) {
        if (leader && l == 0) lb_store(&group_status[0], LB_PREFIX | group_total);
        return in_excl;
    }
    if (leader && l == 0) lb_store(&group_status[g], LB_AGG | group_total);
    uint32_t gex = 0;
    int base = g - 1;
    while (true) {
        const int t = base - l;
        uint32_t w = LB_PREFIX;  // virtual groups before group 0: prefix 0
        if (t >= 0) w = lb_poll<HELP>(&group_status[t], max_spins, err);
        if constexpr (HELP) {
            if (__any((w >> 30) == 0u)) { *stuck = true; return 0u; }
        }
        const unsigned long long pm = __ballot((w >> 30) == 2u);
        const int first = pm ? (int) __builtin_ctzll(pm) : WAVE;  // nearest group with a prefix
        gex += wave_sum_u32((l <= first) ? (w & LB_MASK) : 0u);
        if (pm) break;
        base -= WAVE;
    }
    if (leader && l == 0) lb_store(&group_status[g], LB_PREFIX | (gex + group_total));
    return gex + in_excl;
}

// The chained look-back over the group words requests LB_WINDOW of them at a time.
#ifndef NB_LB_WINDOW
#define NB_LB_WINDOW 8
#endif
constexpr int LB_WINDOW = NB_LB_WINDOW;

// Two-level look-back by a single thread, for one quantity out of `stride` per tile (the radix
// pass: thread d follows digit d).  Tiles form groups of LB_GROUP consecutive tiles.  A tile publishes its aggregate, sums the aggregates
// of the earlier tiles of its own group (they need nothing from anybody, so they are all
// there after one round trip) and adds the exclusive prefix of its GROUP, which it finds by
// the chained look-back over one word per group: the last tile of a group publishes the
// group's total, looks back, and publishes the group's inclusive prefix.  When every tile of
// a launch is resident at once (a few hundred tiles on 256 CUs) a one-level chain makes
// tile t walk over nearly all of its t predecessors; here a tile reads at most LB_GROUP - 1
// tile words and ntiles / LB_GROUP group words.  *stuck: a wait was given up; the result is void.
// tile_status: one row of `stride` words per tile; group_status: one row per group.
#ifndef NB_LB_GROUP
#define NB_LB_GROUP 16
#endif
constexpr int LB_GROUP = NB_LB_GROUP;

// Two calls, so that the caller can do work that needs no global offset between them while
// the aggregates travel.
__device__ __forceinline__ void lb2_publish(uint32_t *tile_status, int stride, int q, int tile,
                                            uint32_t aggregate)
{
    lb_store(&tile_status[(size_t) tile * stride + q], LB_AGG | aggregate);
}

template <bool HELP>
__device__ __forceinline__ uint32_t lb2_resolve(uint32_t *tile_status, uint32_t *group_status,
                                                int stride, int q, int tile, uint32_t aggregate,
                                                uint32_t max_spins, bool *stuck, uint32_t *err)
{
    const int g = tile / LB_GROUP, j = tile % LB_GROUP;
    *stuck = false;
    // (the group words are requested only when the tile words are in: asked for any earlier, with
    // every tile of a launch in the same phase, they are mostly not there yet and cost a second
    // round trip each)
    uint32_t w[LB_GROUP - 1], wg[LB_WINDOW];
    const uint32_t *row = tile_status + (size_t) g * LB_GROUP * stride + q;
#pragma unroll
    for (int i = 0; i < LB_GROUP - 1; ++i) w[i] = i < j ? lb_load(row + (size_t) i * stride) : LB_AGG;
    uint32_t in_excl = 0;
#pragma unroll
    for (int i = 0; i < LB_GROUP - 1; ++i) {
        uint32_t v = w[i];
        if constexpr (HELP) {
            if ((v >> 30) == 0u && !*stuck) v = lb_poll<true>(row + (size_t) i * stride, max_spins, err);
            if ((v >> 30) == 0u) *stuck = true;
        } else {
            if ((v >> 30) == 0u) v = lb_poll<false>(row + (size_t) i * stride, max_spins, err);
        }
        in_excl += v & LB_MASK;
    }
    if constexpr (HELP) {
        if (*stuck) return 0u;
    }
    const bool leader = j == LB_GROUP - 1;
    const uint32_t group_total = in_excl + aggregate;   // meaningful for the leader only
    uint32_t gex = 0;
    if (g > 0) {
        if (leader) lb_store(&group_status[(size_t) g * stride + q], LB_AGG | group_total);
        int t = g - 1;
        bool done = false;
        while (!done && t >= 0) {
#pragma unroll
            for (int i = 0; i < LB_WINDOW; ++i)
                wg[i] = (t - i >= 0) ? lb_load(&group_status[(size_t) (t - i) * stride + q]) : LB_PREFIX;
#pragma unroll
            for (int i = 0; i < LB_WINDOW; ++i) {
                if (done) break;
                uint32_t v = wg[i];
                if ((v >> 30) == 0u) v = lb_poll<HELP>(&group_status[(size_t) (t - i) * stride + q], max_spins, err);
                if constexpr (HELP) {
                    if ((v >> 30) == 0u) { *stuck = true; return 0u; }
                }
                gex += v & LB_MASK;
                done = (v >> 30) == 2u;
            }
            t -= LB_WINDOW;
        }
    }
    if (leader) lb_store(&group_status[(size_t) g * stride + q], LB_PREFIX | (gex + group_total));
    return gex + in_excl;
}

// Bijective block-id remap that puts consecutive logical ids on the same XCD (blocks
// are dealt round-robin over the 8 XCDs, each with a private L2).  Speed only.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks)
{
    const int q = nblocks >> 3, r = nblocks & 7;
    const int x = bid & 7, k = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

}  // namespace nb

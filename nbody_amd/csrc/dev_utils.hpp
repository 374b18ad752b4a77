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
// Tile ids must be handed out in launch order by an atomic ticket so that every
// predecessor of a running tile is itself running; each spin is bounded.
// ---------------------------------------------------------------------------
constexpr uint32_t LB_AGG = 1u << 30;
constexpr uint32_t LB_PREFIX = 2u << 30;
constexpr uint32_t LB_MASK = (1u << 30) - 1u;
constexpr uint32_t LB_SPIN_LIMIT = 1u << 24;

__device__ __forceinline__ void lb_store(uint32_t *p, uint32_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t lb_load(const uint32_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Poll one status word until its flag is set (bounded).
__device__ __forceinline__ uint32_t lb_wait(const uint32_t *p, uint32_t *err)
{
    uint32_t w = lb_load(p);
    uint32_t spins = 0;
    while ((w >> 30) == 0u) {
        if (++spins > LB_SPIN_LIMIT) {
            atomicOr(err, (uint32_t) DEV_ERR_SPIN);
            return LB_PREFIX;  // give up: value 0, chain terminates
        }
        __builtin_amdgcn_s_sleep(1);
        w = lb_load(p);
    }
    return w;
}

// Wave-parallel look-back for ONE scanned quantity.  Must be called by all 64 lanes of
// one wave (wave-uniform arguments).  status has one word per tile.  Returns the
// exclusive prefix of `tile` and publishes its inclusive prefix.
__device__ __forceinline__ uint32_t lb_exclusive_wave(uint32_t *status, int tile,
                                                      uint32_t aggregate, uint32_t *err)
{
    const int l = lane_id();
    if (tile == 0) {
        if (l == 0) lb_store(&status[0], LB_PREFIX | aggregate);
        return 0u;
    }
    if (l == 0) lb_store(&status[tile], LB_AGG | aggregate);
    uint32_t excl = 0;
    int base = tile - 1;
    while (true) {
        const int t = base - l;
        uint32_t w = LB_PREFIX;  // virtual tiles before tile 0: prefix 0
        if (t >= 0) w = lb_wait(&status[t], err);
        const unsigned long long pm = __ballot((w >> 30) == 2u);
        const int first = pm ? (int) __builtin_ctzll(pm) : WAVE;  // nearest tile with a prefix
        const uint32_t v = (l <= first) ? (w & LB_MASK) : 0u;
        excl += wave_sum_u32(v);
        if (pm) break;
        base -= WAVE;
    }
    if (l == 0) lb_store(&status[tile], LB_PREFIX | (excl + aggregate));
    return excl;
}

// Two-level variant for launches whose tiles are all resident at once (a few hundred to a few
// thousand tiles on 256 CUs).  With the one-level chain above the aggregates of all tiles
// appear at about the same moment and tile t then reads its t predecessors 64 at a time --
// t / 64 memory round trips for the last tiles.  Here tiles form groups of LBW_GROUP: a tile
// sums the aggregates of the earlier tiles of its group (one round trip), the last tile of a
// group publishes the group's total, and the chained look-back runs over one word per GROUP
// (64 groups per round trip).  Same protocol and bounds as above; all 64 lanes must call it.
constexpr int LBW_GROUP = 32;

__device__ __forceinline__ uint32_t lb2_exclusive_wave(uint32_t *tile_status, uint32_t *group_status,
                                                       int tile, uint32_t aggregate, uint32_t *err)
{
    const int l = lane_id();
    const int g = tile / LBW_GROUP, j = tile % LBW_GROUP;
    if (l == 0) lb_store(&tile_status[tile], LB_AGG | aggregate);
    uint32_t v = 0;
    if (l < j) v = lb_wait(&tile_status[g * LBW_GROUP + l], err) & LB_MASK;
    const uint32_t in_excl = wave_sum_u32(v);
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
        if (t >= 0) w = lb_wait(&group_status[t], err);
        const unsigned long long pm = __ballot((w >> 30) == 2u);
        const int first = pm ? (int) __builtin_ctzll(pm) : WAVE;  // nearest group with a prefix
        gex += wave_sum_u32((l <= first) ? (w & LB_MASK) : 0u);
        if (pm) break;
        base -= WAVE;
    }
    if (leader && l == 0) lb_store(&group_status[g], LB_PREFIX | (gex + group_total));
    return gex + in_excl;
}

// All tiles of a launch are resident at once, so inclusive prefixes spread from tile 0 at
// LB_WINDOW tiles per memory round trip: the window, not the tile count, sets the pace.
#ifndef NB_LB_WINDOW
#define NB_LB_WINDOW 8
#endif
constexpr int LB_WINDOW = NB_LB_WINDOW;

// Look-back by a single thread for one quantity out of `stride` per tile (used by the radix
// pass: thread d follows digit d).  The words of the 8 nearest unread predecessors are
// requested together, so the walk costs one memory round trip per LB_WINDOW tiles, not per tile.
__device__ __forceinline__ uint32_t lb_exclusive_thread(uint32_t *status, int stride, int q,
                                                        int tile, uint32_t aggregate,
                                                        uint32_t *err)
{
    if (tile == 0) {
        lb_store(&status[q], LB_PREFIX | aggregate);
        return 0u;
    }
    lb_store(&status[(size_t) tile * stride + q], LB_AGG | aggregate);
    uint32_t excl = 0;
    int t = tile - 1;
    bool done = false;
    while (!done && t >= 0) {
        uint32_t w[LB_WINDOW];
#pragma unroll
        for (int i = 0; i < LB_WINDOW; ++i)
            w[i] = (t - i >= 0) ? lb_load(&status[(size_t) (t - i) * stride + q]) : LB_PREFIX;
#pragma unroll
        for (int i = 0; i < LB_WINDOW; ++i) {
            if (done) break;
            uint32_t v = w[i];
            if ((v >> 30) == 0u) v = lb_wait(&status[(size_t) (t - i) * stride + q], err);
            excl += v & LB_MASK;
            done = (v >> 30) == 2u;
        }
        t -= LB_WINDOW;
    }
    lb_store(&status[(size_t) tile * stride + q], LB_PREFIX | (excl + aggregate));
    return excl;
}

// Two-level look-back by a single thread (same role as lb_exclusive_thread).  Tiles form
// groups of LB_GROUP consecutive tiles.  A tile publishes its aggregate, sums the aggregates
// of the earlier tiles of its own group (they need nothing from anybody, so they are all
// there after one round trip) and adds the exclusive prefix of its GROUP, which it finds by
// the chained look-back over one word per group: the last tile of a group publishes the
// group's total, looks back, and publishes the group's inclusive prefix.  When every tile of
// a launch is resident at once (a few hundred tiles on 256 CUs) the one-level chain makes
// tile t walk over nearly all of its t predecessors; here a tile reads at most LB_GROUP - 1
// tile words and ntiles / LB_GROUP group words.
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

__device__ __forceinline__ uint32_t lb2_resolve(uint32_t *tile_status, uint32_t *group_status,
                                                int stride, int q, int tile, uint32_t aggregate,
                                                uint32_t *err)
{
    const int g = tile / LB_GROUP, j = tile % LB_GROUP;
    uint32_t in_excl = 0;
    {
        uint32_t w[LB_GROUP - 1];
        const uint32_t *row = tile_status + (size_t) g * LB_GROUP * stride + q;
#pragma unroll
        for (int i = 0; i < LB_GROUP - 1; ++i) w[i] = i < j ? lb_load(row + (size_t) i * stride) : LB_AGG;
#pragma unroll
        for (int i = 0; i < LB_GROUP - 1; ++i) {
            uint32_t v = w[i];
            if ((v >> 30) == 0u) v = lb_wait(row + (size_t) i * stride, err);
            in_excl += v & LB_MASK;
        }
    }
    const bool leader = j == LB_GROUP - 1;
    const uint32_t group_total = in_excl + aggregate;   // meaningful for the leader only
    uint32_t gex = 0;
    if (g > 0) {
        if (leader) lb_store(&group_status[(size_t) g * stride + q], LB_AGG | group_total);
        int t = g - 1;
        bool done = false;
        while (!done && t >= 0) {
            uint32_t w[LB_WINDOW];
#pragma unroll
            for (int i = 0; i < LB_WINDOW; ++i)
                w[i] = (t - i >= 0) ? lb_load(&group_status[(size_t) (t - i) * stride + q]) : LB_PREFIX;
#pragma unroll
            for (int i = 0; i < LB_WINDOW; ++i) {
                if (done) break;
                uint32_t v = w[i];
                if ((v >> 30) == 0u) v = lb_wait(&group_status[(size_t) (t - i) * stride + q], err);
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

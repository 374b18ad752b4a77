// k_traverse.hip -- theta-criterion group traversal + softened FP32 force accumulation,
// and the tiled O(N^2) direct sum.
//
// Semantics restated from the reference (src/cuda/barnes_hut.cu:31-61,199-537;
// include/cuda/physics_common.cuh:36-75; src/cuda/validator.cu:134-174):
//   * a group is G consecutive Morton-sorted bodies and shares one interaction list;
//   * for a candidate node, d2min = min over the group's bodies of |x_b - com|^2,
//     starting from 3*domain^2;
//   * a node without children is evaluated directly against every particle of its range
//     (self included: contributes exactly 0 because eps^2 > 0);
//   * otherwise size^2 / d2min < theta^2 -> the group interacts with (com, mass), else the
//     node's children become candidates;
//   * acc = gravity * sum m_j d / (|d|^2 + eps^2)^(3/2).
// Design here: one wave64 per group, no workgroup barriers, no global-memory queues.
//   * Inner-node candidates live on a per-wave LDS stack.  Each round pops up to 64 of them,
//     one per lane, and every lane tests its node against the group's bodies (broadcast
//     LDS reads, two bodies per packed-FP32 instruction).
//   * Accepted (com, mass) records go to an LDS source list.  Children of opened nodes are
//     split by the parent's child-leaf mask: inner children return to the node stack, leaf
//     children go to a separate LDS leaf stack and never pay for a distance test.
//   * Leaves are taken 64 at a time in memory order; leaves that are adjacent in the sorted
//     particle array (siblings always are) fuse into runs.  A run is consumed straight from
//     the pair-interleaved particle array through the SCALAR data path: the run bounds are
//     wave-uniform, so each (x0,x1,y0,y1,z0,z1,m0,m1) record is one s_load_dwordx8 into
//     SGPRs that feed the packed force instructions directly -- no LDS staging, no vector
//     address arithmetic, the next record is requested before the current one is evaluated.
//     Only the odd particle at either end of a run goes through the LDS list.
//   * Whenever the source list holds >= 64 entries each lane (= one body) runs the force
//     loop over 32 broadcast entry PAIRS: v_pk_{add,mul,fma}_f32 evaluate two sources per
//     instruction, v_rsq_f32 once per source.
// The set of sources a group interacts with does not depend on the order candidates are
// visited in, so list lengths equal those of the reference's breadth-first walk.
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "dev_utils.hpp"
#include "nbody_internal.hpp"

namespace nb {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v8f __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

// Development aid, off in the shipped library: -DNB_WALK_PROFILE=1 makes every wave add the
// shader-clock cycles it spends in each phase to g_walk_prof and the launcher print the sums.
#ifndef NB_WALK_PROFILE
#define NB_WALK_PROFILE 0
#endif
#if NB_WALK_PROFILE
__device__ unsigned long long g_walk_prof[8];
// per group: wall-clock begin and end (100 MHz ticks) and the hardware id of its wave; the
// launcher writes them to the file named by NB_WALK_TRACE (scratch/trace_an.py reads it)
__device__ unsigned long long g_walk_trace[3 * 65536];
#define PROF_T0() const long long t0_ = clock64()
#define PROF_ADD(slot) do { if (lane == 0) prof_acc[slot] += clock64() - t0_; } while (0)
#else
#define PROF_T0() do {} while (0)
#define PROF_ADD(slot) do {} while (0)
#endif

constexpr int TRAV_RESERVE = 7 * 24;  // single-pop fallback never needs more than 7/level

// Per-wave LDS working set (7936 B with the sizes in nbody_internal.hpp).  Bodies and sources are stored as PAIRS,
// (x0,x1,y0,y1) and (z0,z1[,m0,m1]), so that one 64-bit register pair feeds one packed op.
struct alignas(16) WaveLds {
    v4f bxy[32];
    v2f bz[32];
    v4f lxy[64];   // source ring: 128 entries = 64 pairs
    v4f lzm[64];
    int stack[TRAV_STACK];
    int leaf[TRAV_LEAF_STACK];
    int pend[TRAV_PEND];   // candidates waiting for the exact distance test
};

// MODE: 0 = forces (the step), 1 = walk only (profiling aid: no force arithmetic), 2 = potential
// (sum of m / sqrt(d^2 + eps^2) over the same sources instead of the force, sources at zero
// distance -- the body itself -- left out; writes tail->pot and nothing else).
constexpr int TRAV_FORCE = 0, TRAV_WALK_ONLY = 1, TRAV_POTENTIAL = 2;

// UNI != 0 (G = 64, forces only): every particle has the same mass m0.  Particles streamed through
// the scalar path then skip the multiplication by their mass; the masses of the ring's entries
// (accepted nodes, odd particles at run ends) are divided by m0 on their way in and the caller
// passes gravity * m0, so that every sum is in units of m0.  With m0 = 1 (or any power of two)
// the result is the same bits as the general path's.  UNI == 2: particle records from index
// uni_limit on (imported tree copies, where a pruned subtree travels as one record of its own
// mass) keep their mass: it is divided by m0 like a ring entry's.
template <int G, bool FUSE_VEL, int MODE, int UNI = 0>
__global__ __launch_bounds__(64 * TRAV_WAVES, TRAV_OCC) void k_traverse(
    int n, int g0, int g1, const Float4 *__restrict__ body_posm, const Float4 *__restrict__ posm,
    const v8f *__restrict__ posm2, const Float4 *__restrict__ nodeA,
    const NodeB *__restrict__ nodeB, float theta, float eps2, float gravity, float domain,
    float dt, float keep, const TravTail *__restrict__ tail, int cur, int accumulate,
    int stack_limit, const int32_t *__restrict__ roots, int nroots, float inv_m0, int uni_limit)
{
    static_assert(!UNI || (G == 64 && MODE == TRAV_FORCE), "uniform-mass path: wave-wide groups, forces");
    __shared__ WaveLds lds[TRAV_WAVES];

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // Groups are taken in Morton order, XCD-contiguous: workgroups are dealt round-robin over
    // the 8 XCDs, so the remap gives every XCD one contiguous eighth of the groups, and groups
    // that run at the same time on an XCD are neighbours in space and share their sources in
    // its L2.
    const int lblk = xcd_remap((int) blockIdx.x, (int) gridDim.x);
    const int g = g0 + lblk * TRAV_WAVES + w;
    if (g >= g1) return;
    WaveLds &L = lds[w];
    float *bxyf = reinterpret_cast<float *>(L.bxy);
    float *bzf = reinterpret_cast<float *>(L.bz);
    float *lxyf = reinterpret_cast<float *>(L.lxy);
    float *lzmf = reinterpret_cast<float *>(L.lzm);

    // G == 32: both half-waves hold the same 32 bodies; the lower half evaluates the even source
    // pairs of the ring, the upper half the odd ones, and the two partial sums meet at the end.
    const int b = g * G + (G == 32 ? (lane & 31) : lane);
    const bool active = lane < G && b < n;
    const Float4 me = body_posm[b < n ? b : g * G];
    if (lane < G) {
        const int p = lane >> 1, h = lane & 1;
        bxyf[4 * p + h] = me.x;
        bxyf[4 * p + 2 + h] = me.y;
        bzf[2 * p + h] = me.z;
    }
#if NB_WALK_PROFILE
    long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long t_begin = clock64();
    const long long r_begin = wall_clock64();
#endif
    int sp = 0, leaf_n = 0;
    {
        if (!(accumulate & 4)) {   // (4: a pass over imported copies alone leaves the own root out)
            const bool root_leaf = (nodeB[0].meta & NODE_LEAF) != 0u;
            if (lane == 0) { L.stack[0] = 0; L.leaf[0] = 0; }
            if (root_leaf) leaf_n = 1; else sp = 1;
        }
        // a forest: the roots of the pruned copies of other domains' trees that stand behind
        // this tree's nodes in the same arrays (k_let_graft); nroots <= LET_MAX_BOXES < stack size
        // (every root is a candidate, leaves included: see the opening of leaves below)
        for (int r0 = 0; r0 < nroots; r0 += 64) {
            const int k = min(64, nroots - r0);
            if (lane < k) L.stack[sp + lane] = roots[r0 + lane];
            sp += k;
        }
    }
    __builtin_amdgcn_wave_barrier();

    v2f fx2 = {0.f, 0.f}, fy2 = {0.f, 0.f}, fz2 = {0.f, 0.f};
    const v2f px2 = {me.x, me.x}, py2 = {me.y, me.y}, pz2 = {me.z, me.z};
    const v2f e2 = {eps2, eps2};
    int head = 0, list_n = 0;   // source ring: entries [head, head + list_n) mod 128
    uint32_t total = 0;
    const float th2 = theta * theta;
    const float dinit = domain * domain * 3.0f;

    // Potential mode: 1/sqrt(s) of a source pair.  A source at zero distance (s == eps^2) is
    // kept out of the float sum -- the body itself is one, its own leaf always delivers it, and
    // its m_i / eps can be 10^3 times all the other terms together (a heavy body, a small
    // softening), which would cost the sum its low digits -- and its mass goes to msame2
    // instead.  At the end msame - m_i is the mass of the OTHER sources at zero distance
    // (bodies at the same position, a node whose centre of mass falls on the body): exactly 0
    // in nearly every case, and their m / eps is added then.
    v2f msame2 = {0.f, 0.f};
    auto pot_rsq = [&](const v2f s, const v2f m) -> v2f {
        const bool z0 = s.x == eps2, z1 = s.y == eps2;
        msame2 += v2f{z0 ? m.x : 0.f, z1 ? m.y : 0.f};
        return v2f{z0 ? 0.f : __builtin_amdgcn_rsqf(s.x), z1 ? 0.f : __builtin_amdgcn_rsqf(s.y)};
    };
    auto pair_force = [&](const v4f a, const v4f c) {
        if (MODE == TRAV_WALK_ONLY) {  // keep the LDS reads alive, skip the arithmetic
            fx2 += a.xy; fy2 += c.xy;
            return;
        }
        const v2f dx = a.xy - px2, dy = a.zw - py2, dz = c.xy - pz2;
        const v2f s = __builtin_elementwise_fma(
            dz, dz, __builtin_elementwise_fma(dy, dy, __builtin_elementwise_fma(dx, dx, e2)));
        if (MODE == TRAV_POTENTIAL) {   // fx2 doubles as the potential accumulator
            fx2 = __builtin_elementwise_fma(pot_rsq(s, c.zw), c.zw, fx2);
            return;
        }
        const v2f s3 = s * s * s;
        v2f inv = {__builtin_amdgcn_rsqf(s3.x), __builtin_amdgcn_rsqf(s3.y)};
        inv *= c.zw;
        fx2 = __builtin_elementwise_fma(dx, inv, fx2);
        fy2 = __builtin_elementwise_fma(dy, inv, fy2);
        fz2 = __builtin_elementwise_fma(dz, inv, fz2);
    };
    auto write_entry = [&](int e, const Float4 v) {
        const int p = e >> 1, h = e & 1;
        lxyf[4 * p + h] = v.x;
        lxyf[4 * p + 2 + h] = v.y;
        lzmf[4 * p + h] = v.z;
        lzmf[4 * p + 2 + h] = UNI ? v.w * inv_m0 : v.w;
    };
    auto flush64 = [&]() {
        PROF_T0();
        __builtin_amdgcn_wave_barrier();
        const int hp = head >> 1;
        if constexpr (G == 32) {
            const int half = lane >> 5;
#pragma unroll 8
            for (int j = 0; j < 16; ++j) pair_force(L.lxy[hp + 2 * j + half], L.lzm[hp + 2 * j + half]);
        } else {
#pragma unroll 8
            for (int j = 0; j < 32; ++j) pair_force(L.lxy[hp + j], L.lzm[hp + j]);
        }
        head ^= 64;
        list_n -= 64;
        __builtin_amdgcn_wave_barrier();
        PROF_ADD(2);
    };

    // Sources of one pair record held in SGPRs (wave-uniform): no LDS, no VGPR staging.
    auto pair_force_s = [&](const v8f r, auto own_mass) {   // own_mass: the record's mass counts (not UNI's m0)
        if (MODE == TRAV_WALK_ONLY) { fx2 += v2f{r[0], r[1]}; return; }
        const v2f dx = v2f{r[0], r[1]} - px2, dy = v2f{r[2], r[3]} - py2, dz = v2f{r[4], r[5]} - pz2;
        const v2f s = __builtin_elementwise_fma(
            dz, dz, __builtin_elementwise_fma(dy, dy, __builtin_elementwise_fma(dx, dx, e2)));
        if (MODE == TRAV_POTENTIAL) {
            fx2 = __builtin_elementwise_fma(pot_rsq(s, v2f{r[6], r[7]}), v2f{r[6], r[7]}, fx2);
            return;
        }
        const v2f s3 = s * s * s;
        v2f inv = {__builtin_amdgcn_rsqf(s3.x), __builtin_amdgcn_rsqf(s3.y)};
        if constexpr (!UNI) inv *= v2f{r[6], r[7]};
        else if constexpr (decltype(own_mass)::value) inv *= v2f{r[6], r[7]} * inv_m0;
        fx2 = __builtin_elementwise_fma(dx, inv, fx2);
        fy2 = __builtin_elementwise_fma(dy, inv, fy2);
        fz2 = __builtin_elementwise_fma(dz, inv, fz2);
    };
    // Append the particles selected by `want` (one per lane, index idx) to the source ring.
    auto append_particles = [&](bool want, int idx) {
        const unsigned long long wm = __ballot(want);
        if (wm == 0ull) return;
        if (want) write_entry((head + list_n + mbcnt(wm)) & 127, posm[idx]);
        list_n += (int) __popcll(wm);
        if (list_n >= 64) flush64();
    };

    // Take up to 64 leaves (memory order) from the leaf stack and evaluate their particles.
    auto process_leaves = [&]() {
        const int cnt = min(64, leaf_n);
        const bool valid = lane < cnt;
        const int id = valid ? L.leaf[leaf_n - cnt + lane] : 0;
        leaf_n -= cnt;
        const NodeB Bl = nodeB[id];
        const int pb = Bl.first;
        const int pe = pb + Bl.count;                    // exclusive end
        const int prev_end = __shfl_up(pe, 1, 64);
        const int next_pb = __shfl_down(pb, 1, 64);
        const bool is_head = valid && (lane == 0 || prev_end != pb);
        const bool is_tail = valid && (lane == cnt - 1 || next_pb != pe);
        if constexpr (G == 32) {
            // Half-wave split: wave-uniform SGPR sources would make both halves evaluate the
            // same pairs, so runs are staged through the LDS ring instead, 32 pair records
            // (1 KB, one 16-byte piece per lane) at a time and always as whole pairs: a run
            // that starts on the second particle of a pair or ends on the first gets the mass
            // of the foreign particle replaced by zero on the way in.
            (void) is_tail;
            const v4f *P4 = reinterpret_cast<const v4f *>(posm2);
            unsigned long long hm = __ballot(is_head);
            PROF_T0();
            // the runs of the batch as one stream of 32-pair pieces; the piece after the one
            // being evaluated is always in flight, across run boundaries too
            int rb = 0, re = 0, p0 = 0, p1 = 0, c0 = 0;
            auto next_run = [&]() -> bool {
                while (hm) {
                    const int h = (int) __builtin_ctzll(hm);
                    hm &= hm - 1;
                    const int t = (hm ? (int) __builtin_ctzll(hm) : cnt) - 1;   // last lane of the run
                    rb = __builtin_amdgcn_readlane(pb, h);
                    re = __builtin_amdgcn_readlane(pe, t);
                    total += (uint32_t) (re - rb);
                    p0 = rb >> 1;
                    p1 = (re + 1) >> 1;                                         // pairs [p0, p1)
                    c0 = p0;
                    if (p0 < p1) return true;
                }
                return false;
            };
            const int pp = lane >> 1, part = lane & 1;
            auto fetch = [&]() -> v4f {       // this lane's 16 bytes of the piece at c0
                const int gp = c0 + pp;
                v4f d = {0.f, 0.f, 0.f, 0.f};
                if (gp < p1) {
                    d = P4[2 * (size_t) gp + part];
                    if (part) {
                        if (gp == p0 && (rb & 1)) d.z = 0.f;
                        if (gp == p1 - 1 && (re & 1)) d.w = 0.f;
                    }
                }
                return d;
            };
            bool have = next_run();
            v4f data = {0.f, 0.f, 0.f, 0.f};
            if (have) data = fetch();
            while (have) {
                const int np = min(32, p1 - c0);
                if (c0 + 32 < p1) c0 += 32; else have = next_run();
                if (list_n & 1) {   // keep the ring pair-aligned
                    if (lane == 0) write_entry((head + list_n) & 127, Float4{0.f, 0.f, 0.f, 0.f});
                    list_n += 1;
                }
                if (list_n >= 64) flush64();
                const int wp = ((head + list_n) & 127) >> 1;
                if (pp < np) {
                    const int slot = (wp + pp) & 63;
                    if (part) L.lzm[slot] = data; else L.lxy[slot] = data;
                }
                list_n += 2 * np;
                if (have) data = fetch();   // in flight during the flush
                if (list_n >= 64) flush64();
            }
            PROF_ADD(1);
            return;
        }
        // The scalar stream below has one request in flight per wave, so a miss in L2 stalls
        // it for a full memory round trip.  Touch every 128-byte line of this lane's leaf with
        // an ordinary vector load first: all 64 leaves of the batch are then on their way into
        // L2 at once, long before the stream reaches them.  The values are not used.
        float touch[4];
        {
            const char *lb = reinterpret_cast<const char *>(posm2) + (size_t) (pb >> 1) * 32;
            const int bytes = valid ? (((pe + 1) >> 1) - (pb >> 1)) * 32 : 0;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                touch[t] = t * 128 < bytes ? *reinterpret_cast<const float *>(lb + t * 128) : 0.f;
        }
        // odd particle at either end of a run -> LDS list (two batched gathers per 64 leaves)
        append_particles(is_head && (pb & 1), pb);
        append_particles(is_tail && (pe & 1), pe - 1);
        // the whole pairs of one run [rb, re) through the scalar path
        auto run_stream = [&](const int rb, const int re, auto own_mass) {
            int q0 = (rb + 1) >> 1;
            const int q1 = re >> 1;                                     // pairs [q0, q1)
            if (q0 < q1 && (q0 & 1)) pair_force_s(posm2[q0++], own_mass);         // align to 64 B
            if (q0 + 1 < q1) {   // two pairs (one 64-byte record) per scalar load, one load ahead;
                                 // two register sets used alternately so nothing is copied
                const v16f *P16 = reinterpret_cast<const v16f *>(posm2);
                const int d1 = q1 >> 1;
                int d = q0 >> 1;
                // Scalar loads return out of order, so the only wait there is, is "all of them":
                // wait for the record about to be used BEFORE requesting the next one, or the
                // wait the compiler places ahead of the first use would cover the new request
                // too and the prefetch would hide nothing.
                v16f ra = P16[d], rb2;
                while (true) {
                    const bool hb = d + 1 < d1;
                    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): ra has arrived
                    if (hb) rb2 = P16[d + 1];
                    pair_force_s(ra.lo, own_mass);
                    pair_force_s(ra.hi, own_mass);
                    if (!hb) break;
                    const bool ha = d + 2 < d1;
                    __builtin_amdgcn_s_waitcnt(0xC07F);   // rb2 has arrived
                    if (ha) ra = P16[d + 2];
                    pair_force_s(rb2.lo, own_mass);
                    pair_force_s(rb2.hi, own_mass);
                    if (!ha) break;
                    d += 2;
                }
                q0 = d1 << 1;
            }
            if (q0 < q1) pair_force_s(posm2[q0], own_mass);
        };
        // whole pairs of every run -> scalar path
        unsigned long long hm = __ballot(is_head);
        PROF_T0();
        while (hm) {
            const int h = (int) __builtin_ctzll(hm);
            hm &= hm - 1;
            const int t = (hm ? (int) __builtin_ctzll(hm) : cnt) - 1;   // last lane of the run
            const int rb = __builtin_amdgcn_readlane(pb, h);
            const int re = __builtin_amdgcn_readlane(pe, t);
            total += (uint32_t) (re - rb);
            if (UNI == 2 && rb >= uni_limit) run_stream(rb, re, std::true_type{});   // an imported copy's records
            else run_stream(rb, re, std::false_type{});
        }
        PROF_ADD(1);
        asm volatile("" ::"v"(touch[0]), "v"(touch[1]), "v"(touch[2]), "v"(touch[3]));
    };

    // Bounding box of the group's bodies and four sample body pairs: most candidates are
    // decided from these alone, with the SAME outcome as the exact test below --
    //   * the distance from the box to the centre of mass, computed with the same operation
    //     sequence as a body's distance, is <= every body's computed distance (each step is
    //     monotone under rounding), so  size^2/d_box^2 < theta^2  implies acceptance;
    //   * a sample body's computed distance is one of the terms of the exact minimum, so
    //     !(size^2/d_sample^2 < theta^2)  implies the node is opened.
    // Only the undecided rest is queued (L.pend) for the exact 64-body minimum, which is run
    // 32 candidates at a time with two lanes per candidate.
    float blo[3] = {me.x, me.y, me.z}, bhi[3] = {me.x, me.y, me.z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            blo[a] = fminf(blo[a], __shfl_xor(blo[a], o, 64));
            bhi[a] = fmaxf(bhi[a], __shfl_xor(bhi[a], o, 64));
        }
        // wave-uniform from here on: keep the box in scalar registers
        blo[a] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, blo[a])));
        bhi[a] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, bhi[a])));
    }
    int pend_n = 0;
    // exact d2min of this lane's candidate over the body pairs part, part+nparts, ...
    auto exact_dmin = [&](const Float4 A, int part, int nparts) -> float {
        const v2f ax2 = {A.x, A.x}, ay2 = {A.y, A.y}, az2 = {A.z, A.z};
        float dmin = dinit;
#pragma unroll 4
        for (int j = part; j < G / 2; j += nparts) {
            const v4f q = L.bxy[j];
            const v2f qz = L.bz[j];
            const v2f dx = q.xy - ax2, dy = q.zw - ay2, dz = qz - az2;
            const v2f d = __builtin_elementwise_fma(
                dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
            dmin = fminf(dmin, fminf(d.x, d.y));
        }
        return dmin;
    };

    uint32_t rounds = 0;
    while (true) {
        while (leaf_n >= 64 || (sp == 0 && pend_n == 0 && leaf_n > 0)) process_leaves();
        if (sp == 0 && pend_n == 0) break;
        if (++rounds > (1u << 22)) {   // every round retires a node for good; a tree has < 2^22
            if (lane == 0) atomicOr(tail->err, (uint32_t) DEV_ERR_STACK);   // of them per group walk
            break;
        }

        PROF_T0();
        // Node-stack discipline.  A round normally pops everything it can test (up to 64) and
        // pushes the inner children of the nodes it opens for as long as they fit under
        // stack_limit - TRAV_RESERVE; opened nodes whose children do not fit go back on the
        // stack unexpanded (they occupy slots the round has just freed).  With fewer than 7
        // free slots below that mark the round pops ONE node and always expands it: such
        // rounds are depth-first, add at most 7 entries net per tree level (8 children for the
        // node popped), and the reserve of 7 x 24 entries covers every level a 63-bit key has.
        const int room = stack_limit - TRAV_RESERVE - sp;
        const int leaf_room = TRAV_LEAF_STACK - leaf_n;
        bool valid, approx, may_defer;
        int node;
        Float4 A;
        NodeB B;
        // undecided candidates first once 32 have gathered (or nothing else is left), as long
        // as their children are sure to fit
        const int kx = (pend_n >= 32 || sp == 0) ? min(min(pend_n, 32), min(max(room, 0) >> 3, leaf_room >> 3)) : 0;
        if (kx > 0) {
            const int slot = lane & 31;
            const bool have = slot < kx;
            node = have ? L.pend[pend_n - kx + slot] : 0;
            pend_n -= kx;
            A = nodeA[node];
            B = nodeB[node];
            float dmin = exact_dmin(A, lane >> 5, 2);
            dmin = fminf(dmin, __shfl_xor(dmin, 32, 64));
            valid = have && lane < 32;
            approx = valid && (__fdiv_rn(B.size * B.size, dmin) < th2);
            may_defer = false;
        } else {
            may_defer = room >= 7;
            const int k = may_defer ? min(sp, 64) : 1;
            valid = lane < k;
            node = valid ? L.stack[sp - k + lane] : 0;
            sp -= k;
            A = nodeA[node];
            B = nodeB[node];
            const float sz2 = B.size * B.size;
            if (pend_n + k <= TRAV_PEND) {
                const float ex = fmaxf(0.f, fmaxf(blo[0] - A.x, A.x - bhi[0]));
                const float ey = fmaxf(0.f, fmaxf(blo[1] - A.y, A.y - bhi[1]));
                const float ez = fmaxf(0.f, fmaxf(blo[2] - A.z, A.z - bhi[2]));
                const float dbox = fminf(dinit, fmaf(ez, ez, fmaf(ey, ey, ex * ex)));
                approx = valid && (__fdiv_rn(sz2, dbox) < th2);
                const v2f ax2 = {A.x, A.x}, ay2 = {A.y, A.y}, az2 = {A.z, A.z};
                float ds = dinit;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int jj = t == 3 ? G / 2 - 1 : t * (G / 6);
                    const v4f q = L.bxy[jj];
                    const v2f qz = L.bz[jj];
                    const v2f dx = q.xy - ax2, dy = q.zw - ay2, dz = qz - az2;
                    const v2f d = __builtin_elementwise_fma(
                        dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
                    ds = fminf(ds, fminf(d.x, d.y));
                }
                const bool sure_open = !(__fdiv_rn(sz2, ds) < th2);
                const bool undecided = valid && !approx && !sure_open;
                const unsigned long long um = __ballot(undecided);
                if (undecided) L.pend[pend_n + mbcnt(um)] = node;
                pend_n += (int) __popcll(um);
                valid = valid && !undecided;
            } else {   // no room to defer anything: exact test for every candidate of the round
                approx = valid && (__fdiv_rn(sz2, exact_dmin(A, 0, 1)) < th2);
            }
        }
        const bool open = valid && !approx;

        // accepted nodes -> source ring
        const unsigned long long am = __ballot(approx);
        if (approx) write_entry((head + list_n + mbcnt(am)) & 127, A);
        const int na = (int) __popcll(am);
        list_n += na;
        total += (uint32_t) na;

        // opened nodes -> inner children to the node stack, leaf children to the leaf stack
        // A leaf among the candidates (only imported copies have them there: their senders
        // mark as "leaf child" only what every group of this domain accepts, so that a real leaf
        // is tested like any node and taken as one (com, mass) source by the groups far from
        // it) opens into itself: its id goes to the leaf stack, as its one leaf child.
        const bool self_leaf = open && (B.meta & NODE_LEAF) != 0u;
        const uint32_t nc = open ? (self_leaf ? 1u : (uint32_t) B.count) : 0u;
        const uint32_t lmask = open ? (self_leaf ? 1u : (B.meta & 0xffu)) : 0u;
        const uint32_t cl = (uint32_t) __popc(lmask);
        const uint32_t packed = (nc - cl) | (cl << 16);
        const uint32_t inc = wave_incl_scan_u32(packed);
        const uint32_t exc = inc - packed;
        // the lanes whose children fit are a prefix of the opened lanes (inc is monotone)
        // (a node has up to 8 inner children and the round has freed at least its own slot)
        const bool fits = open && (!may_defer || ((int) (inc & 0xffffu) <= room + 1 &&
                                                  (int) (inc >> 16) <= leaf_room));
        const unsigned long long fm = __ballot(fits);
        const unsigned long long bm = __ballot(open && !fits);
        const int n_back = (int) __popcll(bm);
        if (open && !fits) L.stack[sp + mbcnt(bm)] = node;
        const uint32_t tot = fm ? (uint32_t) __builtin_amdgcn_readlane((int) inc, 63 - (int) __builtin_clzll(fm)) : 0u;
        const int tot_i = (int) (tot & 0xffffu), tot_l = (int) (tot >> 16);
        if (sp + n_back + tot_i > TRAV_STACK || leaf_n + tot_l > TRAV_LEAF_STACK) {
            if (lane == 0) atomicOr(tail->err, (uint32_t) DEV_ERR_STACK);  // unreachable by construction
            break;
        }
        if (fits) {
            int di = sp + n_back + (int) (exc & 0xffffu), dl = leaf_n + (int) (exc >> 16);
            for (uint32_t c = 0; c < nc; ++c) {
                const int child = self_leaf ? node : B.first + (int) c;
                if ((lmask >> c) & 1u) L.leaf[dl++] = child; else L.stack[di++] = child;
            }
        }
        sp += n_back + tot_i;
        leaf_n += tot_l;
        __builtin_amdgcn_wave_barrier();
        PROF_ADD(0);
#if NB_WALK_PROFILE
        if (lane == 0) prof_acc[4] += 1;
#endif

        if (list_n >= 64) flush64();
    }

    // remainder (< 64 entries): pad an odd tail with a massless source
    if (list_n & 1) {
        if (lane == 0) write_entry((head + list_n) & 127, Float4{0.f, 0.f, 0.f, 0.f});
    }
    __builtin_amdgcn_wave_barrier();
    {
        const int hp = head >> 1;
        const int np2 = (list_n + 1) >> 1;
        if constexpr (G == 32) {
            const int half = lane >> 5;
            for (int j = half; j < np2; j += 2) pair_force(L.lxy[hp + j], L.lzm[hp + j]);
        } else {
            for (int j = 0; j < np2; ++j) pair_force(L.lxy[hp + j], L.lzm[hp + j]);
        }
    }
    float fx = fx2.x + fx2.y, fy = fy2.x + fy2.y, fz = fz2.x + fz2.y;
    if constexpr (G == 32) {   // the other half-wave holds the other half of the sources
        fx += __shfl_xor(fx, 32, 64);
        fy += __shfl_xor(fy, 32, 64);
        fz += __shfl_xor(fz, 32, 64);
    }

#if NB_WALK_PROFILE
    if (lane == 0) {
        prof_acc[3] = clock64() - t_begin;
        prof_acc[5] = wall_clock64() - r_begin;
        for (int i = 0; i < 6; ++i) atomicAdd(&g_walk_prof[i], (unsigned long long) prof_acc[i]);
        if (g < 65536) {
            g_walk_trace[3 * g] = (unsigned long long) r_begin;
            g_walk_trace[3 * g + 1] = (unsigned long long) wall_clock64();
            g_walk_trace[3 * g + 2] = (unsigned long long) __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
        }
    }
#endif
    if (MODE == TRAV_POTENTIAL) {
        float ms = msame2.x + msame2.y;
        if constexpr (G == 32) ms += __shfl_xor(ms, 32, 64);
        if (active) {
            // the body's own term never entered the sum; other sources at zero distance do now
            const float p = fx + (ms - me.w) * __builtin_amdgcn_rsqf(eps2);
            tail->pot[b] = (accumulate & 1) ? tail->pot[b] + p : p;
        }
        return;
    }
    if (lane == 0) tail->list_len[g] = total;
    if (active) {
        float axv = gravity * fx, ayv = gravity * fy, azv = gravity * fz;
        if (accumulate) { axv += tail->ax[b]; ayv += tail->ay[b]; azv += tail->az[b]; }
        tail->ax[b] = axv; tail->ay[b] = ayv; tail->az[b] = azv;
        if (FUSE_VEL) {
            tail->vx[cur][b] = fmaf(0.5f * axv, dt, tail->vhx[cur][b]) * keep;
            tail->vy[cur][b] = fmaf(0.5f * ayv, dt, tail->vhy[cur][b]) * keep;
            tail->vz[cur][b] = fmaf(0.5f * azv, dt, tail->vhz[cur][b]) * keep;
        }
    }
}

// *out (u64) = or += sum over groups g0..g1-1 of list_len[g] * bodies_in_group(g).  One workgroup
// (a few tens of thousands of groups): no partial sums to combine, so nothing to zero first.
__global__ __launch_bounds__(1024) void k_count_interactions(int n, int group, int g0, int g1,
                                                             const uint32_t *__restrict__ list_len,
                                                             unsigned long long *out, int add)
{
    __shared__ unsigned long long s[16];
    unsigned long long acc = 0;
    for (int gb = g0 + (int) threadIdx.x; gb < g1; gb += 8 * 1024) {   // eight loads in flight per thread
        uint32_t len[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) len[k] = list_len[min(gb + k * 1024, g1 - 1)];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int g = gb + k * 1024;
            if (g < g1) acc += (unsigned long long) len[k] * (unsigned long long) min(group, n - g * group);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int k = 0; k < 16; ++k) t += s[k];
        *out = add ? *out + t : t;
    }
}

// ---------------------------------------------------------------------------
// Direct sum: 256 targets per workgroup, sources staged through LDS 256 at a time.
// POT additionally accumulates pot_i = sum_j m_j / sqrt(|d|^2 + eps^2); with self_pairs
// set (targets and sources are the same array) the j = i term is left out (the force of that
// term is zero anyway).
// ---------------------------------------------------------------------------
template <bool POT>
__global__ __launch_bounds__(256) void k_direct(const Float4 *__restrict__ targets, int nt,
                                                const Float4 *__restrict__ sources, int ns,
                                                float eps2, float gravity, float *__restrict__ ax,
                                                float *__restrict__ ay, float *__restrict__ az,
                                                int accumulate, float *__restrict__ pot,
                                                int self_pairs)
{
    // sources staged 256 at a time in the pair layout (x0,x1,y0,y1 | z0,z1,m0,m1)
    __shared__ v4f s_xy[128];
    __shared__ v4f s_zm[128];
    float *xyf = reinterpret_cast<float *>(s_xy);
    float *zmf = reinterpret_cast<float *>(s_zm);
    const int i = blockIdx.x * 256 + threadIdx.x;
    const Float4 me = targets[i < nt ? i : 0];
    float fx = 0.f, fy = 0.f, fz = 0.f, ps = 0.f;
    v2f fx2 = {0.f, 0.f}, fy2 = {0.f, 0.f}, fz2 = {0.f, 0.f};
    const v2f px2 = {me.x, me.x}, py2 = {me.y, me.y}, pz2 = {me.z, me.z}, e2 = {eps2, eps2};
    for (int t0 = 0; t0 < ns; t0 += 256) {
        const int c = min(256, ns - t0);
        {
            const int t = (int) threadIdx.x;
            // a massless pad at the body's own distance scale keeps partial tiles branch-free
            const Float4 s = t < c ? sources[t0 + t] : Float4{0.f, 0.f, 0.f, 0.f};
            const int p = t >> 1, h = t & 1;
            xyf[4 * p + h] = s.x; xyf[4 * p + 2 + h] = s.y;
            zmf[4 * p + h] = s.z; zmf[4 * p + 2 + h] = s.w;
        }
        __syncthreads();
        const int np2 = (c + 1) >> 1;
        if (POT) {
            // with self_pairs, source i of the tile that holds this workgroup's own bodies is
            // body i itself: its mass is masked out (exact, whatever m_i / eps is beside the
            // rest of the sum)
            const bool own_tile = self_pairs && t0 == (int) blockIdx.x * 256;
            const int mine = (int) threadIdx.x >> 1, half = (int) threadIdx.x & 1;
            for (int j = 0; j < np2; ++j) {
                const v4f a = s_xy[j];
                v4f b = s_zm[j];
                if (own_tile && j == mine) { if (half) b.w = 0.f; else b.z = 0.f; }
                const v2f dx = a.xy - px2, dy = a.zw - py2, dz = b.xy - pz2;
                const v2f t = __builtin_elementwise_fma(
                    dz, dz, __builtin_elementwise_fma(dy, dy, __builtin_elementwise_fma(dx, dx, e2)));
                const v2f r = {__builtin_amdgcn_rsqf(t.x), __builtin_amdgcn_rsqf(t.y)};
                const v2f mr = b.zw * r;
                ps += mr.x + mr.y;
                const v2f inv = mr * r * r;
                fx2 = __builtin_elementwise_fma(dx, inv, fx2);
                fy2 = __builtin_elementwise_fma(dy, inv, fy2);
                fz2 = __builtin_elementwise_fma(dz, inv, fz2);
            }
        } else {
#pragma unroll 8
            for (int j = 0; j < np2; ++j) {
                const v4f a = s_xy[j], b = s_zm[j];
                const v2f dx = a.xy - px2, dy = a.zw - py2, dz = b.xy - pz2;
                const v2f t = __builtin_elementwise_fma(
                    dz, dz, __builtin_elementwise_fma(dy, dy, __builtin_elementwise_fma(dx, dx, e2)));
                const v2f t3 = t * t * t;
                v2f inv = {__builtin_amdgcn_rsqf(t3.x), __builtin_amdgcn_rsqf(t3.y)};
                inv *= b.zw;
                fx2 = __builtin_elementwise_fma(dx, inv, fx2);
                fy2 = __builtin_elementwise_fma(dy, inv, fy2);
                fz2 = __builtin_elementwise_fma(dz, inv, fz2);
            }
        }
        __syncthreads();
    }
    fx = fx2.x + fx2.y; fy = fy2.x + fy2.y; fz = fz2.x + fz2.y;
    if (i < nt) {
        float rx = gravity * fx, ry = gravity * fy, rz = gravity * fz;
        if (accumulate) { rx += ax[i]; ry += ay[i]; rz += az[i]; }
        ax[i] = rx; ay[i] = ry; az[i] = rz;
        if (POT) {
            pot[i] = accumulate ? pot[i] + ps : ps;
        }
    }
}

// Bodies of `s` against the tree of `src` (the same handle for an ordinary step).
// acc_flag: bit 0 = add to the accelerations, bit 2 = leave the own root out (forest only)
static hipError_t traverse_impl(Sim *s, Sim *src, bool fuse_vel, int acc_flag, bool potential = false,
                                bool forest = false)
{
    const int32_t *roots = forest ? src->imp_roots : nullptr;
    const int nroots = forest ? src->imp_nroots : 0;
    const int ngroups = (s->n + s->group - 1) / s->group;
    const int g0 = s->trav_g1 < 0 ? 0 : s->trav_g0;
    const int g1 = s->trav_g1 < 0 ? ngroups : std::min(s->trav_g1, ngroups);
    const int c = s->cur;
    if (g1 > g0) {
        const int ng = g1 - g0;
        // blocks are dealt round-robin over the 8 XCDs; every XCD gets the same number
        const int per_xcd = ((ng + TRAV_WAVES - 1) / TRAV_WAVES + 7) / 8;
        const int grid = per_xcd * 8;
        const float eps2 = s->p.softening_factor * s->p.softening_factor;
        const float keep = 1.0f - s->p.velocity_dampening;
#define NB_TRAV_U(GG, FV, WO, UU, GRAV, IM0)                                                     \
    hipLaunchKernelGGL((k_traverse<GG, FV, (WO), UU>), dim3(grid), dim3(64 * TRAV_WAVES), 0, s->stream, \
                       s->n, g0, g1, s->posm, src->posm, (const v8f *) src->posm2, src->nodeA,       \
                       src->nodeB, s->p.theta, eps2, GRAV, s->p.domain_size, s->p.dt, keep,          \
                       s->trav_tail, c, acc_flag, stack_limit, roots, nroots, IM0, uni_limit)
#define NB_TRAV(GG, FV, WO) NB_TRAV_U(GG, FV, WO, 0, s->p.gravity, 1.0f)
        // every particle of the handle has mass m0: the uniform-mass variant
        // (imported copies' records keep their own mass: the forest walk runs variant 2)
        const float m0 = (src == s && !(s->p.flags & NBODY_FLAG_NO_UNIFORM_MASS)) ? s->uniform_mass : 0.f;
        // (the copies' particle records start behind the handle's own: nbody_import_trees_dev)
        const int uni_limit = forest ? ((s->cap + 3) & ~3) : 0x7fffffff;
        const bool walk_only = (s->p.flags & NBODY_FLAG_DEBUG_WALK_ONLY) != 0;
        // test aid: a node stack with 32 usable entries makes unexpanded push-backs and
        // depth-first single-node rounds the common case
        const int stack_limit = (s->p.flags & NBODY_FLAG_DEBUG_SMALL_STACK) ? TRAV_RESERVE + 32 : TRAV_STACK;
        if (potential) {
            if (s->group == 64) NB_TRAV(64, false, TRAV_POTENTIAL); else NB_TRAV(32, false, TRAV_POTENTIAL);
        } else if (walk_only) {
            if (s->group == 64) NB_TRAV(64, false, TRAV_WALK_ONLY); else NB_TRAV(32, false, TRAV_WALK_ONLY);
        } else if (s->group == 64 && m0 > 0.f && forest) {
            if (fuse_vel) NB_TRAV_U(64, true, TRAV_FORCE, 2, s->p.gravity * m0, 1.0f / m0);
            else NB_TRAV_U(64, false, TRAV_FORCE, 2, s->p.gravity * m0, 1.0f / m0);
        } else if (s->group == 64 && m0 > 0.f) {
            if (fuse_vel) NB_TRAV_U(64, true, TRAV_FORCE, 1, s->p.gravity * m0, 1.0f / m0);
            else NB_TRAV_U(64, false, TRAV_FORCE, 1, s->p.gravity * m0, 1.0f / m0);
        } else if (s->group == 64) {
            if (fuse_vel) NB_TRAV(64, true, TRAV_FORCE); else NB_TRAV(64, false, TRAV_FORCE);
        } else {
            if (fuse_vel) NB_TRAV(32, true, TRAV_FORCE); else NB_TRAV(32, false, TRAV_FORCE);
        }
#undef NB_TRAV
#undef NB_TRAV_U
#if NB_WALK_PROFILE
        (void) hipStreamSynchronize(s->stream);
        unsigned long long h[8], z[8] = {0};
        (void) hipMemcpyFromSymbol(h, HIP_SYMBOL(g_walk_prof), sizeof h);
        (void) hipMemcpyToSymbol(HIP_SYMBOL(g_walk_prof), z, sizeof z);
        if (const char *path = getenv("NB_WALK_TRACE")) {
            std::vector<unsigned long long> tr(3 * 65536);
            (void) hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(g_walk_trace), tr.size() * 8);
            if (FILE *f = fopen(path, "wb")) { fwrite(tr.data(), 8, 3 * (size_t) std::min(g1 - g0, 65536), f); fclose(f); }
        }
        const double tot = (double) h[3];
        fprintf(stderr, "walk profile: groups %d  MAC rounds/group %.1f  wave-cycles/group %.0f | "
                        "MAC+push %.1f%%  leaf runs (scalar path) %.1f%%  ring flush %.1f%%  other %.1f%%  "
                        "| shader clock %.0f MHz (s_memtime ticks per 100 MHz tick)\n",
                g1 - g0, (double) h[4] / (g1 - g0), tot / (g1 - g0), 100.0 * h[0] / tot,
                100.0 * h[1] / tot, 100.0 * h[2] / tot, 100.0 * (tot - h[0] - h[1] - h[2]) / tot,
                100.0 * tot / (double) h[5]);
#endif
    }
    return hipGetLastError();
}

hipError_t launch_traverse(Sim *s, bool fuse_vel) { return traverse_impl(s, s, fuse_vel, 0); }

// tail->pot[b] = sum over the sources of the body's group of m / sqrt(d^2 + eps^2), own term
// removed; every group, whatever traversal range is set.
hipError_t launch_potential(Sim *s)
{
    const int g0 = s->trav_g0, g1 = s->trav_g1;
    s->trav_g1 = -1;
    const hipError_t e = traverse_impl(s, s, false, 2, true);
    s->trav_g0 = g0; s->trav_g1 = g1;
    return e;
}

hipError_t launch_traverse_from(Sim *bodies, Sim *sources, bool accumulate)
{
    return traverse_impl(bodies, sources, false, accumulate ? 1 : 0);
}

// what = 0: the own tree alone; 1: own tree and the grafted copies of the other domains' trees
// in one walk; 2: the copies alone, added.  Over the handle's traversal range.
hipError_t launch_traverse_forest(Sim *s, int what, bool kick)
{
    return traverse_impl(s, s, kick, what == 2 ? 5 : 0, false, what != 0);
}

// all_groups_add: count every group and add to *out (after a nbody_traverse_from call);
// otherwise *out = count over the traversal range of the step.
// range_add: count the groups of the traversal range and add to *out.
hipError_t launch_count_interactions(Sim *s, unsigned long long *out, bool all_groups_add, bool range_add)
{
    const int ngroups = (s->n + s->group - 1) / s->group;
    const int g0 = (all_groups_add || s->trav_g1 < 0) ? 0 : s->trav_g0;
    const int g1 = (all_groups_add || s->trav_g1 < 0) ? ngroups : std::min(s->trav_g1, ngroups);
    // (an empty range still has to leave 0 behind when it is not added)
    hipLaunchKernelGGL(k_count_interactions, dim3(1), dim3(1024), 0, s->stream, s->n, s->group,
                       g0, std::max(g0, g1), s->list_len, out, (all_groups_add || range_add) ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_direct(const Float4 *targets, int nt, const Float4 *sources, int ns, float eps2,
                         float gravity, float *ax, float *ay, float *az, int accumulate,
                         hipStream_t st)
{
    if (nt <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_direct<false>, dim3((nt + 255) / 256), dim3(256), 0, st, targets, nt,
                       sources, ns, eps2, gravity, ax, ay, az, accumulate, (float *) nullptr, 0);
    return hipGetLastError();
}

hipError_t launch_direct_pot(const Float4 *targets, int nt, const Float4 *sources, int ns,
                             float eps2, float gravity, float *ax, float *ay, float *az,
                             float *pot, int self_pairs, hipStream_t st)
{
    if (nt <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_direct<true>, dim3((nt + 255) / 256), dim3(256), 0, st, targets, nt,
                       sources, ns, eps2, gravity, ax, ay, az, 0, pot, self_pairs);
    return hipGetLastError();
}

}  // namespace nb

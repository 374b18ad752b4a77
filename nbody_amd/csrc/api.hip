// api.hip -- the C-ABI of libnbody_mi355.so (declared in include/nbody_c.h): handle
// lifetime, initial conditions, step sequencing, getters, timing.
//
// Step sequencing follows Simulation<T>::update() and Impl::update_points/update_octree/
// update_bodies_* (reference: src/cuda/simulation.cu:75-119,160-168,224-238); initial
// conditions follow InitialConditions<T> (include/cuda/initial_conditions.cuh:20-147) with
// the constants of Simulation<T>::spawn_* (src/cuda/simulation.cu:170-220).
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <new>
#include <type_traits>
#include <vector>

#include "nbody_internal.hpp"

using namespace nb;

namespace {

#define NB_HIP(sim, call)                                                                    \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            (sim)->err = std::string(#call) + ": " + hipGetErrorString(e_);                  \
            return NBODY_ERR_HIP;                                                            \
        }                                                                                    \
    } while (0)

int fail(Sim *s, int code, const std::string &msg)
{
    s->err = msg;
    return code;
}

template <typename T>
hipError_t dalloc(T **p, size_t count)
{
    return hipMalloc((void **) p, std::max<size_t>(count, 1) * sizeof(T));
}

void free_all(Sim *s)
{
    auto F = [](void *p) { if (p) (void) hipFree(p); };
    for (int i = 0; i < 2; ++i) {
        F(s->x[i]); F(s->y[i]); F(s->z[i]); F(s->m[i]); F(s->vhx[i]); F(s->vhy[i]); F(s->vhz[i]);
        F(s->vx[i]); F(s->vy[i]); F(s->vz[i]); F(s->id[i]); F(s->keys[i]); F(s->vals[i]);
    }
    F(s->ax); F(s->ay); F(s->az); F(s->posm); F(s->posm2); F(s->zero_block); F(s->com_counter);
    F(s->counts); F(s->unique); F(s->first_pt); F(s->lfc); F(s->leaf_key); F(s->rt_pd);
    F(s->rt_level); F(s->rt_lb); F(s->rt_le); F(s->rt_left); F(s->rt_right); F(s->depth_key);
    F(s->bfs); F(s->octid); F(s->o_depth); F(s->o_first_child); F(s->o_num_children); F(s->o_lb);
    F(s->o_le); F(s->o_level); F(s->o_pb); F(s->o_pe); F(s->o_parent); F(s->o_size); F(s->o_wsum);
    F(s->nodeA); F(s->nodeB); F(s->list_len); F(s->inter_dev); F(s->trav_tail); F(s->pot);
    F(s->tw_posm); F(s->tw_pot); F(s->ds_pot); F(s->let_boxes); F(s->let_counts);
    F(s->keybox); F(s->red_scratch); F(s->red_out); F(s->let_mask); F(s->let_seg); F(s->let_frontier);
    F(s->depth_rows); F(s->let_reach); F(s->let_bsum); F(s->imp_roots); F(s->imp_desc);
    if (s->ev_ready) (void) hipEventDestroy(s->ev_ready);
    if (s->ev_hint) (void) hipEventDestroy(s->ev_hint);
    if (s->red_host) (void) hipHostFree(s->red_host);
    for (int k = 0; k < 3; ++k) { F(s->tw_v[k]); F(s->tw_vh[k]); F(s->tw_a[k]); F(s->ds_a[k]); }
    State64 &d = s->d;
    for (int i = 0; i < 2; ++i) {
        F(d.x[i]); F(d.y[i]); F(d.z[i]); F(d.m[i]); F(d.vhx[i]); F(d.vhy[i]); F(d.vhz[i]);
        F(d.vx[i]); F(d.vy[i]); F(d.vz[i]);
    }
    F(d.ax); F(d.ay); F(d.az); F(d.posm); F(d.wsum); F(d.nodeA); F(d.pot); F(d.tw_posm);
    F(d.tw_pot); F(d.ds_pot);
    for (int k = 0; k < 3; ++k) { F(d.tw_v[k]); F(d.tw_vh[k]); F(d.tw_a[k]); F(d.ds_a[k]); }
    if (s->counts_host) (void) hipHostFree(s->counts_host);
    if (s->coh_host) (void) hipHostFree(s->coh_host);
    F(s->coh_ctl);
    for (int r = 0; r < TIMING_RING; ++r)
        for (int k = 0; k < EV_COUNT; ++k)
            if (s->ev[r][k]) (void) hipEventDestroy(s->ev[r][k]);
    if (s->stream) (void) hipStreamDestroy(s->stream);
}

// Particle arrays in the handle's precision: R = float uses the Sim members, R = double the
// State64 ones (same names).
template <typename R, typename St>
int alloc_particles(Sim *s, St &a)
{
    const size_t n = (size_t) s->cap;
    for (int i = 0; i < 2; ++i) {
        R **arr[] = {&a.x[i], &a.y[i], &a.z[i], &a.m[i], &a.vhx[i], &a.vhy[i], &a.vhz[i],
                     &a.vx[i], &a.vy[i], &a.vz[i]};
        for (R **q : arr) NB_HIP(s, dalloc(q, n));
        R *zero[] = {a.vhx[i], a.vhy[i], a.vhz[i], a.vx[i], a.vy[i], a.vz[i]};
        for (R *q : zero) NB_HIP(s, hipMemsetAsync(q, 0, n * sizeof(R), s->stream));
    }
    NB_HIP(s, dalloc(&a.ax, n)); NB_HIP(s, dalloc(&a.ay, n)); NB_HIP(s, dalloc(&a.az, n));
    NB_HIP(s, dalloc(&a.posm, n));
    return NBODY_OK;
}

int alloc_all(Sim *s)
{
    const size_t n = (size_t) s->cap;
    for (int i = 0; i < 2; ++i) {
        NB_HIP(s, dalloc(&s->id[i], n));
        NB_HIP(s, hipMalloc(&s->keys[i], n * s->key_bytes));
        NB_HIP(s, dalloc(&s->vals[i], n));
    }
    if (s->fp64) {
        int rc = alloc_particles<double>(s, s->d);
        if (rc) return rc;
    } else {
        int rc = alloc_particles<float>(s, *s);
        if (rc) return rc;
        // pair records are read two at a time (64 bytes): pad to a whole number of those
        const size_t posm2_floats = 16 * ((n + 3) / 4);
        NB_HIP(s, dalloc(&s->posm2, posm2_floats));
        NB_HIP(s, hipMemsetAsync(s->posm2, 0, posm2_floats * 4, s->stream));
    }

    // Scratch zeroed once per step, one allocation so that one memset does it:
    //   build part: tickets[16] | scan_status[2][2T] | dhist[256] | rank_status[R][256]
    //   sort part : ghist[8*256] | tickets[16] | status[P][rows][256]
    s->sort_tiles = (int) ((n + SORT_TILE - 1) / SORT_TILE);
    const int passes = s->key_bytes == 4 ? 4 : 8;
    s->sort_status_bytes = (8 * 256 + 16 + (size_t) passes * sort_status_rows(s->sort_tiles) * 256) * sizeof(uint32_t);
    s->max_scan_tiles = (int) ((n + 1023) / 1024);  // smallest compaction tile is 1024
    s->rank_tiles = (int) ((2 * n + SORT_TILE - 1) / SORT_TILE);
    // (a scan row = one word per tile, then one per group of tiles: dev_utils.hpp, lb2_exclusive_wave)
    const size_t zwords = 16 + 4 * (size_t) s->max_scan_tiles + 256 + (size_t) sort_status_rows(s->rank_tiles) * 256;
    s->zero_bytes = (zwords * 4 + 15) / 16 * 16;
    NB_HIP(s, hipMalloc((void **) &s->zero_block, s->zero_bytes + s->sort_status_bytes));
    s->ghist = (uint32_t *) (s->zero_block + s->zero_bytes);
    uint32_t *zw = (uint32_t *) s->zero_block;
    s->tickets = zw;
    s->scan_status = zw + 16;
    s->dhist = s->scan_status + 4 * (size_t) s->max_scan_tiles;
    s->rank_status = s->dhist + 256;
    if (s->key_bytes == 8) {
        NB_HIP(s, dalloc(&s->coh_ctl, 4));
        NB_HIP(s, hipMemsetAsync(s->coh_ctl, 0, 16, s->stream));
        NB_HIP(s, hipHostMalloc((void **) &s->coh_host, 4 * sizeof(int32_t)));
        s->coh_host[0] = -1; s->coh_host[1] = 0; s->coh_host[2] = 0; s->coh_host[3] = 0;
        NB_HIP(s, hipEventCreateWithFlags(&s->ev_hint, hipEventDisableTiming));
    }
    NB_HIP(s, dalloc(&s->depth_rows, (size_t) 2048 * 32));
    NB_HIP(s, dalloc(&s->com_counter, 2 * n));
    NB_HIP(s, hipMemsetAsync(s->com_counter, 0, 2 * n * 4, s->stream));

    NB_HIP(s, dalloc(&s->counts, 1));
    NB_HIP(s, hipMemsetAsync(s->counts, 0, sizeof(Counts), s->stream));
    NB_HIP(s, hipHostMalloc((void **) &s->counts_host, sizeof(Counts)));

    NB_HIP(s, hipMalloc(&s->unique, n * s->key_bytes));
    NB_HIP(s, hipMalloc(&s->leaf_key, n * s->key_bytes));
    NB_HIP(s, dalloc(&s->first_pt, n + 1)); NB_HIP(s, dalloc(&s->lfc, n + 1));
    NB_HIP(s, dalloc(&s->rt_pd, 2 * n)); NB_HIP(s, dalloc(&s->rt_level, 2 * n));
    NB_HIP(s, dalloc(&s->rt_lb, n)); NB_HIP(s, dalloc(&s->rt_le, n));
    NB_HIP(s, dalloc(&s->rt_left, n)); NB_HIP(s, dalloc(&s->rt_right, n));
    NB_HIP(s, dalloc(&s->depth_key, 2 * n)); NB_HIP(s, dalloc(&s->bfs, 2 * n));
    NB_HIP(s, dalloc(&s->octid, 2 * n)); NB_HIP(s, dalloc(&s->o_depth, 2 * n));
    NB_HIP(s, dalloc(&s->o_first_child, 2 * n)); NB_HIP(s, dalloc(&s->o_num_children, 2 * n));
    NB_HIP(s, dalloc(&s->o_lb, 2 * n)); NB_HIP(s, dalloc(&s->o_le, 2 * n));
    NB_HIP(s, dalloc(&s->o_level, 2 * n)); NB_HIP(s, dalloc(&s->o_pb, 2 * n));
    NB_HIP(s, dalloc(&s->o_pe, 2 * n)); NB_HIP(s, dalloc(&s->o_parent, 2 * n));
    NB_HIP(s, dalloc(&s->o_size, 2 * n)); NB_HIP(s, dalloc(&s->nodeB, 2 * n));
    if (s->fp64) {
        NB_HIP(s, dalloc(&s->d.wsum, 2 * n)); NB_HIP(s, dalloc(&s->d.nodeA, 2 * n));
    } else {
        NB_HIP(s, dalloc(&s->o_wsum, 2 * n)); NB_HIP(s, dalloc(&s->nodeA, 2 * n));
    }
    const size_t ngroups = (n + s->group - 1) / s->group;
    NB_HIP(s, dalloc(&s->list_len, ngroups));
    NB_HIP(s, hipMemsetAsync(s->list_len, 0, std::max<size_t>(ngroups, 1) * 4, s->stream));
    if (!s->fp64) {
        TravTail t{};
        t.ax = s->ax; t.ay = s->ay; t.az = s->az;
        t.list_len = s->list_len;
        t.err = &s->counts->err;
        for (int i = 0; i < 2; ++i) {
            t.vhx[i] = s->vhx[i]; t.vhy[i] = s->vhy[i]; t.vhz[i] = s->vhz[i];
            t.vx[i] = s->vx[i]; t.vy[i] = s->vy[i]; t.vz[i] = s->vz[i];
        }
        NB_HIP(s, dalloc(&s->trav_tail, 1));
        NB_HIP(s, hipMemcpy(s->trav_tail, &t, sizeof t, hipMemcpyHostToDevice));
    }
    if (s->p.flags & NBODY_FLAG_FIT_KEYS) {
        const uint32_t init[12] = {0, 0, 0, 0, 0, 0x7f7fffffu, 0x7f7fffffu, 0x7f7fffffu, 0, 0, 0, 0};
        NB_HIP(s, dalloc(&s->keybox, 12));
        NB_HIP(s, hipMemcpy(s->keybox, init, sizeof init, hipMemcpyHostToDevice));
    }
    NB_HIP(s, dalloc(&s->let_boxes, (size_t) LET_REC * LET_MAX_BOXES));
    NB_HIP(s, dalloc(&s->let_counts, (size_t) LET_MAX_BOXES + 8));
    NB_HIP(s, dalloc(&s->let_mask, 2 * n));
    NB_HIP(s, dalloc(&s->let_seg, (size_t) LET_MAX_T + 2));
    NB_HIP(s, dalloc(&s->let_frontier, (size_t) LET_MAX_T * LET_REC));
    NB_HIP(s, hipEventCreateWithFlags(&s->ev_ready, hipEventDisableTiming));
    const size_t red_bytes = reduce_scratch_bytes(REDUCE_MAX_BLOCKS, 3);
    NB_HIP(s, hipMalloc((void **) &s->red_scratch, red_bytes));
    NB_HIP(s, hipMemsetAsync(s->red_scratch, 0, red_bytes, s->stream));
    NB_HIP(s, dalloc(&s->red_out, 8));
    NB_HIP(s, hipHostMalloc((void **) &s->red_host, 8 * sizeof(double)));
    NB_HIP(s, dalloc(&s->inter_dev, (size_t) TIMING_RING));
    NB_HIP(s, hipMemsetAsync(s->inter_dev, 0, TIMING_RING * 8, s->stream));
    if (s->p.num_steps_validator > 0 && s->fp64) {
        State64 &d = s->d;
        NB_HIP(s, dalloc(&d.tw_posm, n)); NB_HIP(s, dalloc(&d.tw_pot, n));
        NB_HIP(s, dalloc(&d.ds_pot, n));
        for (int k = 0; k < 3; ++k) {
            NB_HIP(s, dalloc(&d.tw_v[k], n)); NB_HIP(s, dalloc(&d.tw_vh[k], n));
            NB_HIP(s, dalloc(&d.tw_a[k], n)); NB_HIP(s, dalloc(&d.ds_a[k], n));
        }
    } else if (s->p.num_steps_validator > 0) {
        NB_HIP(s, dalloc(&s->tw_posm, n)); NB_HIP(s, dalloc(&s->tw_pot, n));
        NB_HIP(s, dalloc(&s->ds_pot, n));
        for (int k = 0; k < 3; ++k) {
            NB_HIP(s, dalloc(&s->tw_v[k], n)); NB_HIP(s, dalloc(&s->tw_vh[k], n));
            NB_HIP(s, dalloc(&s->tw_a[k], n)); NB_HIP(s, dalloc(&s->ds_a[k], n));
        }
    }
    if (s->p.flags & NBODY_FLAG_TIMING)
        for (int r = 0; r < TIMING_RING; ++r)
            for (int k = 0; k < EV_COUNT; ++k) NB_HIP(s, hipEventCreate(&s->ev[r][k]));
    return NBODY_OK;
}

// ---- counter-based RNG: every draw is a pure function of (seed, particle, draw index) ----
inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
inline double u01(uint64_t seed, uint64_t i, uint64_t k)
{
    const uint64_t h = splitmix64(splitmix64(seed ^ (i * 0xD1342543DE82EF95ull)) + k);
    return (double) (h >> 11) * (1.0 / 9007199254740992.0);  // [0,1)
}
inline double urange(uint64_t seed, uint64_t i, uint64_t k, double lo, double hi)
{
    return u01(seed, i, k) * (hi - lo) + lo;
}
// Box-Muller (include/cuda/initial_conditions.cuh:166-173); 1-u keeps log() finite.
inline double normal(uint64_t seed, uint64_t i, uint64_t k, double mean, double sd)
{
    const double r = std::sqrt(-2.0 * std::log(1.0 - u01(seed, i, k)));
    const double th = urange(seed, i, k + 1, 0.0, 2.0 * M_PI);
    return r * std::cos(th) * sd + mean;
}

// R = float or double: the samples are drawn in double and stored in the handle's precision.
// Particles [first, first + count) of a set of n_total: a particle's draws depend on its index
// in the whole set only, so ranks that sample disjoint index ranges produce the same set.
template <typename R>
int sample_ic(int kind, int count, uint64_t seed, float domain, float gravity, float p0, float p1,
              R *x, R *y, R *z, R *vx, R *vy, R *vz, R *m, long long first = 0, long long n_total = -1)
{
    const double c = domain * 0.5;
    const long long n = n_total < 0 ? count : n_total;
    auto clampd = [&](double v) { return (R) std::max(0.0, std::min((double) domain, v)); };
    for (int idx = 0; idx < count; ++idx) {
        const long long i = first + idx;
        double px = c, py = c, pz = c, ux = 0, uy = 0, uz = 0, mass = 1.0;
        switch (kind) {
        case 0:  // uniform cube of side 1 (initial_conditions.cuh:20-47, simulation.cu:170-182)
            px = c + urange(seed, i, 0, -0.5, 0.5);
            py = c + urange(seed, i, 1, -0.5, 0.5);
            pz = c + urange(seed, i, 2, -0.5, 0.5);
            break;
        case 1: {  // uniform sphere r = 0.5 (initial_conditions.cuh:49-80, simulation.cu:185-198)
            const double r = std::cbrt(u01(seed, i, 0)) * 0.5;
            const double th = std::acos(urange(seed, i, 1, -1.0, 1.0));
            const double ph = urange(seed, i, 2, 0.0, 2.0 * M_PI);
            px = c + r * std::sin(th) * std::cos(ph);
            py = c + r * std::sin(th) * std::sin(ph);
            pz = c + r * std::cos(th);
            break;
        }
        case 2: {  // Keplerian disk (initial_conditions.cuh:82-147, simulation.cu:200-220)
            const double star = 10.0, disk = 1.0, rmin = 0.03, rmax = 0.30, alpha = -1.5, h = 0.05;
            if (i == 0) { mass = star; break; }
            const double maxp = std::pow(rmax, alpha + 1), minp = std::pow(rmin, alpha + 1);
            const double r = std::pow(u01(seed, i, 0) * (maxp - minp) + minp, 1.0 / (alpha + 1));
            const double th = urange(seed, i, 1, 0.0, 2.0 * M_PI);
            px = c + r * std::cos(th);
            py = c + r * std::sin(th);
            pz = c + r * normal(seed, i, 2, 0.0, h);
            const double enc = star + disk * (std::pow(r, alpha + 1) - minp) / (maxp - minp);
            const double vk = std::sqrt(gravity * enc / r);
            ux = vk * std::sin(th);
            uy = -vk * std::cos(th);
            mass = disk / n;
            break;
        }
        case 3: {  // Plummer sphere, scale a = p0, truncated at p1, total mass 1, cold
            const double a = p0, rcut = p1;
            double r;
            uint64_t k = 0;
            do {
                const double u = std::max(u01(seed, i, 8 + k), 1e-12);
                r = a / std::sqrt(std::pow(u, -2.0 / 3.0) - 1.0);
                ++k;
            } while (r > rcut && k < 64);
            if (r > rcut) r = rcut;
            const double cz = urange(seed, i, 1, -1.0, 1.0);
            const double ph = urange(seed, i, 2, 0.0, 2.0 * M_PI);
            const double sxy = std::sqrt(std::max(0.0, 1.0 - cz * cz));
            px = c + r * sxy * std::cos(ph);
            py = c + r * sxy * std::sin(ph);
            pz = c + r * cz;
            mass = 1.0 / n;
            break;
        }
        default: return NBODY_ERR_INVALID;
        }
        x[idx] = clampd(px); y[idx] = clampd(py); z[idx] = clampd(pz);
        if (vx) vx[idx] = (R) ux;
        if (vy) vy[idx] = (R) uy;
        if (vz) vz[idx] = (R) uz;
        m[idx] = (R) mass;
    }
    return NBODY_OK;
}

int check_device_error(Sim *s)
{
    if (s->counts_host->err) {
        char buf[96];
        snprintf(buf, sizeof buf, "device-side error bits 0x%x (1=spin limit, 2=stack, 4=range)",
                 s->counts_host->err);
        return fail(s, NBODY_ERR_DEVICE, buf);
    }
    return NBODY_OK;
}

int fetch_counts(Sim *s)
{
    NB_HIP(s, hipMemcpyAsync(s->counts_host, s->counts, sizeof(Counts), hipMemcpyDeviceToHost,
                             s->stream));
    NB_HIP(s, hipStreamSynchronize(s->stream));
    return check_device_error(s);
}

// update_points + update_octree (src/cuda/simulation.cu:75-102) after the keys are encoded.
int build_after_encode(Sim *s, bool permute_all)
{
    // a particle set that has just arrived (upload, spawn, snapshot, migration) is in no order
    // that the previous sort's keys describe
    if (permute_all) s->old_keys_n = -1;
    NB_HIP(s, launch_sort(s));
    NB_HIP(s, s->fp64 ? launch64_permute(s, permute_all) : launch_permute(s, permute_all));
    return NBODY_OK;
}

// exec_post_spawn (src/cuda/simulation.cu:160-168): sort and build, accelerations stay 0.
int post_spawn(Sim *s)
{
    const size_t n = (size_t) s->n;
    // a fresh particle set starts with a clean device error word (the build rewrites the counts)
    NB_HIP(s, hipMemsetAsync(s->counts, 0, sizeof(Counts), s->stream));
    if (s->fp64) {
        NB_HIP(s, hipMemsetAsync(s->d.ax, 0, n * 8, s->stream));
        NB_HIP(s, hipMemsetAsync(s->d.ay, 0, n * 8, s->stream));
        NB_HIP(s, hipMemsetAsync(s->d.az, 0, n * 8, s->stream));
        NB_HIP(s, launch64_morton_only(s));
    } else {
        NB_HIP(s, hipMemsetAsync(s->ax, 0, n * 4, s->stream));
        NB_HIP(s, hipMemsetAsync(s->ay, 0, n * 4, s->stream));
        NB_HIP(s, hipMemsetAsync(s->az, 0, n * 4, s->stream));
        NB_HIP(s, launch_morton_only(s));
    }
    int rc = build_after_encode(s, true);
    if (rc) return rc;
    NB_HIP(s, launch_build(s));
    NB_HIP(s, s->fp64 ? launch64_com(s) : launch_com(s));
    NB_HIP(s, s->fp64 ? launch64_export_view(s) : launch_export_view(s));
    s->spawned = true;
    s->pending_vel = false;
    s->ev_pending = 0;
    s->t_acc = nbody_timings{};
    s->hist_e.clear(); s->hist_e_ap.clear(); s->hist_err.clear();
    s->val_count = 0;
    return fetch_counts(s);
}

// Host arrays in the handle's own precision R (St = Sim for float, State64 for double).
template <typename R, typename St, typename R4>
int upload_state_r(Sim *s, St &a, R4 *tw_posm, R *const tw_v[3], R *const tw_a[3], const R *x,
                   const R *y, const R *z, const R *vx, const R *vy, const R *vz, const R *m,
                   int count)
{
    s->n = count < 0 ? s->cap : count;
    const size_t n = (size_t) s->n, b = n * sizeof(R);
    const int c = s->cur;
    std::vector<uint32_t> ids(n);
    for (size_t i = 0; i < n; ++i) ids[i] = (uint32_t) i;
    {   // all masses equal (and a normal positive float)?
        bool uni = n > 0 && m[0] > R(1e-30) && m[0] < R(1e30);
        for (size_t i = 1; uni && i < n; ++i) uni = m[i] == m[0];
        s->uniform_mass = uni ? (float) m[0] : 0.f;
    }
    NB_HIP(s, hipMemcpyAsync(a.x[c], x, b, hipMemcpyHostToDevice, s->stream));
    NB_HIP(s, hipMemcpyAsync(a.y[c], y, b, hipMemcpyHostToDevice, s->stream));
    NB_HIP(s, hipMemcpyAsync(a.z[c], z, b, hipMemcpyHostToDevice, s->stream));
    NB_HIP(s, hipMemcpyAsync(a.m[c], m, b, hipMemcpyHostToDevice, s->stream));
    if (vx) NB_HIP(s, hipMemcpyAsync(a.vx[c], vx, b, hipMemcpyHostToDevice, s->stream));
    else NB_HIP(s, hipMemsetAsync(a.vx[c], 0, b, s->stream));
    if (vy) NB_HIP(s, hipMemcpyAsync(a.vy[c], vy, b, hipMemcpyHostToDevice, s->stream));
    else NB_HIP(s, hipMemsetAsync(a.vy[c], 0, b, s->stream));
    if (vz) NB_HIP(s, hipMemcpyAsync(a.vz[c], vz, b, hipMemcpyHostToDevice, s->stream));
    else NB_HIP(s, hipMemsetAsync(a.vz[c], 0, b, s->stream));
    NB_HIP(s, hipMemsetAsync(a.vhx[c], 0, b, s->stream));
    NB_HIP(s, hipMemsetAsync(a.vhy[c], 0, b, s->stream));
    NB_HIP(s, hipMemsetAsync(a.vhz[c], 0, b, s->stream));
    NB_HIP(s, hipMemcpyAsync(s->id[c], ids.data(), n * 4, hipMemcpyHostToDevice, s->stream));
    std::vector<R4> tw;
    if (s->p.num_steps_validator > 0) {
        // Validator::copy_initial_conditions (src/cuda/validator.cu:226-236): the all-pairs
        // twin starts from the same particles, zero accelerations; it keeps spawn order.
        tw.resize(n);
        for (size_t i = 0; i < n; ++i) tw[i] = R4{x[i], y[i], z[i], m[i]};
        NB_HIP(s, hipMemcpyAsync(tw_posm, tw.data(), n * sizeof(R4), hipMemcpyHostToDevice, s->stream));
        const R *vsrc[3] = {vx, vy, vz};
        for (int k = 0; k < 3; ++k) {
            if (vsrc[k]) NB_HIP(s, hipMemcpyAsync(tw_v[k], vsrc[k], b, hipMemcpyHostToDevice, s->stream));
            else NB_HIP(s, hipMemsetAsync(tw_v[k], 0, b, s->stream));
            NB_HIP(s, hipMemsetAsync(tw_a[k], 0, b, s->stream));
        }
        s->val_step = 0;
    }
    NB_HIP(s, hipStreamSynchronize(s->stream));  // host staging buffers go out of scope
    return post_spawn(s);
}

// Host arrays of type In into a handle of either precision (converted on the host when the
// two differ).
template <typename In>
int upload_state(Sim *s, const In *x, const In *y, const In *z, const In *vx, const In *vy,
                 const In *vz, const In *m, int count = -1)
{
    constexpr bool in64 = sizeof(In) == 8;
    if (in64 == s->fp64) {
        if constexpr (in64)
            return upload_state_r<double>(s, s->d, s->d.tw_posm, s->d.tw_v, s->d.tw_a, x, y, z, vx, vy, vz, m, count);
        else
            return upload_state_r<float>(s, *s, s->tw_posm, s->tw_v, s->tw_a, x, y, z, vx, vy, vz, m, count);
    }
    using Out = std::conditional_t<in64, float, double>;
    const size_t n = (size_t) (count < 0 ? s->cap : count);
    std::vector<Out> c[7];
    const In *src[7] = {x, y, z, vx, vy, vz, m};
    for (int k = 0; k < 7; ++k)
        if (src[k]) { c[k].resize(n); for (size_t i = 0; i < n; ++i) c[k][i] = (Out) src[k][i]; }
    auto P = [&](int k) -> const Out * { return src[k] ? c[k].data() : nullptr; };
    return upload_state<Out>(s, P(0), P(1), P(2), P(3), P(4), P(5), P(6), count);
}

int spawn(Sim *s, int kind, float p0, float p1)
{
    s->n = s->cap;
    const size_t n = (size_t) s->n;
    if (s->fp64) {
        std::vector<double> x(n), y(n), z(n), vx(n), vy(n), vz(n), m(n);
        int rc = sample_ic<double>(kind, s->n, s->p.seed, s->p.domain_size, s->p.gravity, p0, p1,
                                   x.data(), y.data(), z.data(), vx.data(), vy.data(), vz.data(), m.data());
        if (rc) return fail(s, rc, "unknown initial condition kind");
        return upload_state<double>(s, x.data(), y.data(), z.data(), vx.data(), vy.data(), vz.data(), m.data());
    }
    std::vector<float> x(n), y(n), z(n), vx(n), vy(n), vz(n), m(n);
    int rc = sample_ic<float>(kind, s->n, s->p.seed, s->p.domain_size, s->p.gravity, p0, p1, x.data(),
                              y.data(), z.data(), vx.data(), vy.data(), vz.data(), m.data());
    if (rc) return fail(s, rc, "unknown initial condition kind");
    return upload_state<float>(s, x.data(), y.data(), z.data(), vx.data(), vy.data(), vz.data(),
                               m.data());
}

// One validator record (reference: Validator<T>::update_all_pairs, src/cuda/validator.cu:238-320,
// metric definitions :19-63).  Differences, on purpose: the relative acceleration error is
// STEP-LOCAL (tree forces against a direct sum at the same positions; the reference compares
// against its separately evolving twin), and a record is taken after a step, not before it.
//   sys_energy    = sum_i m_i (v_i^2/2 - G/2 sum_{j!=i} m_j / sqrt(d^2+eps^2)) of the tree run
//   sys_energy_ap = the same for the all-pairs twin, advanced by the same leapfrog
//   avg_acc_err   = mean_i |a_tree - a_direct| / |a_direct|
int validator_step(Sim *s)
{
    const int n = s->n;
    const int c = s->cur;
    const float eps2 = s->p.softening_factor * s->p.softening_factor;
    const float G = s->p.gravity;
    NB_HIP(s, launch_direct_pot(s->posm, n, s->posm, n, eps2, G, s->ds_a[0], s->ds_a[1], s->ds_a[2],
                                s->ds_pot, 1, s->stream));
    NB_HIP(s, launch_twin_kick_drift(n, s->p.dt, s->p.domain_size, s->tw_posm, s->tw_v[0],
                                     s->tw_v[1], s->tw_v[2], s->tw_a[0], s->tw_a[1], s->tw_a[2],
                                     s->tw_vh[0], s->tw_vh[1], s->tw_vh[2], s->stream));
    NB_HIP(s, launch_direct_pot(s->tw_posm, n, s->tw_posm, n, eps2, G, s->tw_a[0], s->tw_a[1],
                                s->tw_a[2], s->tw_pot, 1, s->stream));
    NB_HIP(s, launch_vel_kick_raw(n, s->p.dt, 1.0f - s->p.velocity_dampening, s->tw_vh[0],
                                  s->tw_vh[1], s->tw_vh[2], s->tw_a[0], s->tw_a[1], s->tw_a[2],
                                  s->tw_v[0], s->tw_v[1], s->tw_v[2], s->stream));
    // the three sums are reduced on the device (reference: cub::DeviceReduce::Sum,
    // src/cuda/validator.cu:261-287); six doubles cross PCIe per record
    const float *acc[3] = {s->ax, s->ay, s->az};
    NB_HIP(s, launch_energy_reduce(n, s->m[c], 1, s->vx[c], s->vy[c], s->vz[c], s->ds_pot, (double) G,
                                   acc, s->ds_a, s->red_scratch, s->red_out, s->stream));
    NB_HIP(s, launch_energy_reduce(n, &s->tw_posm[0].w, 4, s->tw_v[0], s->tw_v[1], s->tw_v[2], s->tw_pot,
                                   (double) G, nullptr, nullptr, s->red_scratch, s->red_out + 3, s->stream));
    NB_HIP(s, hipMemcpyAsync(s->red_host, s->red_out, 6 * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    NB_HIP(s, hipStreamSynchronize(s->stream));
    const double e = s->red_host[0] + s->red_host[1], e_ap = s->red_host[3] + s->red_host[4];
    const double err = s->red_host[2];
    const size_t cap = (size_t) s->p.num_steps_validator;
    if (s->hist_e.size() < cap) { s->hist_e.resize(cap); s->hist_e_ap.resize(cap); s->hist_err.resize(cap); }
    s->hist_e[s->val_step] = e;
    s->hist_e_ap[s->val_step] = e_ap;
    s->hist_err[s->val_step] = err / n;
    s->val_step = (s->val_step + 1) % s->p.num_steps_validator;  // wraps like the reference
    if (s->val_count < (int) cap) s->val_count++;
    return NBODY_OK;
}

// The same record for a double-precision handle: tree run, twin and direct sum all in double
// (Validator<double>, src/cuda/validator.cu:381).
int validator_step64(Sim *s)
{
    State64 &d = s->d;
    const int n = s->n;
    const int c = s->cur;
    const double eps = (double) s->p.softening_factor, eps2 = eps * eps;
    const double G = (double) s->p.gravity, dt = (double) s->p.dt;
    NB_HIP(s, launch64_direct(d.posm, n, d.posm, n, eps2, G, d.ds_a[0], d.ds_a[1], d.ds_a[2], d.ds_pot, 1, s->stream));
    NB_HIP(s, launch64_twin_kick_drift(n, dt, (double) s->p.domain_size, d.tw_posm, d.tw_v, d.tw_a, d.tw_vh, s->stream));
    NB_HIP(s, launch64_direct(d.tw_posm, n, d.tw_posm, n, eps2, G, d.tw_a[0], d.tw_a[1], d.tw_a[2], d.tw_pot, 1, s->stream));
    NB_HIP(s, launch64_vel_kick_raw(n, dt, 1.0 - (double) s->p.velocity_dampening, d.tw_vh, d.tw_a, d.tw_v, s->stream));
    const double *acc[3] = {d.ax, d.ay, d.az};
    NB_HIP(s, launch_energy_reduce64(n, d.m[c], 1, d.vx[c], d.vy[c], d.vz[c], d.ds_pot, G, acc, d.ds_a,
                                     s->red_scratch, s->red_out, s->stream));
    NB_HIP(s, launch_energy_reduce64(n, &d.tw_posm[0].w, 4, d.tw_v[0], d.tw_v[1], d.tw_v[2], d.tw_pot, G,
                                     nullptr, nullptr, s->red_scratch, s->red_out + 3, s->stream));
    NB_HIP(s, hipMemcpyAsync(s->red_host, s->red_out, 6 * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    NB_HIP(s, hipStreamSynchronize(s->stream));
    const double e = s->red_host[0] + s->red_host[1], e_ap = s->red_host[3] + s->red_host[4];
    const double err = s->red_host[2];
    const size_t cap = (size_t) s->p.num_steps_validator;
    if (s->hist_e.size() < cap) { s->hist_e.resize(cap); s->hist_e_ap.resize(cap); s->hist_err.resize(cap); }
    s->hist_e[s->val_step] = e;
    s->hist_e_ap[s->val_step] = e_ap;
    s->hist_err[s->val_step] = err / n;
    s->val_step = (s->val_step + 1) % s->p.num_steps_validator;
    if (s->val_count < (int) cap) s->val_count++;
    return NBODY_OK;
}

// upload_state + accelerations and ids carried through the sort it performs (identity when
// the particles arrive in the sorted order of a previous step).
template <typename R>
int set_particles(Sim *s, int count, const R *x, const R *y, const R *z, const R *vx, const R *vy,
                  const R *vz, const R *m, const R *ax, const R *ay, const R *az, const uint32_t *id)
{
    int rc = upload_state<R>(s, x, y, z, vx, vy, vz, m, count);
    if (rc) return rc;
    if (!ax && !ay && !az && !id) return NBODY_OK;
    const size_t n = (size_t) count;
    std::vector<uint32_t> perm(n), nid(n);
    NB_HIP(s, hipMemcpy(perm.data(), s->id[s->cur], n * 4, hipMemcpyDeviceToHost));
    const R *src[3] = {ax, ay, az};
    for (int k = 0; k < 3; ++k) {
        if (!src[k]) continue;
        if (s->fp64) {
            std::vector<double> a(n);
            double *dst[3] = {s->d.ax, s->d.ay, s->d.az};
            for (size_t i = 0; i < n; ++i) a[i] = (double) src[k][perm[i]];
            NB_HIP(s, hipMemcpy(dst[k], a.data(), n * 8, hipMemcpyHostToDevice));
        } else {
            std::vector<float> a(n);
            float *dst[3] = {s->ax, s->ay, s->az};
            for (size_t i = 0; i < n; ++i) a[i] = (float) src[k][perm[i]];
            NB_HIP(s, hipMemcpy(dst[k], a.data(), n * 4, hipMemcpyHostToDevice));
        }
    }
    if (id) {
        for (size_t i = 0; i < n; ++i) nid[i] = id[perm[i]];
        NB_HIP(s, hipMemcpy(s->id[s->cur], nid.data(), n * 4, hipMemcpyHostToDevice));
    }
    return NBODY_OK;
}

}  // namespace

namespace nb {

int install_particles_dev(Sim *s, int n)
{
    if (n < 1 || n > s->cap) return fail(s, NBODY_ERR_CAPACITY, "install_particles: count outside [1, num_points]");
    s->n = n;
    s->uniform_mass = 0.f;   // particles of other ranks: their masses are not known here
    NB_HIP(s, hipMemsetAsync(s->counts, 0, sizeof(Counts), s->stream));
    NB_HIP(s, s->fp64 ? launch64_morton_only(s) : launch_morton_only(s));
    int rc = build_after_encode(s, true);   // permutes x,y,z,m,v,vh,id: the accelerations ride in vh
    if (rc) return rc;
    const int c = s->cur;
    if (s->fp64) {
        const size_t b = (size_t) n * sizeof(double);
        NB_HIP(s, hipMemcpyAsync(s->d.ax, s->d.vhx[c], b, hipMemcpyDeviceToDevice, s->stream));
        NB_HIP(s, hipMemcpyAsync(s->d.ay, s->d.vhy[c], b, hipMemcpyDeviceToDevice, s->stream));
        NB_HIP(s, hipMemcpyAsync(s->d.az, s->d.vhz[c], b, hipMemcpyDeviceToDevice, s->stream));
    } else {
        const size_t b = (size_t) n * sizeof(float);
        NB_HIP(s, hipMemcpyAsync(s->ax, s->vhx[c], b, hipMemcpyDeviceToDevice, s->stream));
        NB_HIP(s, hipMemcpyAsync(s->ay, s->vhy[c], b, hipMemcpyDeviceToDevice, s->stream));
        NB_HIP(s, hipMemcpyAsync(s->az, s->vhz[c], b, hipMemcpyDeviceToDevice, s->stream));
    }
    NB_HIP(s, launch_build(s));
    NB_HIP(s, s->fp64 ? launch64_com(s) : launch_com(s));
    NB_HIP(s, s->fp64 ? launch64_export_view(s) : launch_export_view(s));
    s->spawned = true;
    s->pending_vel = false;
    s->ev_pending = 0;
    s->t_acc = nbody_timings{};
    return NBODY_OK;
}

// Particles [first, first + count) of a sampled set of n_total, drawn in the handle's precision,
// ids = index in the whole set.
template <typename R>
static int spawn_range_r(Sim *s, int kind, long long first, int count, long long n_total, float p0, float p1)
{
    const size_t n = (size_t) count;
    std::vector<R> x(n), y(n), z(n), vx(n), vy(n), vz(n), m(n);
    std::vector<uint32_t> id(n);
    int rc = sample_ic<R>(kind, count, s->p.seed, s->p.domain_size, s->p.gravity, p0, p1, x.data(), y.data(),
                          z.data(), vx.data(), vy.data(), vz.data(), m.data(), first, n_total);
    if (rc) return rc;
    for (size_t i = 0; i < n; ++i) id[i] = (uint32_t) (first + (long long) i);
    if (s->p.num_steps_validator > 0 && count != s->cap)
        return fail(s, NBODY_ERR_STATE, "a handle with a validator keeps num_points particles");
    return set_particles<R>(s, count, x.data(), y.data(), z.data(), vx.data(), vy.data(), vz.data(), m.data(),
                            (const R *) nullptr, (const R *) nullptr, (const R *) nullptr, id.data());
}

int spawn_range(Sim *s, int kind, long long first, int count, long long n_total, float p0, float p1)
{
    if (count < 1 || count > s->cap) return fail(s, NBODY_ERR_CAPACITY, "spawn_range: count outside [1, num_points]");
    return s->fp64 ? spawn_range_r<double>(s, kind, first, count, n_total, p0, p1)
                   : spawn_range_r<float>(s, kind, first, count, n_total, p0, p1);
}

int sample_ic_range(int kind, long long first, int count, long long n_total, uint64_t seed, float domain,
                    float gravity, float p0, float p1, float *x, float *y, float *z, float *vx,
                    float *vy, float *vz, float *m)
{
    return sample_ic<float>(kind, count, seed, domain, gravity, p0, p1, x, y, z, vx, vy, vz, m, first, n_total);
}

}  // namespace nb

namespace {

// Device array of the handle's precision Src copied out to a host array of type Dst.
template <typename Dst, typename Src>
hipError_t copy_out(Sim *s, Dst *dst, const Src *src, size_t n)
{
    if (!dst) return hipSuccess;
    if constexpr (std::is_same<Dst, Src>::value) {
        return hipMemcpyAsync(dst, src, n * sizeof(Src), hipMemcpyDeviceToHost, s->stream);
    } else {
        std::vector<Src> tmp(n);
        hipError_t e = hipMemcpyAsync(tmp.data(), src, n * sizeof(Src), hipMemcpyDeviceToHost, s->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
        for (size_t i = 0; i < n; ++i) dst[i] = (Dst) tmp[i];
        return e;
    }
}

template <typename Dst>
int get_state(Sim *s, Dst *x, Dst *y, Dst *z, Dst *vx, Dst *vy, Dst *vz, Dst *m, Dst *ax, Dst *ay,
              Dst *az, uint32_t *id)
{
    const size_t n = (size_t) s->n;
    const int c = s->cur;
    if (s->fp64) {
        const State64 &d = s->d;
        Dst *dst[10] = {x, y, z, vx, vy, vz, m, ax, ay, az};
        const double *src[10] = {d.x[c], d.y[c], d.z[c], d.vx[c], d.vy[c], d.vz[c], d.m[c], d.ax, d.ay, d.az};
        for (int k = 0; k < 10; ++k) NB_HIP(s, copy_out(s, dst[k], src[k], n));
    } else {
        Dst *dst[10] = {x, y, z, vx, vy, vz, m, ax, ay, az};
        const float *src[10] = {s->x[c], s->y[c], s->z[c], s->vx[c], s->vy[c], s->vz[c], s->m[c], s->ax, s->ay, s->az};
        for (int k = 0; k < 10; ++k) NB_HIP(s, copy_out(s, dst[k], src[k], n));
    }
    NB_HIP(s, copy_out(s, id, s->id[c], n));
    return NBODY_OK;
}

// Centres of mass and masses of the first `no` octree nodes as host arrays of type Dst.
template <typename Dst>
int get_com(Sim *s, size_t no, Dst *comx, Dst *comy, Dst *comz, Dst *mass)
{
    auto spread = [&](auto &a) {
        for (size_t i = 0; i < no; ++i) {
            if (comx) comx[i] = (Dst) a[i].x;
            if (comy) comy[i] = (Dst) a[i].y;
            if (comz) comz[i] = (Dst) a[i].z;
            if (mass) mass[i] = (Dst) a[i].w;
        }
    };
    if (s->fp64) {
        std::vector<Double4> a(no);
        NB_HIP(s, hipMemcpy(a.data(), s->d.nodeA, no * sizeof(Double4), hipMemcpyDeviceToHost));
        spread(a);
    } else {
        std::vector<Float4> a(no);
        NB_HIP(s, hipMemcpy(a.data(), s->nodeA, no * sizeof(Float4), hipMemcpyDeviceToHost));
        spread(a);
    }
    return NBODY_OK;
}

#define NB_FLOAT_ONLY(s, what)                                                                 \
    if ((s)->fp64) return fail(s, NBODY_ERR_STATE, what ": not available on a double-precision handle")

// Folds the oldest pending timing record into the host accumulators and frees its ring slot.
// wait: the record's step may still be running (the ring is full and nbody_update needs the
// slot): wait for it.  The interaction counter of a step is written by a kernel queued after
// the step's last event and before the next step's first one, so that next event is the one
// to wait for; the caller guarantees it exists (ring full) or has synchronised the stream.
int fold_oldest_timing(Sim *s, bool wait)
{
    const int slot = ((s->ev_head - s->ev_pending) % TIMING_RING + TIMING_RING) % TIMING_RING;
    if (wait) NB_HIP(s, hipEventSynchronize(s->ev[(slot + 1) % TIMING_RING][EV_BEGIN]));
    float t[EV_COUNT] = {0};
    if (s->p.flags & NBODY_FLAG_TIMING_STAGES) {
        for (int k = 1; k <= EV_TRAVERSE; ++k)
            NB_HIP(s, hipEventElapsedTime(&t[k], s->ev[slot][k - 1], s->ev[slot][k]));
    } else {   // coarse: everything before the traversal is reported as "build"
        NB_HIP(s, hipEventElapsedTime(&t[EV_BUILD], s->ev[slot][EV_BEGIN], s->ev[slot][EV_COM]));
        NB_HIP(s, hipEventElapsedTime(&t[EV_TRAVERSE], s->ev[slot][EV_COM], s->ev[slot][EV_TRAVERSE]));
    }
    float tot = 0;
    NB_HIP(s, hipEventElapsedTime(&tot, s->ev[slot][EV_BEGIN], s->ev[slot][EV_TRAVERSE]));
    for (int k = 0; k < s->tf_calls[slot]; ++k) {   // traversals driven through nbody_traverse_from
        float tf = 0;
        NB_HIP(s, hipEventElapsedTime(&tf, s->ev[slot][EV_TF0_BEGIN + 2 * k], s->ev[slot][EV_TF0_END + 2 * k]));
        t[EV_TRAVERSE] += tf;
        tot += tf;
    }
    uint64_t inter = 0;
    NB_HIP(s, hipMemcpy(&inter, s->inter_dev + slot, sizeof inter, hipMemcpyDeviceToHost));
    nbody_timings &a = s->t_acc;
    a.steps += 1;
    a.ms_integrate += t[EV_INTEGRATE];
    a.ms_sort += t[EV_SORT];
    a.ms_build += t[EV_BUILD];
    a.ms_com += t[EV_COM];
    a.ms_traverse += t[EV_TRAVERSE];
    a.ms_total += tot;
    a.interactions += inter;
    s->ev_pending--;
    return NBODY_OK;
}

struct DevGuard {
    int prev = -1;
    explicit DevGuard(int dev) { (void) hipGetDevice(&prev); if (prev != dev) (void) hipSetDevice(dev); }
    ~DevGuard() { if (prev >= 0) (void) hipSetDevice(prev); }
};

}  // namespace

extern "C" {

void nbody_default_params(nbody_params *p)
{
    std::memset(p, 0, sizeof *p);
    p->num_points = 1 << 18;
    p->max_num_codes_per_leaf = 32;
    p->traversal_group_size = 64;
    p->mem_traversal_queues = (size_t) (1 << 20) * 50;
    p->theta = 0.75f;
    p->dt = 1e-4f;
    p->gravity = 1.0f;
    p->softening_factor = 0.001f;
    p->domain_size = 5.0f;
    p->velocity_dampening = 0.0f;
    p->num_steps_validator = 0;
    p->morton_bits = 10;
    p->device_id = -1;
    p->flags = 0;
    p->seed = 42;
}

int nbody_create_external(const nbody_params *params, void *dev_ptrs[7], nbody_t **out)
{
    const int rc = nbody_create(params, out);
    if (rc != NBODY_OK || !dev_ptrs) return rc;
    Sim *s = reinterpret_cast<Sim *>(*out);
    for (int i = 0; i < 7; ++i) {
        if (s->fp64) s->d.ext[i] = static_cast<double *>(dev_ptrs[i]);
        else s->ext[i] = static_cast<float *>(dev_ptrs[i]);
        s->has_ext = s->has_ext || dev_ptrs[i] != nullptr;
    }
    return NBODY_OK;
}

int nbody_create(const nbody_params *params, nbody_t **out)
{
    if (!params || !out) return NBODY_ERR_INVALID;
    *out = nullptr;
    const nbody_params &p = *params;
    if (p.num_points < 1 || p.num_points >= (1 << 29)) return NBODY_ERR_INVALID;
    if (p.morton_bits != 10 && p.morton_bits != 21) return NBODY_ERR_INVALID;
    if (p.traversal_group_size != 64 && p.traversal_group_size != 32) return NBODY_ERR_INVALID;
    if (p.max_num_codes_per_leaf < 1 || !(p.domain_size > 0.0f) || !(p.theta >= 0.0f) ||
        !(p.softening_factor > 0.0f))
        return NBODY_ERR_INVALID;
    if ((p.flags & NBODY_FLAG_FIT_KEYS) && (p.flags & NBODY_FLAG_FP64)) return NBODY_ERR_INVALID;
    Sim *s = new (std::nothrow) Sim();
    if (!s) return NBODY_ERR_INVALID;
    s->p = p;
    s->n = p.num_points;
    s->cap = p.num_points;
    s->group = p.traversal_group_size;
    s->key_bytes = p.morton_bits == 10 ? 4 : 8;
    s->fp64 = (p.flags & NBODY_FLAG_FP64) != 0;
    int dev = p.device_id;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) { delete s; return NBODY_ERR_HIP; }
    s->device = dev;
    DevGuard guard(dev);
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) {
        delete s;
        return NBODY_ERR_HIP;
    }
    int rc = alloc_all(s);
    if (rc == NBODY_OK && hipStreamSynchronize(s->stream) != hipSuccess) rc = NBODY_ERR_HIP;
    if (rc != NBODY_OK) {
        fprintf(stderr, "nbody_create: %s\n", s->err.c_str());
        free_all(s);
        delete s;
        return rc;
    }
    noalloc_prepare_current_device();
    *out = reinterpret_cast<nbody_t *>(s);
    return NBODY_OK;
}

void nbody_destroy(nbody_t *h)
{
    if (!h) return;
    Sim *s = reinterpret_cast<Sim *>(h);
    DevGuard guard(s->device);
    if (s->stream) (void) hipStreamSynchronize(s->stream);
    free_all(s);
    delete s;
}

const char *nbody_last_error(nbody_t *h)
{
    return h ? reinterpret_cast<Sim *>(h)->err.c_str() : "null handle";
}

int nbody_sample_ic(int kind, int n, uint64_t seed, float domain_size, float gravity, float p0,
                    float p1, float *x, float *y, float *z, float *vx, float *vy, float *vz,
                    float *m)
{
    if (n < 1 || !x || !y || !z || !m) return NBODY_ERR_INVALID;
    return sample_ic<float>(kind, n, seed, domain_size, gravity, p0, p1, x, y, z, vx, vy, vz, m);
}

int nbody_sample_ic_f64(int kind, int n, uint64_t seed, float domain_size, float gravity, float p0,
                        float p1, double *x, double *y, double *z, double *vx, double *vy,
                        double *vz, double *m)
{
    if (n < 1 || !x || !y || !z || !m) return NBODY_ERR_INVALID;
    return sample_ic<double>(kind, n, seed, domain_size, gravity, p0, p1, x, y, z, vx, vy, vz, m);
}

#define NB_ENTER(h)                                  \
    if (!(h)) return NBODY_ERR_INVALID;              \
    Sim *s = reinterpret_cast<Sim *>(h);             \
    DevGuard guard_(s->device)

int nbody_spawn_uniform(nbody_t *h) { NB_ENTER(h); return spawn(s, 0, 0, 0); }
int nbody_spawn_sphere(nbody_t *h) { NB_ENTER(h); return spawn(s, 1, 0, 0); }
int nbody_spawn_disk(nbody_t *h) { NB_ENTER(h); return spawn(s, 2, 0, 0); }
int nbody_spawn_plummer(nbody_t *h, float a, float r_max)
{
    NB_ENTER(h);
    if (!(a > 0.0f) || !(r_max > 0.0f)) return fail(s, NBODY_ERR_INVALID, "plummer: a, r_max > 0");
    return spawn(s, 3, a, r_max);
}

int nbody_set_state(nbody_t *h, const float *x, const float *y, const float *z, const float *vx,
                    const float *vy, const float *vz, const float *m)
{
    NB_ENTER(h);
    if (!x || !y || !z || !m) return fail(s, NBODY_ERR_INVALID, "set_state: x, y, z, m required");
    return upload_state<float>(s, x, y, z, vx, vy, vz, m);
}

int nbody_set_state_f64(nbody_t *h, const double *x, const double *y, const double *z,
                        const double *vx, const double *vy, const double *vz, const double *m)
{
    NB_ENTER(h);
    if (!x || !y || !z || !m) return fail(s, NBODY_ERR_INVALID, "set_state: x, y, z, m required");
    return upload_state<double>(s, x, y, z, vx, vy, vz, m);
}

int nbody_update(nbody_t *h)
{
    NB_ENTER(h);
    if (!s->spawned) return fail(s, NBODY_ERR_STATE, "update before spawn/set_state");
    if (s->pending_vel) return fail(s, NBODY_ERR_STATE, "nbody_finish_step not called");
    s->imp_nroots = 0;   // copies of other trees belong to the step that imported them
    const bool timing = (s->p.flags & NBODY_FLAG_TIMING) != 0;
    const int slot = s->ev_head;
    if (timing && s->ev_pending == TIMING_RING) {
        // ring full: the record of the step TIMING_RING steps back moves into the host
        // accumulators (that step finished long ago unless the caller queues faster than the
        // GPU runs, in which case this waits for it)
        int rcf = fold_oldest_timing(s, true);
        if (rcf) return rcf;
    }
    // every recorded event costs a few microseconds of stream time: without
    // NBODY_FLAG_TIMING_STAGES only the step's begin and the traversal's two ends are marked
    const bool stages = (s->p.flags & NBODY_FLAG_TIMING_STAGES) != 0;
    auto mark = [&](int k) -> hipError_t {
        const bool wanted = timing && (stages || k == EV_BEGIN || k == EV_COM || k == EV_TRAVERSE);
        return wanted ? hipEventRecord(s->ev[slot][k], s->stream) : hipSuccess;
    };
    const bool sharded = s->trav_g1 >= 0;
    const bool fuse = !sharded && !(s->p.flags & NBODY_FLAG_NO_FUSED_VEL);
    NB_HIP(s, mark(EV_BEGIN));
    NB_HIP(s, s->fp64 ? launch64_integrate_morton(s) : launch_integrate_morton(s));
    NB_HIP(s, mark(EV_INTEGRATE));
    int rc = build_after_encode(s, false);
    if (rc) return rc;
    NB_HIP(s, mark(EV_SORT));
    NB_HIP(s, launch_build(s));
    NB_HIP(s, mark(EV_BUILD));
    NB_HIP(s, s->fp64 ? launch64_com(s) : launch_com(s));
    NB_HIP(s, mark(EV_COM));
    if (s->fp64) {
        NB_HIP(s, launch64_export_view(s));
        NB_HIP(s, launch64_traverse(s, false));
        if (!sharded) NB_HIP(s, launch64_vel_kick(s));
    } else {
        NB_HIP(s, launch_export_view(s));
        NB_HIP(s, launch_traverse(s, fuse));
        if (!fuse && !sharded) NB_HIP(s, launch_vel_kick(s));
    }
    NB_HIP(s, mark(EV_TRAVERSE));
    if (timing) {
        NB_HIP(s, launch_count_interactions(s, (unsigned long long *) (s->inter_dev + slot)));
        s->tf_calls[slot] = 0;
        s->ev_head = (slot + 1) % TIMING_RING;
        s->ev_pending++;
    }
    if (sharded) s->pending_vel = true;
    if (s->p.num_steps_validator > 0 && !sharded) return s->fp64 ? validator_step64(s) : validator_step(s);
    return NBODY_OK;
}

int nbody_finish_step(nbody_t *h)
{
    NB_ENTER(h);
    if (!s->pending_vel) return fail(s, NBODY_ERR_STATE, "no sharded step pending");
    NB_HIP(s, s->fp64 ? launch64_vel_kick(s) : launch_vel_kick(s));
    s->pending_vel = false;
    return NBODY_OK;
}

int nbody_set_traversal_range(nbody_t *h, int32_t g0, int32_t g1)
{
    NB_ENTER(h);
    const int ngroups = (s->n + s->group - 1) / s->group;
    if (g1 >= 0 && (g0 < 0 || g0 > g1 || g1 > ngroups))
        return fail(s, NBODY_ERR_INVALID, "traversal range outside [0, ngroups]");
    s->trav_g0 = g0;
    s->trav_g1 = g1;
    return NBODY_OK;
}

int nbody_step(nbody_t *h, int n)
{
    for (int i = 0; i < n; ++i) {
        int rc = nbody_update(h);
        if (rc) return rc;
    }
    return NBODY_OK;
}

int nbody_sync(nbody_t *h)
{
    NB_ENTER(h);
    return fetch_counts(s);
}

int nbody_get_num_octree_nodes(nbody_t *h, int32_t *out)
{
    NB_ENTER(h);
    if (!out) return fail(s, NBODY_ERR_INVALID, "null output");
    int rc = fetch_counts(s);
    if (rc) return rc;
    *out = s->counts_host->n_oct;
    return NBODY_OK;
}

int nbody_get_key_domain(nbody_t *h, float *out)
{
    NB_ENTER(h);
    if (!out) return fail(s, NBODY_ERR_INVALID, "null output");
    out[0] = out[1] = out[2] = 0.f;
    out[3] = s->p.domain_size;
    if (!s->keybox) return NBODY_OK;
    NB_HIP(s, hipMemcpyAsync(out, s->keybox, 4 * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    NB_HIP(s, hipStreamSynchronize(s->stream));
    return NBODY_OK;
}

int nbody_get_sort_stats(nbody_t *h, uint64_t *coherent_sorts, uint64_t *full_sorts, int32_t *last_moved)
{
    NB_ENTER(h);
    NB_HIP(s, hipStreamSynchronize(s->stream));
    if (coherent_sorts) *coherent_sorts = s->coh_sorts;
    if (full_sorts) *full_sorts = s->full_sorts;
    if (last_moved) *last_moved = s->coh_host ? s->coh_host[0] : -1;
    return NBODY_OK;
}

int nbody_get_tree_counts(nbody_t *h, int32_t *U, int32_t *M, int32_t *n_oct)
{
    NB_ENTER(h);
    int rc = fetch_counts(s);
    if (rc) return rc;
    if (U) *U = s->counts_host->U;
    if (M) *M = s->counts_host->M;
    if (n_oct) *n_oct = s->counts_host->n_oct;
    return NBODY_OK;
}

int nbody_get_state(nbody_t *h, float *x, float *y, float *z, float *vx, float *vy, float *vz,
                    float *m, float *ax, float *ay, float *az, uint32_t *id)
{
    NB_ENTER(h);
    int rc = get_state<float>(s, x, y, z, vx, vy, vz, m, ax, ay, az, id);
    return rc ? rc : fetch_counts(s);
}

int nbody_get_state_f64(nbody_t *h, double *x, double *y, double *z, double *vx, double *vy,
                        double *vz, double *m, double *ax, double *ay, double *az, uint32_t *id)
{
    NB_ENTER(h);
    int rc = get_state<double>(s, x, y, z, vx, vy, vz, m, ax, ay, az, id);
    return rc ? rc : fetch_counts(s);
}

int nbody_get_device_ptrs(nbody_t *h, nbody_dev_view *out)
{
    NB_ENTER(h);
    if (!out) return fail(s, NBODY_ERR_INVALID, "null output");
    const int c = s->cur;
    std::memset(out, 0, sizeof *out);
    out->n = s->n;
    out->id = s->id[c];
    out->stream = (void *) s->stream;
    if (s->fp64) return NBODY_OK;   // no float arrays: nbody_get_device_ptrs_f64 has the doubles
    out->x = s->x[c]; out->y = s->y[c]; out->z = s->z[c];
    out->vx = s->vx[c]; out->vy = s->vy[c]; out->vz = s->vz[c];
    out->ax = s->ax; out->ay = s->ay; out->az = s->az;
    out->m = s->m[c];
    out->posm = reinterpret_cast<const float *>(s->posm);
    return NBODY_OK;
}

int nbody_get_device_ptrs_f64(nbody_t *h, nbody_dev_view_f64 *out)
{
    NB_ENTER(h);
    if (!out) return fail(s, NBODY_ERR_INVALID, "null output");
    const int c = s->cur;
    std::memset(out, 0, sizeof *out);
    out->n = s->n;
    out->id = s->id[c];
    out->stream = (void *) s->stream;
    if (!s->fp64) return NBODY_OK;
    const State64 &d = s->d;
    out->x = d.x[c]; out->y = d.y[c]; out->z = d.z[c];
    out->vx = d.vx[c]; out->vy = d.vy[c]; out->vz = d.vz[c];
    out->ax = d.ax; out->ay = d.ay; out->az = d.az;
    out->m = d.m[c];
    out->posm = reinterpret_cast<const double *>(d.posm);
    return NBODY_OK;
}

int nbody_get_octree(nbody_t *h, nbody_octree_view *v)
{
    NB_ENTER(h);
    if (!v) return fail(s, NBODY_ERR_INVALID, "null output");
    int rc = fetch_counts(s);
    if (rc) return rc;
    const size_t n = (size_t) s->n;
    const size_t U = (size_t) s->counts_host->U, M = (size_t) s->counts_host->M;
    const size_t no = (size_t) s->counts_host->n_oct;
    auto D2H = [&](void *dst, const void *src, size_t bytes) -> hipError_t {
        return dst ? hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) : hipSuccess;
    };
    auto keys64 = [&](uint64_t *dst, const void *src, size_t cnt) -> hipError_t {
        if (!dst) return hipSuccess;
        if (s->key_bytes == 8) return hipMemcpy(dst, src, cnt * 8, hipMemcpyDeviceToHost);
        std::vector<uint32_t> tmp(cnt);
        hipError_t e = hipMemcpy(tmp.data(), src, cnt * 4, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < cnt; ++i) dst[i] = tmp[i];
        return e;
    };
    NB_HIP(s, keys64(v->codes_sorted, s->keys[s->sorted_buf], n));
    NB_HIP(s, D2H(v->perm, s->vals[s->sorted_buf], n * 4));
    NB_HIP(s, keys64(v->unique, s->unique, U));
    NB_HIP(s, D2H(v->first_pt, s->first_pt, (U + 1) * 4));
    NB_HIP(s, D2H(v->leaf_first_code_idx, s->lfc, (M + 1) * 4));
    NB_HIP(s, D2H(v->first_child, s->o_first_child, no * 4));
    NB_HIP(s, D2H(v->num_children, s->o_num_children, no * 4));
    NB_HIP(s, D2H(v->leaves_begin, s->o_lb, no * 4));
    NB_HIP(s, D2H(v->leaves_end, s->o_le, no * 4));
    NB_HIP(s, D2H(v->depth, s->o_depth, no * 4));
    NB_HIP(s, D2H(v->level, s->o_level, no * 4));
    NB_HIP(s, D2H(v->size, s->o_size, no * 4));
    NB_HIP(s, D2H(v->points_begin, s->o_pb, no * 4));
    NB_HIP(s, D2H(v->points_end, s->o_pe, no * 4));
    NB_HIP(s, D2H(v->parent, s->o_parent, no * 4));
    if (v->comx || v->comy || v->comz || v->mass) {
        int rc2 = get_com<float>(s, no, v->comx, v->comy, v->comz, v->mass);
        if (rc2) return rc2;
    }
    const size_t ngroups = (n + s->group - 1) / s->group;
    NB_HIP(s, D2H(v->list_len, s->list_len, ngroups * 4));
    return NBODY_OK;
}

int nbody_get_octree_com_f64(nbody_t *h, double *comx, double *comy, double *comz, double *mass)
{
    NB_ENTER(h);
    int rc = fetch_counts(s);
    if (rc) return rc;
    return get_com<double>(s, (size_t) s->counts_host->n_oct, comx, comy, comz, mass);
}

int nbody_get_timings(nbody_t *h, nbody_timings *out)
{
    NB_ENTER(h);
    if (!out) return fail(s, NBODY_ERR_INVALID, "null output");
    std::memset(out, 0, sizeof *out);
    if (!(s->p.flags & NBODY_FLAG_TIMING)) return fail(s, NBODY_ERR_STATE, "NBODY_FLAG_TIMING not set");
    int rc = fetch_counts(s);   // synchronises the stream: every pending record is complete
    if (rc) return rc;
    while (s->ev_pending > 0) {
        rc = fold_oldest_timing(s, false);
        if (rc) return rc;
    }
    *out = s->t_acc;
    s->t_acc = nbody_timings{};
    return NBODY_OK;
}

int nbody_direct_sum_dev(const float *targets_posm, int32_t nt, const float *sources_posm,
                         int32_t ns, float softening_factor, float gravity, float *ax, float *ay,
                         float *az, int32_t accumulate, void *stream)
{
    if (!targets_posm || !sources_posm || !ax || !ay || !az || nt < 0 || ns < 0 ||
        !(softening_factor > 0.0f))
        return NBODY_ERR_INVALID;
    hipError_t e = launch_direct(reinterpret_cast<const Float4 *>(targets_posm), nt,
                                 reinterpret_cast<const Float4 *>(sources_posm), ns,
                                 softening_factor * softening_factor, gravity, ax, ay, az,
                                 accumulate, (hipStream_t) stream);
    return e == hipSuccess ? NBODY_OK : NBODY_ERR_HIP;
}

int nbody_direct_sum(nbody_t *h, float *ax, float *ay, float *az)
{
    NB_ENTER(h);
    if (!s->spawned) return fail(s, NBODY_ERR_STATE, "direct sum before spawn/set_state");
    const size_t n = (size_t) s->n;
    if (s->fp64) {   // computed in double, narrowed
        std::vector<double> t(3 * n);
        int rc = nbody_direct_sum_f64(h, &t[0], &t[n], &t[2 * n]);
        if (rc) return rc;
        for (size_t i = 0; i < n; ++i) {
            if (ax) ax[i] = (float) t[i];
            if (ay) ay[i] = (float) t[n + i];
            if (az) az[i] = (float) t[2 * n + i];
        }
        return NBODY_OK;
    }
    float *d = nullptr;
    NB_HIP(s, hipMalloc((void **) &d, 3 * n * 4));
    hipError_t e = launch_direct(s->posm, s->n, s->posm, s->n,
                                 s->p.softening_factor * s->p.softening_factor, s->p.gravity, d,
                                 d + n, d + 2 * n, 0, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    if (e == hipSuccess && ax) e = hipMemcpy(ax, d, n * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && ay) e = hipMemcpy(ay, d + n, n * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && az) e = hipMemcpy(az, d + 2 * n, n * 4, hipMemcpyDeviceToHost);
    (void) hipFree(d);
    NB_HIP(s, e);
    return NBODY_OK;
}

// The direct sum evaluated in double: on a double-precision handle over its own state, on a
// float handle over its float positions and masses widened (the accuracy reference for the
// float tree forces at sizes a host loop cannot reach).
int nbody_direct_sum_f64(nbody_t *h, double *ax, double *ay, double *az)
{
    NB_ENTER(h);
    if (!s->spawned) return fail(s, NBODY_ERR_STATE, "direct sum before spawn/set_state");
    const size_t n = (size_t) s->n;
    double *d = nullptr;
    Double4 *wide = nullptr;
    NB_HIP(s, hipMalloc((void **) &d, 3 * n * 8));
    hipError_t e = hipSuccess;
    const Double4 *src = s->d.posm;
    if (!s->fp64) {
        std::vector<Float4> p4(n);
        std::vector<Double4> p8(n);
        e = hipMemcpyAsync(p4.data(), s->posm, n * sizeof(Float4), hipMemcpyDeviceToHost, s->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
        for (size_t i = 0; i < n; ++i) p8[i] = Double4{p4[i].x, p4[i].y, p4[i].z, p4[i].w};
        if (e == hipSuccess) e = hipMalloc((void **) &wide, n * sizeof(Double4));
        if (e == hipSuccess) e = hipMemcpy(wide, p8.data(), n * sizeof(Double4), hipMemcpyHostToDevice);
        src = wide;
    }
    const double eps = (double) s->p.softening_factor;
    if (e == hipSuccess)
        e = launch64_direct(src, s->n, src, s->n, eps * eps, (double) s->p.gravity, d, d + n, d + 2 * n,
                            nullptr, 0, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    if (e == hipSuccess && ax) e = hipMemcpy(ax, d, n * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess && ay) e = hipMemcpy(ay, d + n, n * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess && az) e = hipMemcpy(az, d + 2 * n, n * 8, hipMemcpyDeviceToHost);
    (void) hipFree(d);
    if (wide) (void) hipFree(wide);
    NB_HIP(s, e);
    return NBODY_OK;
}

// ---- multi-GPU locally-essential-tree exchange (no counterpart in the single-GPU reference) ----

int nbody_get_bbox(nbody_t *h, float *lo_hi)
{
    NB_ENTER(h);
    NB_FLOAT_ONLY(s, "nbody_get_bbox");
    if (!lo_hi) return fail(s, NBODY_ERR_INVALID, "null output");
    if (!s->spawned || s->n < 1) return fail(s, NBODY_ERR_STATE, "bbox of an empty handle");
    uint32_t *box = s->let_counts + LET_MAX_BOXES;
    const uint32_t init[6] = {0x7f7fffffu, 0x7f7fffffu, 0x7f7fffffu, 0u, 0u, 0u};
    NB_HIP(s, hipMemcpyAsync(box, init, sizeof init, hipMemcpyHostToDevice, s->stream));
    NB_HIP(s, launch_bbox(s, box));
    NB_HIP(s, hipMemcpyAsync(lo_hi, box, 6 * 4, hipMemcpyDeviceToHost, s->stream));
    NB_HIP(s, hipStreamSynchronize(s->stream));
    return NBODY_OK;
}

int nbody_let_export(nbody_t *h, const float *boxes, int32_t nboxes, float *out_dev,
                     int32_t capacity, int32_t *counts)
{
    NB_ENTER(h);
    NB_FLOAT_ONLY(s, "nbody_let_export");
    if (!boxes || !out_dev || !counts || nboxes < 0 || nboxes > LET_MAX_BOXES || capacity < 0)
        return fail(s, NBODY_ERR_INVALID, "let_export: bad arguments");
    if (!s->spawned || s->n < 1) return fail(s, NBODY_ERR_STATE, "let_export before spawn/set_state");
    if (nboxes == 0) return NBODY_OK;
    // every box is a remote domain described by one frontier record
    std::vector<float> rec((size_t) nboxes * LET_REC, 0.f);
    for (int i = 0; i < nboxes; ++i) std::memcpy(&rec[(size_t) i * LET_REC], boxes + 6 * i, 6 * sizeof(float));
    NB_HIP(s, hipMemcpyAsync(s->let_boxes, rec.data(), rec.size() * sizeof(float), hipMemcpyHostToDevice, s->stream));
    for (int b0 = 0; b0 < nboxes; b0 += LET_MASK_DOMAINS) {
        const int nd = std::min(LET_MASK_DOMAINS, nboxes - b0);
        NB_HIP(s, launch_let_export(s, s->let_boxes + (size_t) b0 * LET_REC, nd, 1, -1,
                                    reinterpret_cast<Float4 *>(out_dev) + (size_t) b0 * capacity, capacity,
                                    s->let_counts + b0));
    }
    NB_HIP(s, hipMemcpyAsync(counts, s->let_counts, (size_t) nboxes * 4, hipMemcpyDeviceToHost, s->stream));
    NB_HIP(s, hipStreamSynchronize(s->stream));   // also keeps `rec` alive until it has been copied
    for (int i = 0; i < nboxes; ++i)
        if (counts[i] > capacity)
            return fail(s, NBODY_ERR_CAPACITY, "let_export: a box needs more records than capacity");
    return NBODY_OK;
}

int nbody_let_frontier_dev(nbody_t *h, float *records_dev, int32_t max_records)
{
    NB_ENTER(h);
    if (!records_dev || max_records < 1 || max_records > LET_MAX_T)
        return fail(s, NBODY_ERR_INVALID, "let_frontier: max_records outside [1, 32]");
    if (!s->spawned || s->n < 1) return fail(s, NBODY_ERR_STATE, "let_frontier of an empty handle");
    NB_HIP(s, launch_frontier(s, records_dev, max_records));
    return NBODY_OK;
}

int nbody_let_frontier(nbody_t *h, float *records, int32_t max_records)
{
    NB_ENTER(h);
    if (!records) return fail(s, NBODY_ERR_INVALID, "null output");
    int rc = nbody_let_frontier_dev(h, s->let_frontier, max_records);
    if (rc) return rc;
    NB_HIP(s, hipMemcpyAsync(records, s->let_frontier, (size_t) max_records * LET_REC * sizeof(float),
                             hipMemcpyDeviceToHost, s->stream));
    NB_HIP(s, hipStreamSynchronize(s->stream));
    return NBODY_OK;
}

int nbody_let_export_dev(nbody_t *h, const float *frontiers_dev, int32_t ndomains, int32_t records_per_domain,
                         int32_t skip_domain, float *out_dev, int32_t capacity, int32_t *counts_dev)
{
    NB_ENTER(h);
    NB_FLOAT_ONLY(s, "nbody_let_export_dev");
    if (!frontiers_dev || !out_dev || !counts_dev || ndomains < 1 || ndomains > LET_MAX_BOXES ||
        records_per_domain < 1 || records_per_domain > LET_MAX_T || capacity < 0)
        return fail(s, NBODY_ERR_INVALID, "let_export_dev: bad arguments");
    if (!s->spawned || s->n < 1) return fail(s, NBODY_ERR_STATE, "let_export before spawn/set_state");
    for (int d0 = 0; d0 < ndomains; d0 += LET_MASK_DOMAINS) {
        const int nd = std::min(LET_MASK_DOMAINS, ndomains - d0);
        NB_HIP(s, launch_let_export(s, frontiers_dev + (size_t) d0 * records_per_domain * LET_REC, nd,
                                    records_per_domain, skip_domain - d0,
                                    reinterpret_cast<Float4 *>(out_dev) + (size_t) d0 * capacity, capacity,
                                    reinterpret_cast<uint32_t *>(counts_dev) + d0));
    }
    return NBODY_OK;
}

// Source handle: particles only (no velocities, no forces of its own).  Everything is queued
// on the handle's stream; nothing is read back.
int nbody_set_sources_dev(nbody_t *h, const float *posm_dev, int32_t count)
{
    NB_ENTER(h);
    NB_FLOAT_ONLY(s, "nbody_set_sources_dev");
    if (count < 0 || count > s->cap || (count > 0 && !posm_dev))
        return fail(s, NBODY_ERR_INVALID, "set_sources: count outside [0, num_points]");
    if (s->p.num_steps_validator > 0)
        return fail(s, NBODY_ERR_STATE, "set_sources on a handle with a validator");
    s->n = count;
    s->pending_vel = false;
    s->uniform_mass = 0.f;
    if (count == 0) { s->spawned = false; return NBODY_OK; }
    NB_HIP(s, hipMemsetAsync(s->counts, 0, sizeof(Counts), s->stream));
    NB_HIP(s, launch_unpack_sources(s, reinterpret_cast<const Float4 *>(posm_dev), count));
    NB_HIP(s, launch_morton_only(s));
    s->old_keys_n = -1;   // (records in the sender's order)
    int rc = build_after_encode(s, false);
    if (rc) return rc;
    NB_HIP(s, launch_build(s));
    NB_HIP(s, launch_com(s));
    s->spawned = true;
    return NBODY_OK;
}

int nbody_traverse_from(nbody_t *h, nbody_t *sources, int32_t accumulate)
{
    NB_ENTER(h);
    NB_FLOAT_ONLY(s, "nbody_traverse_from");
    if (!sources) return fail(s, NBODY_ERR_INVALID, "null sources");
    Sim *src = reinterpret_cast<Sim *>(sources);
    if (src->fp64) return fail(s, NBODY_ERR_STATE, "traverse_from: sources are a double-precision handle");
    if (!s->spawned) return fail(s, NBODY_ERR_STATE, "traverse_from before spawn/set_state");
    if (src->device != s->device || src->key_bytes != s->key_bytes)
        return fail(s, NBODY_ERR_INVALID, "traverse_from: handles differ in device or key width");
    const size_t b = (size_t) s->n * 4;
    if (src->n == 0) {  // nothing to add
        if (!accumulate) {
            NB_HIP(s, hipMemsetAsync(s->ax, 0, b, s->stream));
            NB_HIP(s, hipMemsetAsync(s->ay, 0, b, s->stream));
            NB_HIP(s, hipMemsetAsync(s->az, 0, b, s->stream));
        }
        return NBODY_OK;
    }
    if (!src->spawned) return fail(s, NBODY_ERR_STATE, "traverse_from: sources hold no tree");
    if (src != s) {   // the sources' tree is built on their stream: wait for it on the GPU, not here
        NB_HIP(s, hipEventRecord(src->ev_ready, src->stream));
        NB_HIP(s, hipStreamWaitEvent(s->stream, src->ev_ready, 0));
    }
    // with NBODY_FLAG_TIMING up to two calls per step (own tree, imported tree) are timed and
    // their interactions counted into the record of the last nbody_update()
    const bool timed = (s->p.flags & NBODY_FLAG_TIMING) && s->ev_pending > 0;
    const int slot = (s->ev_head + TIMING_RING - 1) % TIMING_RING;
    const int k = timed ? s->tf_calls[slot] : 2;
    if (k < 2) NB_HIP(s, hipEventRecord(s->ev[slot][EV_TF0_BEGIN + 2 * k], s->stream));
    const int g0 = s->trav_g0, g1 = s->trav_g1;
    s->trav_g1 = -1;  // all groups
    hipError_t e = launch_traverse_from(s, src, accumulate != 0);
    s->trav_g0 = g0; s->trav_g1 = g1;
    NB_HIP(s, e);
    if (k < 2) {
        NB_HIP(s, hipEventRecord(s->ev[slot][EV_TF0_END + 2 * k], s->stream));
        NB_HIP(s, launch_count_interactions(s, (unsigned long long *) (s->inter_dev + slot), true));
        s->tf_calls[slot] = k + 1;
    }
    if (src != s) {   // the sources may be rebuilt only after this traversal has read them
        NB_HIP(s, hipEventRecord(s->ev_ready, s->stream));
        NB_HIP(s, hipStreamWaitEvent(src->stream, s->ev_ready, 0));
    }
    return NBODY_OK;
}

}  // extern "C"

// Room for imported tree copies behind the own tree: node arrays from index 2 * cap on, particle
// arrays from imp_part_base(s) on.  Growing re-allocates the four arrays and keeps their contents.
static int imp_part_base(const Sim *s) { return (s->cap + 3) & ~3; }

template <typename R4>
static int grow_import_arrays(Sim *s, R4 *&nodeA, R4 *&posm, int ncap, int pcap)
{
    const size_t own_nodes = 2 * (size_t) s->cap, pb = (size_t) imp_part_base(s);
    R4 *nA = nullptr, *pm = nullptr;
    NodeB *nB = nullptr;
    NB_HIP(s, dalloc(&nA, own_nodes + ncap)); NB_HIP(s, dalloc(&nB, own_nodes + ncap));
    NB_HIP(s, dalloc(&pm, pb + pcap));
    NB_HIP(s, hipMemcpyAsync(nA, nodeA, (own_nodes + s->imp_node_cap) * sizeof(R4), hipMemcpyDeviceToDevice, s->stream));
    NB_HIP(s, hipMemcpyAsync(nB, s->nodeB, (own_nodes + s->imp_node_cap) * sizeof(NodeB), hipMemcpyDeviceToDevice, s->stream));
    const size_t old_parts = s->imp_part_cap ? pb + (size_t) s->imp_part_cap : (size_t) s->cap;
    NB_HIP(s, hipMemcpyAsync(pm, posm, old_parts * sizeof(R4), hipMemcpyDeviceToDevice, s->stream));
    float *pm2 = nullptr;
    if (!s->fp64) {   // the pair-interleaved copy of the float traversal
        const size_t posm2_floats = 16 * ((pb + (size_t) pcap + 3) / 4);
        NB_HIP(s, dalloc(&pm2, posm2_floats));
        NB_HIP(s, hipMemsetAsync(pm2, 0, posm2_floats * 4, s->stream));
        NB_HIP(s, hipMemcpyAsync(pm2, s->posm2, 16 * ((old_parts + 3) / 4) * sizeof(float), hipMemcpyDeviceToDevice, s->stream));
    }
    NB_HIP(s, hipStreamSynchronize(s->stream));
    NB_HIP(s, hipFree(nodeA)); NB_HIP(s, hipFree(s->nodeB)); NB_HIP(s, hipFree(posm));
    if (!s->fp64) { NB_HIP(s, hipFree(s->posm2)); s->posm2 = pm2; }
    nodeA = nA; s->nodeB = nB; posm = pm;
    s->imp_node_cap = ncap; s->imp_part_cap = pcap;
    return NBODY_OK;
}

static int ensure_import_capacity(Sim *s, int nodes, int parts)
{
    if (nodes <= s->imp_node_cap && parts <= s->imp_part_cap) return NBODY_OK;
    const int ncap = std::max(s->imp_node_cap, nodes + nodes / 4 + 1024);
    const int pcap = std::max(s->imp_part_cap, parts + parts / 4 + 1024);
    return s->fp64 ? grow_import_arrays<Double4>(s, s->d.nodeA, s->d.posm, ncap, pcap)
                   : grow_import_arrays<Float4>(s, s->nodeA, s->posm, ncap, pcap);
}

extern "C" {

int nbody_let_export_tree_dev(nbody_t *h, const float *frontiers_dev, int32_t ndomains, int32_t records_per_domain,
                              int32_t skip_domain, void *out_dev, size_t capacity_bytes, int32_t *counts_dev)
{
    NB_ENTER(h);
    if (!frontiers_dev || !out_dev || !counts_dev || ndomains < 1 || ndomains > LET_MAX_BOXES ||
        records_per_domain < 1 || records_per_domain > LET_MAX_T || (capacity_bytes & 15))
        return fail(s, NBODY_ERR_INVALID, "let_export_tree_dev: bad arguments");
    if (!s->spawned || s->n < 1) return fail(s, NBODY_ERR_STATE, "let_export_tree before spawn/set_state");
    if (!s->let_reach) {
        NB_HIP(s, dalloc(&s->let_reach, 2 * (size_t) s->cap));
        NB_HIP(s, dalloc(&s->let_bsum, (size_t) LET_MASK_DOMAINS * ((2 * (size_t) s->cap + 255) / 256) * 3));
    }
    for (int d0 = 0; d0 < ndomains; d0 += LET_MASK_DOMAINS) {
        const int nd = std::min(LET_MASK_DOMAINS, ndomains - d0);
        NB_HIP(s, launch_let_export_tree(s, frontiers_dev + (size_t) d0 * records_per_domain * LET_REC, nd,
                                         records_per_domain, skip_domain - d0,
                                         reinterpret_cast<unsigned char *>(out_dev) + (size_t) d0 * capacity_bytes,
                                         capacity_bytes, reinterpret_cast<uint32_t *>(counts_dev) + 2 * d0));
    }
    return NBODY_OK;
}

int nbody_import_trees_dev(nbody_t *h, const void *recv_dev, int32_t ncopies, const size_t *byte_offset,
                           const int32_t *node_count, const int32_t *record_count)
{
    NB_ENTER(h);
    if (ncopies < 0 || ncopies > LET_MAX_BOXES || (ncopies > 0 && (!recv_dev || !byte_offset || !node_count || !record_count)))
        return fail(s, NBODY_ERR_INVALID, "import_trees: bad arguments");
    if (!s->spawned) return fail(s, NBODY_ERR_STATE, "import_trees before spawn/set_state");
    if (!s->imp_roots) {
        NB_HIP(s, dalloc(&s->imp_roots, (size_t) LET_MAX_BOXES));
        NB_HIP(s, dalloc(&s->imp_desc, (size_t) 6 * LET_MAX_BOXES + 1));
    }
    long long nodes = 0, parts = 0;
    for (int r = 0; r < ncopies; ++r) {
        if (node_count[r] < 0 || record_count[r] < 0 || (byte_offset[r] & 15) || (node_count[r] == 0 && record_count[r] != 0))
            return fail(s, NBODY_ERR_INVALID, "import_trees: bad copy descriptor");
        nodes += node_count[r];
        parts += record_count[r];
    }
    // (the double traversal keeps a flag in bit 30 of its stack entries)
    if (nodes + parts > 0x7fffffffLL || 2LL * s->cap + nodes > (s->fp64 ? 0x3fffffffLL : 0x7fffffffLL))
        return fail(s, NBODY_ERR_CAPACITY, "import_trees: too many records");
    int rc = ensure_import_capacity(s, (int) nodes, (int) parts);
    if (rc) return rc;
    // (pageable staging: a copy from it has read the words by the time the call returns)
    std::vector<int32_t> words((size_t) 7 * LET_MAX_BOXES + 1);
    int32_t *desc = words.data(), *prefix = desc + 5 * LET_MAX_BOXES, *roots = prefix + LET_MAX_BOXES + 1;
    int nb = 2 * s->cap, pb = imp_part_base(s), items = 0, nroots = 0;
    for (int r = 0; r < ncopies; ++r) {
        desc[5 * r] = (int32_t) (byte_offset[r] / 16);
        desc[5 * r + 1] = node_count[r]; desc[5 * r + 2] = record_count[r];
        desc[5 * r + 3] = nb; desc[5 * r + 4] = pb;
        prefix[r] = items;
        if (node_count[r] > 0) roots[nroots++] = nb;
        items += node_count[r] + record_count[r];
        nb += node_count[r]; pb += record_count[r];
    }
    prefix[ncopies] = items;
    s->imp_nroots = 0;
    if (items > 0) {
        NB_HIP(s, hipMemcpyAsync(s->imp_desc, desc, ((size_t) 6 * LET_MAX_BOXES + 1) * 4, hipMemcpyHostToDevice, s->stream));
        NB_HIP(s, hipMemcpyAsync(s->imp_roots, roots, (size_t) std::max(nroots, 1) * 4, hipMemcpyHostToDevice, s->stream));
        NB_HIP(s, launch_let_graft(s, ncopies, s->imp_desc, s->imp_desc + 5 * LET_MAX_BOXES, items,
                                   reinterpret_cast<const unsigned char *>(recv_dev)));
        s->imp_nroots = nroots;
    }
    return NBODY_OK;
}

}  // extern "C"

int nb::traverse_forest(Sim *s, int what, int group_begin, int group_end, bool kick)
{
    if (!s->spawned) return fail(s, NBODY_ERR_STATE, "traverse_forest before spawn/set_state");
    const int ngroups = (s->n + s->group - 1) / s->group;
    if (group_end < 0) { group_begin = 0; group_end = ngroups; }
    if (what < 0 || what > 2 || group_begin < 0 || group_begin > group_end || group_end > ngroups)
        return fail(s, NBODY_ERR_INVALID, "traverse_forest: bad arguments");
    if (group_begin == group_end || (what == 2 && s->imp_nroots == 0)) return NBODY_OK;
    const bool timed = (s->p.flags & NBODY_FLAG_TIMING) && s->ev_pending > 0;
    const int slot = (s->ev_head + TIMING_RING - 1) % TIMING_RING;
    const int k = timed ? s->tf_calls[slot] : 2;
    if (k < 2) NB_HIP(s, hipEventRecord(s->ev[slot][EV_TF0_BEGIN + 2 * k], s->stream));
    const int g0 = s->trav_g0, g1 = s->trav_g1;
    s->trav_g0 = group_begin; s->trav_g1 = group_end;
    // (double handles have no fused kick: theirs is its own launch, over every particle, and
    // only a pass that covers all of them may ask for it)
    hipError_t e = s->fp64 ? launch64_traverse_forest(s, what) : launch_traverse_forest(s, what, kick);
    if (e == hipSuccess && s->fp64 && kick) e = launch64_vel_kick(s);
    s->trav_g0 = g0; s->trav_g1 = g1;
    NB_HIP(s, e);
    if (k < 2) NB_HIP(s, hipEventRecord(s->ev[slot][EV_TF0_END + 2 * k], s->stream));
    if (timed) {
        // list lengths of the groups of this call (a pass of the copies alone replaces a group's
        // length by the copies' share, which is counted on top of the own tree's)
        s->trav_g0 = group_begin; s->trav_g1 = group_end;
        e = launch_count_interactions(s, (unsigned long long *) (s->inter_dev + slot), false, true);
        s->trav_g0 = g0; s->trav_g1 = g1;
        NB_HIP(s, e);
    }
    if (k < 2) s->tf_calls[slot] = k + 1;
    return NBODY_OK;
}

extern "C" {

int nbody_traverse_forest(nbody_t *h, int32_t what, int32_t group_begin, int32_t group_end)
{
    NB_ENTER(h);
    return nb::traverse_forest(s, what, group_begin, group_end, false);
}

// Energy of the current state with the potential taken from the tree (the reference only has
// the all-pairs energy of its validator, src/cuda/validator.cu:19-63):
//   kinetic = sum m v^2 / 2,  potential = -G/2 sum_i m_i sum_{j != i} m_j / sqrt(d_ij^2 + eps^2)
// where the inner sum runs over the same sources the force traversal uses for i's group.
int nbody_tree_energy(nbody_t *h, double *kinetic, double *potential)
{
    NB_ENTER(h);
    if (!s->spawned) return fail(s, NBODY_ERR_STATE, "energy before spawn/set_state");
    if (s->pending_vel) return fail(s, NBODY_ERR_STATE, "nbody_finish_step not called");
    // per-body potential sums through the tree, then kinetic and potential energy reduced on
    // the device: two doubles cross PCIe
    const int c = s->cur;
    const double G = (double) s->p.gravity;
    if (s->fp64) {
        State64 &d = s->d;
        if (!d.pot) NB_HIP(s, dalloc(&d.pot, (size_t) s->cap));
        NB_HIP(s, launch64_traverse(s, true));
        NB_HIP(s, launch_energy_reduce64(s->n, d.m[c], 1, d.vx[c], d.vy[c], d.vz[c], d.pot, G, nullptr,
                                         nullptr, s->red_scratch, s->red_out, s->stream));
    } else {
        if (!s->pot) {
            NB_HIP(s, dalloc(&s->pot, (size_t) s->cap));
            NB_HIP(s, hipMemcpy(reinterpret_cast<char *>(s->trav_tail) + offsetof(TravTail, pot), &s->pot,
                                sizeof(float *), hipMemcpyHostToDevice));
        }
        NB_HIP(s, launch_potential(s));
        NB_HIP(s, launch_energy_reduce(s->n, s->m[c], 1, s->vx[c], s->vy[c], s->vz[c], s->pot, G, nullptr,
                                       nullptr, s->red_scratch, s->red_out, s->stream));
    }
    NB_HIP(s, hipMemcpyAsync(s->red_host, s->red_out, 2 * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    int rc = fetch_counts(s);   // synchronises the stream
    if (rc) return rc;
    if (kinetic) *kinetic = s->red_host[0];
    if (potential) *potential = s->red_host[1];
    return NBODY_OK;
}

// ---- snapshots: binary SoA state for checkpoint / restart (the reference has no state I/O) ----
namespace {
struct SnapHeader {
    char magic[8];       // "NBSNAP1"
    int32_t n;
    int32_t real_bytes;  // 8: double payload, anything else: float payload
    float domain_size;
    float pad;
};
}  // namespace

extern "C++" {
namespace {
template <typename R>
int save_snapshot(Sim *s, const char *path)
{
    const size_t n = (size_t) s->n;
    std::vector<R> buf(10 * n);
    std::vector<uint32_t> id(n);
    int rc = get_state<R>(s, &buf[0], &buf[n], &buf[2 * n], &buf[3 * n], &buf[4 * n], &buf[5 * n],
                          &buf[6 * n], &buf[7 * n], &buf[8 * n], &buf[9 * n], id.data());
    if (!rc) rc = fetch_counts(s);
    if (rc) return rc;
    SnapHeader hd{};
    std::memcpy(hd.magic, "NBSNAP1", 8);
    hd.n = s->n;
    hd.real_bytes = (int32_t) sizeof(R);
    hd.domain_size = s->p.domain_size;
    std::ofstream out(path, std::ios::binary);
    if (!out) return fail(s, NBODY_ERR_IO, std::string("cannot open ") + path);
    out.write(reinterpret_cast<const char *>(&hd), sizeof hd);
    out.write(reinterpret_cast<const char *>(buf.data()), (std::streamsize) (buf.size() * sizeof(R)));
    out.write(reinterpret_cast<const char *>(id.data()), (std::streamsize) (n * 4));
    return out ? NBODY_OK : fail(s, NBODY_ERR_IO, "write failed");
}

template <typename R>
int load_snapshot(Sim *s, std::ifstream &in, int count)
{
    const size_t n = (size_t) count;
    std::vector<R> buf(10 * n);
    std::vector<uint32_t> id(n);
    in.read(reinterpret_cast<char *>(buf.data()), (std::streamsize) (buf.size() * sizeof(R)));
    in.read(reinterpret_cast<char *>(id.data()), (std::streamsize) (n * 4));
    if (!in) return fail(s, NBODY_ERR_IO, "truncated snapshot");
    return set_particles<R>(s, count, &buf[0], &buf[n], &buf[2 * n], &buf[3 * n], &buf[4 * n], &buf[5 * n],
                            &buf[6 * n], &buf[7 * n], &buf[8 * n], &buf[9 * n], id.data());
}
}  // namespace
}  // extern "C++"

int nbody_save_snapshot(nbody_t *h, const char *path)
{
    NB_ENTER(h);
    if (!path) return fail(s, NBODY_ERR_INVALID, "null path");
    if (!s->spawned) return fail(s, NBODY_ERR_STATE, "snapshot before spawn/set_state");
    if (s->pending_vel)   // half a step: positions advanced, velocities not yet kicked
        return fail(s, NBODY_ERR_STATE, "snapshot between nbody_update and nbody_finish_step");
    return s->fp64 ? save_snapshot<double>(s, path) : save_snapshot<float>(s, path);
}

int nbody_load_snapshot(nbody_t *h, const char *path)
{
    NB_ENTER(h);
    if (!path) return fail(s, NBODY_ERR_INVALID, "null path");
    std::ifstream in(path, std::ios::binary);
    if (!in) return fail(s, NBODY_ERR_IO, std::string("cannot open ") + path);
    SnapHeader hd{};
    in.read(reinterpret_cast<char *>(&hd), sizeof hd);
    if (!in || std::memcmp(hd.magic, "NBSNAP1", 8) != 0) return fail(s, NBODY_ERR_IO, "not a snapshot file");
    if (hd.n < 1 || hd.n > s->cap)
        return fail(s, NBODY_ERR_INVALID, "snapshot holds more particles than num_points");
    if (s->p.num_steps_validator > 0 && hd.n != s->cap)
        return fail(s, NBODY_ERR_STATE, "a handle with a validator keeps num_points particles");
    if (hd.domain_size != s->p.domain_size)
        return fail(s, NBODY_ERR_INVALID, "snapshot was written with another domain_size");
    // a payload of the other precision is converted; a restart is bit-identical only when
    // file and handle agree
    return hd.real_bytes == 8 ? load_snapshot<double>(s, in, hd.n) : load_snapshot<float>(s, in, hd.n);
}

int nbody_set_particles(nbody_t *h, int32_t count, const float *x, const float *y, const float *z,
                        const float *vx, const float *vy, const float *vz, const float *m,
                        const float *ax, const float *ay, const float *az, const uint32_t *id)
{
    NB_ENTER(h);
    if (count < 1 || count > s->cap) return fail(s, NBODY_ERR_INVALID, "set_particles: count outside [1, num_points]");
    if (!x || !y || !z || !m) return fail(s, NBODY_ERR_INVALID, "set_particles: x, y, z, m required");
    if (s->p.num_steps_validator > 0 && count != s->cap)
        return fail(s, NBODY_ERR_STATE, "set_particles: a handle with a validator keeps num_points particles");
    return set_particles<float>(s, count, x, y, z, vx, vy, vz, m, ax, ay, az, id);
}

int nbody_write_validation_history(nbody_t *h, const char *csv_path)
{
    NB_ENTER(h);
    if (!csv_path) return fail(s, NBODY_ERR_INVALID, "null path");
    std::ofstream out(csv_path);
    if (!out) return fail(s, NBODY_ERR_IO, std::string("cannot open ") + csv_path);
    out << "sys_energy,sys_energy_ap,avg_acc_err" << std::endl;
    out.precision(9);
    for (int i = 0; i < s->val_count; ++i)
        out << s->hist_e[i] << "," << s->hist_e_ap[i] << "," << s->hist_err[i] << std::endl;
    return out ? NBODY_OK : fail(s, NBODY_ERR_IO, "write failed");
}

int nbody_get_validation_history(nbody_t *h, double *sys_energy, double *sys_energy_ap,
                                 double *avg_acc_err, int32_t capacity, int32_t *count)
{
    NB_ENTER(h);
    if (!count) return fail(s, NBODY_ERR_INVALID, "null count");
    const int k = std::min<int>(s->val_count, capacity < 0 ? 0 : capacity);
    for (int i = 0; i < k; ++i) {
        if (sys_energy) sys_energy[i] = s->hist_e[i];
        if (sys_energy_ap) sys_energy_ap[i] = s->hist_e_ap[i];
        if (avg_acc_err) avg_acc_err[i] = s->hist_err[i];
    }
    *count = s->val_count;
    return NBODY_OK;
}

}  // extern "C"

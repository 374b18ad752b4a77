// nbody_internal.hpp -- handle layout and stage launchers of libnbody_mi355.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/nbody_c.h"

namespace nb {

// Work distribution constants shared by host sizing code and kernels.
#ifndef NB_SORT_WAVES
#define NB_SORT_WAVES 8
#endif
#ifndef NB_SORT_ITEMS
#define NB_SORT_ITEMS (64 / NB_SORT_WAVES)
#endif
constexpr int SORT_WAVES = NB_SORT_WAVES;           // waves per workgroup of a radix tile
constexpr int SORT_THREADS = 64 * SORT_WAVES;
constexpr int SORT_ITEMS = NB_SORT_ITEMS;           // keys per thread per radix tile
constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS;
#ifndef NB_LB_GROUP
#define NB_LB_GROUP 16
#endif
// rows of 256 look-back words one radix pass over `tiles` tiles needs: one per tile and one
// per group of NB_LB_GROUP tiles (dev_utils.hpp, lb2_exclusive_thread)
constexpr int sort_status_rows(int tiles) { return tiles + (tiles + NB_LB_GROUP - 1) / NB_LB_GROUP; }
constexpr int SCAN_THREADS = 256;
#ifndef NB_SCAN_ITEMS
#define NB_SCAN_ITEMS 16
#endif
constexpr int SCAN_ITEMS = NB_SCAN_ITEMS;           // elements per thread in compaction tiles
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;
constexpr int LET_MAX_BOXES = 256;
#ifndef NB_LET_MAX_T
#define NB_LET_MAX_T 32
#endif
constexpr int LET_MAX_T = NB_LET_MAX_T;             // frontier records a domain advertises (< 256)
constexpr int LET_REC = 12;                         // floats per frontier record
constexpr int LET_MASK_DOMAINS = 32;                // remote domains per export launch (mask bits)
#ifndef NB_TRAV_WAVES
#define NB_TRAV_WAVES 1
#endif
constexpr int TRAV_WAVES = NB_TRAV_WAVES;           // groups (waves) per traversal workgroup
#ifndef NB_TRAV_STACK
#define NB_TRAV_STACK 512
#endif
#ifndef NB_TRAV_LEAF_STACK
#define NB_TRAV_LEAF_STACK 544
#endif
#ifndef NB_TRAV_OCC
#define NB_TRAV_OCC 5
#endif
constexpr int TRAV_STACK = NB_TRAV_STACK;           // LDS inner-node stack entries per wave
constexpr int TRAV_LEAF_STACK = NB_TRAV_LEAF_STACK; // LDS leaf stack entries per wave
constexpr int TRAV_OCC = NB_TRAV_OCC;               // waves per SIMD the traversal is built for (LDS and registers)
constexpr int TRAV_PEND = 96;                       // LDS queue of candidates awaiting the exact test
constexpr int MAX_DEPTH_BINS = 256;                 // depth key space of the node ranking pass
constexpr int TIMING_RING = 64;
constexpr int REDUCE_HEADER_BYTES = 256;            // arrival counter + padding ahead of the partial sums
constexpr int REDUCE_MAX_BLOCKS = 1024;             // workgroups of a handle's energy reductions
enum { EV_BEGIN = 0, EV_INTEGRATE, EV_SORT, EV_BUILD, EV_COM, EV_TRAVERSE,
       EV_TF0_BEGIN, EV_TF0_END, EV_TF1_BEGIN, EV_TF1_END,   // nbody_traverse_from calls of the step
       EV_COUNT };

struct Float4 { float x, y, z, w; };
struct Double4 { double x, y, z, w; };
// Traversal record of an octree node.  Leaf: first/count = particle range, meta = NODE_LEAF.
// Inner node: first/count = child range, meta bit c set iff child c is a leaf.
struct NodeB { float size; int32_t first; int32_t count; uint32_t meta; };
constexpr uint32_t NODE_LEAF = 0x80000000u;

// Output pointers of the traversal kernel, kept in device memory and read once at the end
// of the kernel so that they do not occupy scalar registers during the walk.  Velocity
// buffers come in pairs; the kernel indexes them with the handle's `cur`.
struct TravTail {
    float *ax, *ay, *az;
    uint32_t *list_len;
    uint32_t *err;
    float *vhx[2], *vhy[2], *vhz[2];
    float *vx[2], *vy[2], *vz[2];
    float *pot;
};

// Device-resident counters written by the build kernels (one int each).
struct Counts {
    int32_t U;       // unique codes
    int32_t M;       // leaves after coarsening
    int32_t n_bt;    // radix-tree nodes = 2M-1
    int32_t n_oct;   // octree nodes
    uint32_t err;    // DEV_ERR_* bits
    int32_t pad[3];
};

// Floating-point state of a handle created with NBODY_FLAG_FP64 (the reference's
// Simulation<double>, src/cuda/simulation.cu:256): the arrays a float handle keeps in Sim as
// float, here in double.  Keys, the sort, every tree array, node sizes (domain / 2^level,
// exact in float) and ids are shared with the float path.
struct State64 {
    double *x[2]{}, *y[2]{}, *z[2]{}, *m[2]{};
    double *vhx[2]{}, *vhy[2]{}, *vhz[2]{};
    double *vx[2]{}, *vy[2]{}, *vz[2]{};
    double *ax = nullptr, *ay = nullptr, *az = nullptr;
    Double4 *posm = nullptr;      // (x,y,z,m) in sorted order
    Double4 *wsum = nullptr;      // unnormalised node sums
    Double4 *nodeA = nullptr;     // (comx, comy, comz, mass)
    double *pot = nullptr;
    // validator twin (spawn order) and direct sum at the tree run's positions
    Double4 *tw_posm = nullptr;
    double *tw_v[3]{}, *tw_vh[3]{}, *tw_a[3]{}, *tw_pot = nullptr;
    double *ds_a[3]{}, *ds_pot = nullptr;
    double *ext[7]{};             // caller-owned viewer buffers, double
};

struct Sim {
    nbody_params p{};
    bool fp64 = false;
    State64 d;
    int n = 0;            // particles currently held (<= cap)
    int cap = 0;          // capacity every array is sized for (= params.num_points)
    int group = 64;
    int key_bytes = 4;    // 4 (B=10) or 8 (B=21)
    int device = 0;
    hipStream_t stream = nullptr;
    bool spawned = false;
    std::string err;

    // particle state, SoA, double buffered for the permute
    float *x[2]{}, *y[2]{}, *z[2]{}, *m[2]{};
    float *vhx[2]{}, *vhy[2]{}, *vhz[2]{};
    float *vx[2]{}, *vy[2]{}, *vz[2]{};
    uint32_t *id[2]{};
    float *ax = nullptr, *ay = nullptr, *az = nullptr;
    Float4 *posm = nullptr;       // (x,y,z,m) in sorted order
    float *posm2 = nullptr;       // same, pair-interleaved: (x0,x1,y0,y1,z0,z1,m0,m1) per pair
    int cur = 0;                  // which buffer of each pair is current
    // > 0: every particle has this mass (seen on the host when the state was uploaded): the
    // traversal then leaves the mass out of the particle-particle arithmetic (one packed
    // multiply in fourteen instructions per source pair) and scales at the end.  0 = unknown.
    float uniform_mass = 0.f;

    // fitted key domain (NBODY_FLAG_FIT_KEYS): lo xyz, side, valid + 6 words of bounding-box
    // scratch, device memory; null = keys span [0, domain_size]^3 like the reference's
    float *keybox = nullptr;
    // With several domains (nbody_dd_*) every rank must draw the SAME cube: between the bounding
    // box of the own particles and the decision (k_keyfit) this hook turns the six box words
    // (device, min xyz / max xyz as bit patterns) into the box of ALL ranks' particles.
    int (*bbox_hook)(void *ctx, uint32_t *box_dev, hipStream_t stream) = nullptr;
    void *bbox_hook_ctx = nullptr;

    // sort
    void *keys[2]{};              // KeyT[n] ping-pong
    uint32_t *vals[2]{};          // index ping-pong; after the sort vals[sorted_buf] = perm
    int sorted_buf = 0;
    uint32_t *ghist = nullptr;    // [8][256] digit histograms
    uint32_t *sort_status = nullptr;  // [passes][sort_status_rows(tiles)][256] look-back words + tickets
    size_t sort_status_bytes = 0;
    int sort_tiles = 0;
    // sort of nearly sorted keys (k_sort.hip, k_coh_*): device control words, the moved count the
    // last such pass reported (pinned: {moved, n, sort call, calm streak}), the element count of the keys the last sort left
    // in keys[sorted_buf], how many sorts went which way
    int32_t *coh_ctl = nullptr;
    int32_t *coh_host = nullptr;
    int old_keys_n = -1;
    uint32_t sort_calls = 0;
    uint32_t probe_interval = 8, next_probe = 0;   // when the moved elements are counted without the path being taken
    hipEvent_t ev_hint = nullptr;   // after the kernel that reports the moved count
    uint32_t hint_call = ~0u - 1;   // the sort call (0-based) that recorded it
    uint64_t coh_sorts = 0, full_sorts = 0;

    // scratch zeroed once per build (tickets, look-back words, counters, histograms)
    uint8_t *zero_block = nullptr;   // build part, then the sort's part (ghist points into it)
    bool build_scratch_zeroed = false;  // set by the sort's memset, consumed by the build
    size_t scratch_prezeroed = 0;       // bytes of zero_block the kick-drift kernel has zeroed for the sort
    size_t zero_bytes = 0;
    uint32_t *tickets = nullptr;      // [8]
    uint32_t *scan_status = nullptr;  // [3][max_scan_tiles]
    uint32_t *dhist = nullptr;        // [256] depth histogram
    uint32_t *depth_rows = nullptr;   // [2048][32] per-workgroup depth counts (k_depth)
    uint32_t *rank_status = nullptr;  // [sort_status_rows(rank_tiles)][256]
    uint32_t *com_counter = nullptr;  // [2n] children-arrived counters
    int max_scan_tiles = 0, rank_tiles = 0;

    Counts *counts = nullptr;     // device
    Counts *counts_host = nullptr;  // pinned

    // build products
    void *unique = nullptr;       // KeyT[n]
    int32_t *first_pt = nullptr;  // [n+1]
    int32_t *lfc = nullptr;       // [n+1] leaf_first_code_idx
    void *leaf_key = nullptr;     // KeyT[n] first code of each leaf
    // radix tree (transient), r = radix node id (leaf j -> 2j, internal i -> 2i+1)
    int32_t *rt_pd = nullptr;     // [2n] parent*2 + edge_delta
    int32_t *rt_level = nullptr;  // [2n] lcp/3 (leaves: parent level + 1)
    int32_t *rt_lb = nullptr, *rt_le = nullptr, *rt_left = nullptr, *rt_right = nullptr;  // [n]
    uint32_t *depth_key = nullptr;   // [2n] depth or 255 for edge_delta == 0
    uint32_t *bfs = nullptr;         // [2n] octree id -> radix node
    uint32_t *octid = nullptr;       // [2n] radix node -> octree id
    // octree (capacity 2n)
    uint32_t *o_depth = nullptr;
    int32_t *o_first_child = nullptr, *o_num_children = nullptr, *o_lb = nullptr, *o_le = nullptr;
    int32_t *o_level = nullptr, *o_pb = nullptr, *o_pe = nullptr, *o_parent = nullptr;
    float *o_size = nullptr;
    Float4 *o_wsum = nullptr;     // unnormalised (sum m*x, sum m*y, sum m*z, sum m)
    Float4 *nodeA = nullptr;      // (comx, comy, comz, mass)
    NodeB *nodeB = nullptr;
    uint32_t *list_len = nullptr; // [ceil(n/group)]
    TravTail *trav_tail = nullptr;
    float *pot = nullptr;            // per-body potential sums (allocated by the first energy call)
    // LET exchange scratch
    float *let_boxes = nullptr;      // [LET_MAX_BOXES][LET_REC] device copy of caller-given boxes
    uint32_t *let_counts = nullptr;  // [LET_MAX_BOXES] + 6 words of bounding box
    uint32_t *let_mask = nullptr;    // [2n] per node: domains whose opening criterion it passes
    int32_t *let_seg = nullptr;      // [LET_MAX_T + 2] first particles of the frontier nodes, count
    float *let_frontier = nullptr;   // [LET_MAX_T][LET_REC] frontier of nbody_let_frontier
    uint32_t *let_reach = nullptr;   // [2n] per node: domains for which no ancestor passes
    uint32_t *let_bsum = nullptr;    // [LET_MASK_DOMAINS][ceil(2n/256)][3] sums of the tree export
    // pruned copies of other domains' trees grafted behind the own tree (nbody_import_trees_dev):
    // nodeA/nodeB hold them from index 2*cap on, posm/posm2 from imp_part_base on
    int imp_node_cap = 0, imp_part_cap = 0;
    int32_t *imp_roots = nullptr;    // [LET_MAX_BOXES] device: node index of every copy's root
    int32_t *imp_desc = nullptr;     // [6 * LET_MAX_BOXES + 1] device: copy descriptors + item prefix
    int imp_nroots = 0;
    hipEvent_t ev_ready = nullptr;   // marks this handle's queued work for a waiting handle

    // caller-owned viewer buffers [pos.x, pos.y, pos.z, com.x, com.y, com.z, node_size] (nullable)
    float *ext[7]{};
    bool has_ext = false;

    // sharded traversal
    int trav_g0 = 0, trav_g1 = -1;
    bool pending_vel = false;

    // timing
    hipEvent_t ev[TIMING_RING][EV_COUNT]{};
    int ev_head = 0, ev_pending = 0;
    int tf_calls[TIMING_RING]{};     // timed nbody_traverse_from calls per recorded step (<= 2)
    uint64_t *inter_dev = nullptr;   // [TIMING_RING] interactions per step
    nbody_timings t_acc{};           // records folded out of the ring, not yet handed to the caller
    // validator: all-pairs twin simulation (state in spawn order) + history
    Float4 *tw_posm = nullptr;
    float *tw_v[3]{}, *tw_vh[3]{}, *tw_a[3]{}, *tw_pot = nullptr;
    float *ds_a[3]{}, *ds_pot = nullptr;   // direct sum at the Barnes-Hut positions
    int val_step = 0, val_count = 0;
    unsigned char *red_scratch = nullptr;  // partial sums of the on-device reductions
    double *red_out = nullptr;             // [8] device results
    double *red_host = nullptr;            // [8] pinned copy
    std::vector<double> hist_e, hist_e_ap, hist_err;
};

// ---- stage launchers (each enqueues on s->stream and returns a hipError_t) ----
hipError_t launch_integrate_morton(Sim *s);                 // kick-drift + encode -> keys[0]
size_t scratch_zero_bytes(const Sim *s);                    // what the sort of s->n keys wants zeroed in zero_block
hipError_t launch_morton_only(Sim *s);                      // encode current positions
hipError_t launch_keyfit_only(Sim *s);                      // cube decision from the box words
hipError_t launch_bbox_merge(const uint32_t *all_dev, int ranks, uint32_t *box_dev, hipStream_t st);
hipError_t launch_sort(Sim *s);                             // keys/vals -> sorted keys, perm
hipError_t launch_permute(Sim *s, bool all_fields);         // gather by perm, write posm
hipError_t launch_build(Sim *s);                            // unique .. octree emit
hipError_t launch_com(Sim *s);
hipError_t launch_traverse(Sim *s, bool fuse_vel);
hipError_t launch_vel_kick(Sim *s);
hipError_t launch_direct(const Float4 *targets, int nt, const Float4 *sources, int ns, float eps2,
                         float gravity, float *ax, float *ay, float *az, int accumulate,
                         hipStream_t st);

hipError_t launch_direct_pot(const Float4 *targets, int nt, const Float4 *sources, int ns,
                             float eps2, float gravity, float *ax, float *ay, float *az,
                             float *pot, int self_pairs, hipStream_t st);
hipError_t launch_twin_kick_drift(int n, float dt, float domain, Float4 *posm, const float *vx,
                                  const float *vy, const float *vz, const float *ax,
                                  const float *ay, const float *az, float *vhx, float *vhy,
                                  float *vhz, hipStream_t st);
hipError_t launch_vel_kick_raw(int n, float dt, float keep, const float *vhx, const float *vhy,
                               const float *vhz, const float *ax, const float *ay, const float *az,
                               float *vx, float *vy, float *vz, hipStream_t st);
hipError_t launch_count_interactions(Sim *s, unsigned long long *out, bool all_groups_add = false,
                                     bool range_add = false);
hipError_t launch_traverse_from(Sim *bodies, Sim *sources, bool accumulate);
hipError_t launch_potential(Sim *s);                        // -> s->pot
hipError_t launch_export_view(Sim *s);                      // positions + node COM/size -> ext[]
hipError_t launch_bbox(Sim *s, uint32_t *box_dev);
hipError_t launch_frontier(Sim *s, float *records_dev, int T);
hipError_t launch_let_export(Sim *s, const float *frontiers_dev, int ndom, int T, int skip,
                             Float4 *out, int cap, uint32_t *out_count);
hipError_t launch_unpack_sources(Sim *s, const Float4 *src, int n);
hipError_t launch_let_export_tree(Sim *s, const float *frontiers_dev, int ndom, int T, int skip,
                                  unsigned char *out, size_t cap_bytes, uint32_t *out_count);
hipError_t launch_let_graft(Sim *s, int ncopies, const int32_t *desc_dev, const int32_t *prefix_dev,
                            int total_items, const unsigned char *recv);
hipError_t launch_traverse_forest(Sim *s, int what, bool kick);
// nbody_traverse_forest with the velocity kick of the groups it covers fused into the launch
// (kick = true: the caller sees to it that every group is kicked exactly once and clears
// pending_vel itself).
int traverse_forest(Sim *s, int what, int group_begin, int group_end, bool kick);
hipError_t launch_key_lower_bound(Sim *s, const uint64_t *queries_dev, int nq, int32_t *out_dev);
hipError_t launch_pack_particles(Sim *s, void *out);               // 11 words of the handle's precision per particle
hipError_t launch_unpack_particles(Sim *s, const void *in, int n);
// Sort + build over the n particles k_unpack_particles has just put into the current buffers
// (accelerations in the half-step velocity arrays); everything on the handle's stream.
int install_particles_dev(Sim *s, int n);
// Samples particles [first, first + count) of a set of n_total in the handle's precision and
// installs them (ids = index in the whole set); the handle is sorted and built afterwards.
int spawn_range(Sim *s, int kind, long long first, int count, long long n_total, float p0, float p1);
// Host sampler of the initial conditions for particle indices [first, first + count) of a set of
// n_total particles.
int sample_ic_range(int kind, long long first, int count, long long n_total, uint64_t seed, float domain,
                    float gravity, float p0, float p1, float *x, float *y, float *z, float *vx,
                    float *vy, float *vz, float *m);

// ---- on-device reductions (k_reduce.hip) ----
void noalloc_prepare_current_device();   // resolves nbody_direct_energy_dev_noalloc's scratch address
size_t reduce_scratch_bytes(int blocks, int k);
// out[0] = sum m v^2/2, out[1] = -G/2 sum m pot, out[2] = sum |a - d| / |d| (a, d nullable)
hipError_t launch_energy_reduce(int n, const float *m, int m_stride, const float *vx, const float *vy,
                                const float *vz, const float *pot, double G, const float *const a[3],
                                const float *const d[3], unsigned char *scratch, double *out,
                                hipStream_t st);
hipError_t launch_energy_reduce64(int n, const double *m, int m_stride, const double *vx, const double *vy,
                                  const double *vz, const double *pot, double G, const double *const a[3],
                                  const double *const d[3], unsigned char *scratch, double *out,
                                  hipStream_t st);
hipError_t launch_direct_energy(const Float4 *posm, const float *vx, const float *vy, const float *vz,
                                int n, float eps2, float gravity, unsigned char *scratch, double *out,
                                hipStream_t st);


// ---- double-precision handles (k_fp64.hip) ----
hipError_t launch64_integrate_morton(Sim *s);
hipError_t launch64_morton_only(Sim *s);
hipError_t launch64_permute(Sim *s, bool all_fields);
hipError_t launch64_com(Sim *s);
hipError_t launch64_traverse(Sim *s, bool potential);       // -> d.ax/ay/az, or d.pot
hipError_t launch64_traverse_forest(Sim *s, int what);      // own tree and/or imported copies
hipError_t launch64_vel_kick(Sim *s);
hipError_t launch64_export_view(Sim *s);
hipError_t launch64_direct(const Double4 *targets, int nt, const Double4 *sources, int ns,
                           double eps2, double gravity, double *ax, double *ay, double *az,
                           double *pot, int self_pairs, hipStream_t st);
hipError_t launch64_twin_kick_drift(int n, double dt, double domain, Double4 *posm,
                                    double *const v[3], double *const a[3], double *const vh[3],
                                    hipStream_t st);
hipError_t launch64_vel_kick_raw(int n, double dt, double keep, double *const vh[3],
                                 double *const a[3], double *const v[3], hipStream_t st);

}  // namespace nb

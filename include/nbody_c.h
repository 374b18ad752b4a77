/* nbody_c.h -- C-ABI of libnbody_mi355.so, the MI355X (gfx950) Barnes-Hut step engine.
 *
 * This is the drop-in boundary for the reference's `Simulation<float>` class
 * (reference: include/simulation.hpp:11-47, src/cuda/simulation.cu).  Every entry
 * point cites the reference interface it replaces.  Conventions:
 *   - plain C types only; opaque handle; int status return (0 = NBODY_OK);
 *     nbody_last_error(handle) gives the text of the last failure on that handle
 *     (the reference reports no errors at all and checks no CUDA call);
 *   - a handle is not thread-safe; distinct handles are independent (kernel
 *     arguments instead of the reference's process-global __constant__ parameters,
 *     src/cuda/physics_common.cu:13-14, src/cuda/barnes_hut.cu:12-14);
 *   - all device work of a handle is issued on one HIP stream owned by the handle;
 *     entry points that return host data synchronise that stream, nbody_update()
 *     and nbody_step() do not synchronise at all.
 */
#ifndef NBODY_C_H
#define NBODY_C_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nbody_sim nbody_t;

enum {
    NBODY_OK = 0,
    NBODY_ERR_INVALID = 1,  /* bad argument / unsupported parameter value        */
    NBODY_ERR_HIP = 2,      /* a HIP runtime call failed                         */
    NBODY_ERR_STATE = 3,    /* call sequence error (e.g. update before spawn)    */
    NBODY_ERR_DEVICE = 4,   /* a kernel reported an internal error (spin limit)  */
    NBODY_ERR_IO = 5,
    NBODY_ERR_CAPACITY = 6  /* an output buffer of the caller was too small          */
};

/* nbody_params.flags */
enum {
    NBODY_FLAG_TIMING = 1u << 0,       /* HIP events at a step's begin and around its traversal:
                                          nbody_timings.ms_traverse, ms_total, interactions;
                                          ms_build then holds everything before the traversal */
    NBODY_FLAG_TIMING_STAGES = 1u << 2, /* with TIMING: also mark integrate / sort / build / COM
                                          (three more events, a few microseconds each)        */
    NBODY_FLAG_NO_FUSED_VEL = 1u << 1, /* run the velocity half-kick as its own kernel        */
    NBODY_FLAG_FP64 = 1u << 3,         /* Simulation<double> (reference: src/cuda/simulation.cu:256
                                          and the `template class X<double>` lines of points.cu:228,
                                          octree.cu:367, barnes_hut.cu:677, validator.cu:381): the
                                          handle keeps positions, velocities, masses, accelerations
                                          and centres of mass in double and runs every
                                          floating-point stage in double.  The parameters stay
                                          float, as in the reference (include/simulation.hpp:21-26).
                                          Every entry point works on such a handle except those of
                                          the domain decomposition (bbox, let_export,
                                          set_sources_dev, traverse_from); the sharded traversal
                                          (traversal range, finish_step) does;
                                          float arguments are widened and
                                          float results narrowed, the *_f64 entry points move
                                          doubles unchanged.                                     */
    NBODY_FLAG_FIT_KEYS = 1u << 4,     /* Morton keys relative to a cube fitted to the particles
                                          instead of [0, domain_size]^3 (the reference's,
                                          src/cuda/points.cu:28-30).  With 10 bits per axis the
                                          reference's cube leaves a system that fills a tenth of
                                          the domain 100 cells across; the fitted cube gives it
                                          all 1024.  The cube is kept from step to step while it
                                          holds every particle and the particles' largest extent
                                          is at least 0.4 of its side, and is otherwise re-drawn
                                          around their bounding box (side = 1.25 x the largest
                                          extent), on the device; octree cells subdivide it, so
                                          node sizes are side / 2^level.  Float handles only;
                                          under nbody_dd_* the ranks draw one cube together.
                                          nbody_get_key_domain() returns the cube.            */
    NBODY_FLAG_NO_UNIFORM_MASS = 1u << 5, /* When every particle handed to set_state / spawn has
                                          the same mass m0, a float handle with groups of 64
                                          leaves the mass out of the particle-particle
                                          arithmetic (one packed multiply of the fourteen
                                          instructions per source pair) and scales the sums by
                                          m0 at the end: interaction lists unchanged, forces the
                                          same bits when m0 is a power of two, within a few
                                          float roundings otherwise.  This flag keeps the
                                          general kernel (the reference multiplies by the
                                          mass per pair, include/cuda/physics_common.cuh:62-75). */
    NBODY_FLAG_NO_COHERENT_SORT = 1u << 6, /* With 63-bit keys a step whose particles have barely
                                          moved (the reference's time steps: 1e-6 for the sphere and
                                          the uniform cube, src/renderer.cpp:314-318) sorts only the
                                          elements whose key left the gap between their neighbours'
                                          previous keys and merges them into the rest, which is in
                                          order already -- the same permutation as the full sort.
                                          The library takes that path when the previous sort has
                                          finished and moved at most 1 element in 64, or when the
                                          last three counts it has seen (at most 256 sorts old) moved
                                          at most 1 in 1024.  This flag always runs the digit passes
                                          over all elements.                                    */
    NBODY_FLAG_ALWAYS_COHERENT_SORT = 1u << 11, /* take that path on every sort whose predecessor's
                                            keys are there, however many elements moved and whether or
                                            not the previous sort has finished: for a caller that
                                            queues many steps of a slowly moving system without
                                            waiting (correct for any input; slower than the digit
                                            passes over everything once more than about 1 element in
                                            50 moves per step), and for the tests                  */
    NBODY_FLAG_DEBUG_SMALL_STACK = 1u << 9, /* test aid: the traversal runs with a node stack of
                                            32 usable entries, so that its overflow handling
                                            (push-backs, depth-first rounds) is exercised;
                                            results are unchanged                            */
    NBODY_FLAG_DEBUG_WALK_ONLY = 1u << 8, /* profiling aid: traversal walks the tree and fills
                                            its source lists but skips the force arithmetic
                                            (accelerations are meaningless)                  */
    NBODY_FLAG_DEBUG_LATE_TILES = 1u << 10 /* test aid: in the tiled scan kernels of the build
                                            (digit passes, unique codes, leaf heads) every fourth
                                            workgroup starts 0.3 ms late and waits are given up
                                            after a few polls, so that waiting workgroups take
                                            over the late ones' tiles; results are unchanged   */
};

/* The first 11 fields are Simulation<T>::Params, same order and meaning
 * (reference: include/simulation.hpp:15-28).  mem_traversal_queues is accepted and
 * ignored: the traversal keeps its node stack in LDS and needs no global queues.
 * traversal_group_size must be 64 (one wave64 per group) or 32. */
typedef struct nbody_params {
    int32_t num_points;
    int32_t max_num_codes_per_leaf;
    int32_t traversal_group_size;
    uint64_t mem_traversal_queues;
    float theta;
    float dt;
    float gravity;
    float softening_factor;
    float domain_size;
    float velocity_dampening;
    int32_t num_steps_validator;
    /* extensions */
    int32_t morton_bits; /* bits per axis: 10 (30-bit keys, u32) or 21 (63-bit keys, u64) */
    int32_t device_id;   /* HIP device ordinal; -1 = the calling thread's current device  */
    uint32_t flags;
    uint64_t seed;       /* seed of the spawn_* samplers (reference: 42, simulation.cu:67) */
} nbody_params;

/* Fills *p with the reference renderer's defaults (src/renderer.cpp:297-321, disk
 * values for dt/softening) plus morton_bits=10, traversal_group_size=64, seed=42. */
void nbody_default_params(nbody_params *p);

/* Simulation(Params&)  (reference: include/simulation.hpp:30, src/cuda/simulation.cu:131-136).
 * Unlike the reference's headless constructor this one does install the physics
 * parameters (the reference only does so in the GL constructor, simulation.cu:67-72). */
int nbody_create(const nbody_params *params, nbody_t **out);
/* Simulation(Params&, GLuint buffers[7])  (reference: include/simulation.hpp:31,
 * src/cuda/simulation.cu:39-51,138-158): the viewer-owned buffers
 * [pos.x, pos.y, pos.z, com.x, com.y, com.z, node_size].  The reference maps seven GL buffers
 * and uses them as storage; here dev_ptrs are seven caller-owned DEVICE pointers -- the first
 * three with room for num_points ELEMENTS, the last four for 2*num_points elements (one per
 * octree node) -- into which the library copies the current positions (sorted order) and
 * the octree nodes' centres of mass and sizes after every spawn/set_state/update, on the
 * handle's stream.  An element has the handle's precision: float (4 bytes), or double
 * (8 bytes) on a handle created with NBODY_FLAG_FP64 -- size the buffers accordingly (the
 * reference sizes its buffers for double in either case, src/renderer.cpp:255-267).
 * NULL entries are skipped; dev_ptrs == NULL equals nbody_create. */
int nbody_create_external(const nbody_params *params, void *dev_ptrs[7], nbody_t **out);
/* ~Simulation() (include/simulation.hpp:38). */
void nbody_destroy(nbody_t *sim);
const char *nbody_last_error(nbody_t *sim);

/* spawn_uniform / spawn_sphere / spawn_disk (include/simulation.hpp:32-34,
 * src/cuda/simulation.cu:170-220, include/cuda/initial_conditions.cuh:20-147):
 * same distributions and constants; counter-based RNG keyed by (seed, particle).
 * Each one ends with the reference's exec_post_spawn(): sort + tree build, no forces. */
int nbody_spawn_uniform(nbody_t *sim);
int nbody_spawn_sphere(nbody_t *sim);
int nbody_spawn_disk(nbody_t *sim);
/* Plummer sphere, scale radius a, truncated at r_max, total mass 1, at rest (no
 * counterpart in the reference beyond commented-out code, initial_conditions.cuh:226-243). */
int nbody_spawn_plummer(nbody_t *sim, float a, float r_max);

/* Host-side samplers only (no device work): fill caller arrays of n floats.
 * kind: 0 uniform, 1 sphere, 2 disk, 3 plummer (p0 = a, p1 = r_max). */
int nbody_sample_ic(int kind, int n, uint64_t seed, float domain_size, float gravity, float p0,
                    float p1, float *x, float *y, float *z, float *vx, float *vy, float *vz,
                    float *m);

/* Load particles from host arrays (each n floats; vx/vy/vz may be NULL = 0) and run
 * exec_post_spawn().  No reference counterpart (it has no particle input besides spawn_*). */
int nbody_set_state(nbody_t *sim, const float *x, const float *y, const float *z,
                    const float *vx, const float *vy, const float *vz, const float *m);

/* The same with double arrays.  On a handle without NBODY_FLAG_FP64 the values are narrowed;
 * nbody_set_state() on a handle with it widens. */
int nbody_set_state_f64(nbody_t *sim, const double *x, const double *y, const double *z,
                        const double *vx, const double *vy, const double *vz, const double *m);
int nbody_sample_ic_f64(int kind, int n, uint64_t seed, float domain_size, float gravity, float p0,
                        float p1, double *x, double *y, double *z, double *vx, double *vy,
                        double *vz, double *m);

/* Simulation::update()  (include/simulation.hpp:35, src/cuda/simulation.cu:224-238):
 * kick-drift, Morton sort, tree build, traversal, kick.  nbody_step = n updates. */
int nbody_update(nbody_t *sim);
int nbody_step(nbody_t *sim, int n);
/* Blocks until all queued work of this handle is done; reports device-side errors. */
int nbody_sync(nbody_t *sim);

/* Simulation::get_num_octree_nodes() (include/simulation.hpp:36). */
int nbody_get_num_octree_nodes(nbody_t *sim, int32_t *out);
/* Simulation::write_validation_history() (include/simulation.hpp:37,
 * src/cuda/validator.cu:322-353): CSV `sys_energy,sys_energy_ap,avg_acc_err`. */
int nbody_write_validation_history(nbody_t *sim, const char *csv_path);
/* The same records as arrays (up to `capacity`); *count = records available.  With
 * num_steps_validator > 0 every nbody_update() appends one record (and synchronises).  The
 * record is taken at the END of the update: record i describes the state after step i + 1.
 * The reference takes it at the START of update() (src/cuda/simulation.cu:227-229), so its
 * row i is the state after step i and its row 0 compares two all-zero accelerations; the
 * rows here are the reference's rows shifted by one step.  A record holds the
 * total energy of the tree run, total energy of an all-pairs twin run advanced by direct
 * sums (Validator<T>, src/cuda/validator.cu:238-320), and the step-local mean relative
 * acceleration error of the tree forces against a direct sum at the same positions
 * (metric of src/cuda/validator.cu:38-63).  The record index wraps at num_steps_validator
 * like the reference's curr_step_. */
int nbody_get_validation_history(nbody_t *sim, double *sys_energy, double *sys_energy_ap,
                                 double *avg_acc_err, int32_t capacity, int32_t *count);

/* Current particle state in the CURRENT (Morton-sorted) order; id[i] is the index the
 * particle had when it was spawned / set.  Any pointer may be NULL.  The reference has
 * no getters (state lives in GL buffers, src/cuda/simulation.cu:39-51). */
int nbody_get_state(nbody_t *sim, float *x, float *y, float *z, float *vx, float *vy,
                    float *vz, float *m, float *ax, float *ay, float *az, uint32_t *id);

/* The same as doubles: exact on a handle with NBODY_FLAG_FP64, widened otherwise. */
int nbody_get_state_f64(nbody_t *sim, double *x, double *y, double *z, double *vx, double *vy,
                        double *vz, double *m, double *ax, double *ay, double *az, uint32_t *id);

/* Zero-copy device views, valid until the next update/spawn/set_state.  A handle holds its
 * arrays in one precision: on a handle with NBODY_FLAG_FP64 the float view has n, id and
 * stream set and NULL array pointers, and the _f64 view has the arrays; without the flag it
 * is the other way round. */
typedef struct nbody_dev_view {
    int32_t n;
    const float *x, *y, *z, *vx, *vy, *vz, *ax, *ay, *az, *m;
    const float *posm; /* float4 (x,y,z,m) per particle */
    const uint32_t *id;
    void *stream;      /* hipStream_t of the handle */
} nbody_dev_view;
int nbody_get_device_ptrs(nbody_t *sim, nbody_dev_view *out);
typedef struct nbody_dev_view_f64 {
    int32_t n;
    const double *x, *y, *z, *vx, *vy, *vz, *ax, *ay, *az, *m;
    const double *posm; /* double4 (x,y,z,m) per particle */
    const uint32_t *id;
    void *stream;
} nbody_dev_view_f64;
int nbody_get_device_ptrs_f64(nbody_t *sim, nbody_dev_view_f64 *out);

/* The cube the Morton keys of the last build were taken relative to: out[0..2] = lower
 * corner, out[3] = side.  {0, 0, 0, domain_size} without NBODY_FLAG_FIT_KEYS. */
int nbody_get_key_domain(nbody_t *sim, float *out);

/* How the handle's sorts have gone so far (no counterpart in the reference, which sorts with one
 * library call, src/cuda/points.cu:73-84): sorts that took the nearly-sorted path (see
 * NBODY_FLAG_NO_COHERENT_SORT), sorts by digit passes over all elements, and the number of elements
 * that had left their place in the last sort that counted them (-1: none has).  Waits for the
 * handle's stream.  Pointers may be NULL. */
int nbody_get_sort_stats(nbody_t *sim, uint64_t *coherent_sorts, uint64_t *full_sorts, int32_t *last_moved);

/* Sizes of the last built tree: U unique codes, M leaves, n_oct octree nodes. */
int nbody_get_tree_counts(nbody_t *sim, int32_t *U, int32_t *M, int32_t *n_oct);

/* Host copies of the build products of the last update/spawn (the integer topology
 * contract + centre of mass).  Every pointer may be NULL (skipped); the caller sizes the
 * arrays from num_points / nbody_get_tree_counts: codes,perm: N; unique: U; first_pt:
 * U+1; leaf_first_code_idx: M+1; node arrays: n_oct; list_len: ceil(N/group).
 * codes/unique are widened to u64 for both key widths. */
typedef struct nbody_octree_view {
    uint64_t *codes_sorted;
    int32_t *perm;
    uint64_t *unique;
    int32_t *first_pt;
    int32_t *leaf_first_code_idx;
    int32_t *first_child;   /* -1 for leaves */
    int32_t *num_children;
    int32_t *leaves_begin;
    int32_t *leaves_end;
    int32_t *depth;
    int32_t *level;
    float *size;
    int32_t *points_begin;
    int32_t *points_end;    /* inclusive */
    int32_t *parent;        /* -1 for the root */
    float *comx, *comy, *comz, *mass;
    uint32_t *list_len;     /* per group: sources interacted with in the last traversal */
} nbody_octree_view;
int nbody_get_octree(nbody_t *sim, nbody_octree_view *host_out);
/* Centres of mass and masses of the n_oct nodes as doubles (exact on a handle with
 * NBODY_FLAG_FP64, where nbody_octree_view.comx.. are narrowed copies).  Pointers may be NULL. */
int nbody_get_octree_com_f64(nbody_t *sim, double *comx, double *comy, double *comz, double *mass);

/* Per-stage device times (ms, HIP events on the handle's stream) summed over the steps
 * run since the last call, and the number of steps summed; needs NBODY_FLAG_TIMING.
 * Without NBODY_FLAG_TIMING_STAGES only ms_traverse and ms_total are split out and
 * ms_build holds everything before the traversal.  The library keeps the events of the last
 * 64 steps; records older than that are folded into host-side sums as steps go on (the
 * 65th nbody_update() since the last call may therefore wait for the oldest of those steps
 * to finish), so any number of steps may pass between two calls.  Traversals driven through
 * nbody_traverse_from (at most two per step) count into ms_traverse, ms_total and
 * interactions of the step that precedes them.
 * interactions = sum over those steps of sum_g list_len[g] * bodies_in_group(g). */
typedef struct nbody_timings {
    int32_t steps;
    float ms_integrate; /* kick-drift + Morton encode                        */
    float ms_sort;      /* radix sort + particle permute                     */
    float ms_build;     /* unique, leaves, radix tree, depth, rank, emit     */
    float ms_com;       /* centre-of-mass aggregation                        */
    float ms_traverse;  /* traversal kernel (+ fused velocity kick)          */
    float ms_total;     /* first event to last event of each step            */
    uint64_t interactions;
} nbody_timings;
int nbody_get_timings(nbody_t *sim, nbody_timings *out);

/* Energy of the current state with the potential evaluated through the tree, O(N log N), so
 * that energy conservation can be followed at sizes where the validator's all-pairs energy
 * (reference: src/cuda/validator.cu:19-63) is out of reach:
 *   kinetic = sum m v^2/2,  potential = -G/2 sum_i m_i sum_{j!=i} m_j / sqrt(d_ij^2 + eps^2),
 * the inner sum over the sources the force traversal uses for i's group (monopoles for
 * accepted nodes).  Does not change the state.  Outputs may be NULL. */
int nbody_tree_energy(nbody_t *sim, double *kinetic, double *potential);

/* Checkpoint / restart.  The reference has no state I/O (its particles only ever come from
 * include/cuda/initial_conditions.cuh).  The file holds x,y,z,vx,vy,vz,m,ax,ay,az and the ids
 * of all particles in the handle's current (sorted) order; a handle with the same parameters
 * that loads it continues bit-identically to the run that saved it.  The file may hold fewer
 * particles than the loading handle's num_points (a handle created with head-room); its
 * domain_size must equal the handle's.  Not available between nbody_update() and
 * nbody_finish_step() of a sharded step.  The validator's all-pairs
 * twin is not part of the file.  A handle with NBODY_FLAG_FP64 writes doubles; a file of the
 * other precision loads with conversion. */
int nbody_save_snapshot(nbody_t *sim, const char *path);
int nbody_load_snapshot(nbody_t *sim, const char *path);

/* O(N^2) direct sum (reference: compute_ap_forces, src/cuda/validator.cu:134-174) of
 * nt target bodies against ns source bodies, both float4 (x,y,z,m) device arrays.
 * acc = gravity * sum; if accumulate != 0 the result is added to ax/ay/az (tile-by-tile
 * use by the ring all-gather path).  stream = hipStream_t or NULL. */
int nbody_direct_sum_dev(const float *targets_posm, int32_t nt, const float *sources_posm,
                         int32_t ns, float softening_factor, float gravity, float *ax,
                         float *ay, float *az, int32_t accumulate, void *stream);
/* Direct sum over the handle's own particles (current order); host outputs, n floats each. */
int nbody_direct_sum(nbody_t *sim, float *ax, float *ay, float *az);
/* The same evaluated in double (Validator<double>, src/cuda/validator.cu:381): over the double
 * state of a handle with NBODY_FLAG_FP64; on a float handle over its float positions and
 * masses widened, which is the accuracy reference for the float tree forces at sizes a host
 * loop cannot reach. */
int nbody_direct_sum_f64(nbody_t *sim, double *ax, double *ay, double *az);

/* Total energy of n bodies by direct summation, reduced on the device (reference:
 * compute_conserved_quantities and the cub::DeviceReduce::Sum calls behind it,
 * src/cuda/validator.cu:19-36,65-132,278-287).  posm = n float4 (x,y,z,m) and vx,vy,vz = n
 * floats each, all device arrays; out_dev = two device doubles:
 *   out_dev[0] = sum_i m_i |v_i|^2 / 2
 *   out_dev[1] = -gravity/2 sum_i m_i sum_{j != i} m_j / sqrt(d_ij^2 + softening_factor^2).
 * Per-body potential sums are float like the force kernels, the sums over bodies double and
 * independent of scheduling.  The call allocates and frees a scratch buffer of 16 bytes per
 * 256 bodies and therefore returns only when the result is in out_dev.  stream = hipStream_t
 * or NULL; the stream's device must be current. */
int nbody_direct_energy_dev(const float *posm, const float *vx, const float *vy, const float *vz,
                            int32_t n, float softening_factor, float gravity, double *out_dev,
                            void *stream);

/* nbody_direct_energy_dev for callers that cannot allocate device memory: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  It has no counterpart in the
 * reference.  Same arguments, same kernel and the same result, bit for bit; the partial sums
 * live in a fixed array in device memory (256 KiB + 256 B per GPU, held until the process
 * ends) instead of a buffer allocated per call, so the call allocates and frees nothing,
 * does not synchronise, and is limited to n <= 4,194,304 (NBODY_ERR_CAPACITY beyond that: use
 * nbody_direct_energy_dev).  The stream's device must be current.  All calls on one GPU share
 * that GPU's array: calls that overlap in time on one GPU (different streams, different
 * threads) race; calls on one stream, or otherwise ordered, are safe.
 * Under stream capture: the address of the array is looked up once per device
 * (hipGetSymbolAddress, not a stream operation and not legal inside a capture); nbody_create
 * does that for its handle's device, and so does the first call of this function on a device.
 * A capture may therefore contain calls of this function if a handle has been created on the
 * device, or one call has been made there, BEFORE the capture began. */
int nbody_direct_energy_dev_noalloc(const float *posm, const float *vx, const float *vy,
                                    const float *vz, int32_t n, float softening_factor,
                                    float gravity, double *out_dev, void *stream);

/* Multi-GPU sharded traversal: restrict nbody_update()'s traversal to groups
 * [group_begin, group_end) of the sorted order; the caller (one process per GPU) then
 * exchanges the accelerations (nbody_dev_view.ax/ay/az slices) with an all-gather and
 * calls nbody_finish_step() to apply the velocity kick to every particle.
 * group_end = -1 restores the single-GPU behaviour. */
int nbody_set_traversal_range(nbody_t *sim, int32_t group_begin, int32_t group_end);
int nbody_finish_step(nbody_t *sim);

/* ---------------------------------------------------------------------------------------
 * Multi-GPU domain decomposition along the Morton curve with a locally-essential-tree (LET)
 * exchange.  The reference is single-GPU (its update() is src/cuda/simulation.cu:224-238); the
 * scheme is the hybrid distributed octree of the paper its README cites (Cazalbou et al.,
 * PASC'24, README.md:78): every GPU OWNS the particles of one contiguous range of Morton keys,
 * builds a tree over them alone, tells the others about its domain through the top of that
 * tree, and ships to every other domain only the part of its tree that domain's bodies can
 * ever touch.  One process per GPU.  Per step and domain:
 *   1. nbody_update() with nbody_set_traversal_range(0, 0): kick-drift, sort, build, centres of
 *      mass of the own tree, no forces yet;
 *   2. nbody_let_frontier_dev(): the top of the own tree -- at most 32 nodes that hold every
 *      particle once, each with the tight box of its particles, centre of mass, mass, size;
 *      ALL-GATHER of the frontiers;
 *   3. nbody_let_export_dev(): for every remote domain the (x,y,z,m) records it needs from
 *      this tree -- (centre of mass, mass) of nodes that pass the opening criterion against
 *      the nearest of the domain's frontier boxes, the particles of leaves that do not;
 *      ALL-TO-ALL of the record counts, then of the records;
 *   4. nbody_traverse_from(own, own, 0) (runs while the records travel), nbody_set_sources_dev()
 *      on a second handle builds a tree over the received records,
 *      nbody_traverse_from(own, second, 1) adds their attraction, nbody_finish_step() kicks.
 * Every few hundred steps the key ranges are re-drawn from a global histogram of key prefixes
 * (all-gather) and particles migrate (all-to-all).  nbody_dd_* below runs this choreography
 * natively over caller-supplied collectives; the pieces are exported as well.
 *
 * nbody_get_bbox: lo_hi[0..2] = min, lo_hi[3..5] = max of the handle's current positions.
 * nbody_let_frontier[_dev]: max_records in [1, 32] records of 12 floats each -- lo xyz, hi xyz,
 *   centre of mass xyz, mass, cell size, particle count -- ordered by first particle; unused
 *   records are empty boxes (lo = 3e38, hi = 0).  _dev writes device memory on the handle's
 *   stream and does not synchronise.
 * nbody_let_export_dev: frontiers_dev = ndomains x records_per_domain records (device); domain
 *   skip_domain (the caller's own; -1 = none) gets nothing; records of domain d are written to
 *   out_dev + 4*capacity*d (device, float4 each, order unspecified); counts_dev[d] (device) =
 *   records domain d needs, which may exceed capacity (then out_dev holds none of the surplus:
 *   call again with a larger buffer).  Nothing is read back; ndomains <= 256.
 * nbody_let_export: the same for nboxes <= 256 domains of ONE box each (host array of nboxes x
 *   6 floats), counts returned to the host; NBODY_ERR_CAPACITY if any count > capacity.
 * nbody_set_sources_dev: replace the handle's particles by `count` (x,y,z,m) device records,
 *   0 <= count <= num_points, velocities zero, then sort + build + centres of mass, all queued
 *   on the handle's stream (the records must be complete on that stream; nothing is read
 *   back).  The handle keeps that size until the next set_sources/set_state/spawn.
 * nbody_traverse_from: accelerations of `sim`'s particles due to the tree of `sources` (same
 *   device and key width; theta/softening/gravity of `sim`); added to ax/ay/az when
 *   accumulate != 0.  Ordered after the work queued on the sources' stream by an event, not
 *   by a host wait. */
/* nbody_set_particles: replace the handle's particles by `count` particles, 1 <= count <=
 *   num_points, given as host arrays: positions, velocities (NULL = 0), masses, the
 *   accelerations of their last force evaluation (NULL = 0) and caller-defined 32-bit ids
 *   (NULL = 0..count-1).  Sorts and builds like nbody_set_state but keeps accelerations and
 *   ids, so the step sequence continues as if the particles had always lived here.  The
 *   handle keeps that count until the next set_particles/set_state/spawn. */
int nbody_set_particles(nbody_t *sim, int32_t count, const float *x, const float *y,
                        const float *z, const float *vx, const float *vy, const float *vz,
                        const float *m, const float *ax, const float *ay, const float *az,
                        const uint32_t *id);
int nbody_get_bbox(nbody_t *sim, float *lo_hi);
int nbody_let_frontier(nbody_t *sim, float *records, int32_t max_records);
int nbody_let_frontier_dev(nbody_t *sim, float *records_dev, int32_t max_records);
int nbody_let_export(nbody_t *sim, const float *boxes, int32_t nboxes, float *out_dev,
                     int32_t capacity, int32_t *counts);
int nbody_let_export_dev(nbody_t *sim, const float *frontiers_dev, int32_t ndomains,
                         int32_t records_per_domain, int32_t skip_domain, float *out_dev,
                         int32_t capacity, int32_t *counts_dev);
int nbody_set_sources_dev(nbody_t *sim, const float *posm_dev, int32_t count);
int nbody_traverse_from(nbody_t *sim, nbody_t *sources, int32_t accumulate);

/* The same exchange with the essential part of a tree shipped AS A TREE, so that the receiver
 * neither sorts nor builds anything for it and walks own and imported nodes in ONE traversal
 * (what nbody_dd_step does; no counterpart in the single-GPU reference).
 * nbody_let_export_tree_dev: as nbody_let_export_dev, but domain d receives a pruned copy of this
 *   handle's octree: the REACHED nodes (no ancestor passes d's opening criterion) in breadth-first
 *   order, children contiguous.  Segment d starts at out_dev + capacity_bytes * d and holds
 *   counts_dev[2d] node records of 32 bytes -- (com x, y, z, mass | size, first, count, meta), the
 *   traversal's own node layout, `first` counted from the segment's first node (inner nodes) or
 *   first particle record (leaves, meta = 0x80000000) -- followed by counts_dev[2d+1] particle
 *   records (x,y,z,m).  A reached inner node that passes the criterion is a leaf of one particle
 *   record, its (centre of mass, mass).  If 32 * nodes + 16 * records > capacity_bytes nothing
 *   is written for that domain (call again with more room).  capacity_bytes: multiple of 16.
 *   A handle with NBODY_FLAG_FP64 ships doubles: node records of 48 bytes (four doubles, then
 *   size, first, count, meta), particle records of 32; the frontiers stay float boxes (rounded
 *   outwards) and the criterion that prunes a copy is evaluated in float either way, as it is
 *   conservative: what a domain finally takes from a copy is decided by its own traversal.
 * nbody_import_trees_dev: hang `ncopies` such segments (device memory, complete on the handle's
 *   stream; copy r at recv_dev + byte_offset[r] with node_count[r] nodes and record_count[r]
 *   records, host arrays; empty copies allowed) behind the handle's own tree until the next
 *   nbody_update().  Queued on the handle's stream; grows the handle's node and particle arrays
 *   when the copies need more room than they have (that synchronises).
 * nbody_traverse_forest: accelerations of the particles of groups [group_begin, group_end) of the
 *   sorted order (group_end = -1: every group) in one launch, no velocity kick
 *   (nbody_finish_step()); what = 0: due to the own tree alone, 1: due to the own tree AND the
 *   imported copies in one walk, 2: due to the copies alone, ADDED to what ax/ay/az hold (for
 *   groups that were walked with what = 0 before the copies had arrived). */
int nbody_let_export_tree_dev(nbody_t *sim, const float *frontiers_dev, int32_t ndomains,
                              int32_t records_per_domain, int32_t skip_domain, void *out_dev,
                              size_t capacity_bytes, int32_t *counts_dev);
int nbody_import_trees_dev(nbody_t *sim, const void *recv_dev, int32_t ncopies,
                           const size_t *byte_offset, const int32_t *node_count,
                           const int32_t *record_count);
int nbody_traverse_forest(nbody_t *sim, int32_t what, int32_t group_begin, int32_t group_end);

/* Host-side sampler for a slice of a particle set: particles [first, first + count) of n_total
 * (nbody_sample_ic = first 0, count n_total).  A particle's draws depend only on (seed, its
 * index in the whole set), so processes that sample disjoint slices hold, together, exactly
 * the set one process would sample. */
int nbody_sample_ic_range(int kind, int64_t first, int32_t count, int64_t n_total, uint64_t seed,
                          float domain_size, float gravity, float p0, float p1, float *x, float *y,
                          float *z, float *vx, float *vy, float *vz, float *m);

/* The collectives nbody_dd_* needs, supplied by the caller (RCCL in a C++ host, see
 * examples/dd_rccl_main.cpp; torch.distributed in the Python binding).  Both work on DEVICE memory, are
 * ordered like kernels on the given hipStream_t (work queued on the stream before the call
 * precedes the collective, work queued after it follows), must not block the host longer than
 * enqueueing takes, and return 0 on success.
 *   all_gather: every rank contributes `bytes` at send_dev; recv_dev receives size * bytes,
 *     rank-major.
 *   all_to_all: send_bytes[r] bytes at send_dev + send_off[r] go to rank r, recv_bytes[r] bytes
 *     from rank r arrive at recv_dev + recv_off[r] (host arrays of `size` entries, valid only
 *     during the call; entries may be 0). */
typedef struct nbody_comm {
    int32_t rank, size;
    void *ctx;
    int (*all_gather)(void *ctx, const void *send_dev, void *recv_dev, size_t bytes, void *stream);
    int (*all_to_all)(void *ctx, const void *send_dev, const size_t *send_bytes,
                      const size_t *send_off, void *recv_dev, const size_t *recv_bytes,
                      const size_t *recv_off, void *stream);
} nbody_comm;

typedef struct nbody_dd nbody_dd_t;

/* Domain decomposition over comm->size processes, this one's particles in `own` (a handle of
 * either precision -- the same on every rank -- created with head-room: num_points >= the most
 * particles the domain will ever hold; the handle stays the caller's, and stays usable for
 * getters).  A double handle exchanges doubles: tree copies, migrating particles, sums.
 * *comm is copied. */
int nbody_dd_create(nbody_t *own, const nbody_comm *comm, nbody_dd_t **out);
void nbody_dd_destroy(nbody_dd_t *dd);
const char *nbody_dd_last_error(nbody_dd_t *dd);
/* Initial conditions of n_total particles (kind, p0, p1 as nbody_sample_ic, seed and physics
 * of the handle): every rank samples n_total / size consecutive indices, ids = the global
 * index, then nbody_dd_rebalance() distributes them.  No rank ever holds the whole set. */
int nbody_dd_spawn(nbody_dd_t *dd, int kind, int64_t n_total, float p0, float p1);
/* Re-draws the domains as contiguous Morton-key ranges of (nearly) equal particle counts and
 * migrates the particles: rounds of 4096-bin histograms of key prefixes (12 more bits per
 * round, all-gather; a cut leaves the rounds as soon as a bin edge lies within 0.5 % of a
 * domain's share of its target, and takes the coarsest such cell boundary),
 * one all-to-all of 44-byte particle records (88 on double handles; accelerations and ids included, so the step
 * sequence continues unchanged).  Call between steps; synchronises.  NBODY_ERR_CAPACITY if a
 * domain would exceed its handle's num_points. */
int nbody_dd_rebalance(nbody_dd_t *dd);
/* One step of the whole system (collective: every rank calls it).  One host wait per call
 * (for the record counts); everything else is queued. */
int nbody_dd_step(nbody_dd_t *dd);
/* Current decomposition: splitters[0..size] = first key of each domain (63- or 30-bit keys
 * widened; splitters[size] = 2^bits), and counters of the last step.  Pointers may be NULL. */
typedef struct nbody_dd_stats {
    int64_t n_local;            /* particles this rank owns                       */
    int64_t records_sent;       /* particle records (16 bytes; double handles 32) exported in the last step */
    int64_t records_received;   /* particle records imported in the last step     */
    int64_t migrated_out;       /* particles given away by the last rebalance     */
    int32_t frontier_records;   /* frontier nodes advertised in the last step     */
    int32_t export_capacity;    /* 16-byte units per destination the export buffer holds */
    int64_t nodes_sent;         /* node records (32 bytes; double handles 48) exported in the last step */
    int64_t nodes_received;     /* node records imported in the last step            */
} nbody_dd_stats;
int nbody_dd_get_stats(nbody_dd_t *dd, nbody_dd_stats *out, uint64_t *splitters);
/* Where a step's time goes on this rank, for diagnosing a multi-GPU run: with phase timing on,
 * nbody_dd_step() records a HIP event between its phases (a few microseconds of stream time
 * each: leave it off in timed regions) and nbody_dd_get_phases() returns the last step's times
 * in ms (it waits for that step).  GPU-side spans, in stream order: build (kick-drift, sort,
 * tree, centres of mass), frontier (+ its all-gather), export (+ the all-to-all of the sizes),
 * early_traversal (first groups, own tree only), sizing (the stream idles here while the host
 * reads the sizes and queues the exchange), exchange (all-to-all of the copies: includes waiting
 * for the slowest peer), graft, traversal (own tree + copies, and the copies' pass over the
 * first groups, kick included).  ms_host_wait = host time blocked on the sizes; ms_host_total =
 * host time inside nbody_dd_step. */
typedef struct nbody_dd_phases {
    float ms_build, ms_frontier, ms_export, ms_early_traversal, ms_sizing, ms_exchange, ms_graft,
          ms_traversal, ms_total, ms_host_wait, ms_host_total;
    int32_t pad;
    int64_t bytes_frontier;     /* this rank's contribution to the frontier all-gather */
    int64_t bytes_sent;         /* copies sent in the all-to-all (nodes 32 B, records 16 B; doubles 48, 32) */
    int64_t bytes_received;
} nbody_dd_phases;
int nbody_dd_set_phase_timing(nbody_dd_t *dd, int32_t on);
int nbody_dd_get_phases(nbody_dd_t *dd, nbody_dd_phases *out);
/* Pure host function behind nbody_dd_rebalance: cut positions from a global histogram.
 * hist[nbins] = particles per bin (bins in key order), parts domains, slack = tolerated
 * deviation of a cut from its target in particles.  For cut k = 1..parts-1 writes
 * edge[k-1] = the bin edge in [0, nbins] chosen (the coarsest edge -- most trailing zero bits
 * -- whose cumulative count lies within slack of k * total / parts) or -1 if no edge is
 * within slack, in which case bin[k-1] = the bin that holds the target. */
void nbody_dd_choose_edges(const uint64_t *hist, int32_t nbins, int32_t parts, uint64_t slack,
                           int32_t *edge, int32_t *bin);
/* Refinement round for a cut that no edge so far met: hist[nbins] = particles per sub-bin of
 * the bin that holds the target, base_count = particles before that bin, base_edge = index of
 * the bin's first sub-bin edge among all edges of this resolution, edge_bits = key bits that
 * resolution has.  Returns the sub-bin edge in [0, nbins]: the coarsest within slack of
 * target, else the nearest. */
int32_t nbody_dd_refine_edge(const uint64_t *hist, int32_t nbins, uint64_t base_count, uint64_t target,
                             uint64_t slack, uint64_t base_edge, int32_t edge_bits);

#ifdef __cplusplus
}
#endif
#endif /* NBODY_C_H */

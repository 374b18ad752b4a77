// dd.hip -- multi-GPU domain decomposition along the Morton curve with a locally-essential-tree
// exchange (nbody_dd_* of include/nbody_c.h): the per-step choreography, the re-balancing of
// the key ranges and the distributed initial conditions, over caller-supplied collectives.
//
// The reference is single-GPU and has none of this; the scheme is the hybrid distributed
// octree of the paper its README cites (Cazalbou et al., PASC'24, README.md:78).
//
// A step is queued on the own handle's stream:
//   update (kick-drift, sort, build, centres of mass) -> frontier -> ALL-GATHER frontiers ->
//   export of one pruned copy of the tree per remote domain -> ALL-TO-ALL of the copies' sizes ->
//   copy of the sizes to pinned memory -> [event C] -> ALL-TO-ALL of the copies -> graft behind
//   the own tree -> ONE traversal over own and imported nodes, velocity kick included.
// The host waits for event C only, sizes the exchange from the counts and queues the rest.
// NBODY_DD_EARLY_PERMILLE=k (default 0) walks the first k/1000 of the groups against the own
// tree alone before the host wait, to give the GPU work while the sizes and the copies travel,
// and adds the imported nodes to those groups in a pass of their own at the end; on one GPU
// playing 8 ranks of 2,097,152 particles that costs 0.55 ms per step at k = 60 (two more
// launches, each with its own tail), more than the idle time it can cover, so it is off.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "nbody_internal.hpp"

using namespace nb;

namespace {

constexpr int DD_BINS = 4096;        // histogram bins per refinement round (12 key bits)
constexpr int DD_BIN_BITS = 12;
// migrating particle record (k_pack_particles): eleven words of the handle's precision
inline size_t mig_bytes(const Sim *s) { return s->fp64 ? 88 : 44; }
// a tree copy's records: node = (centre of mass, mass) + NodeB, particle = (position, mass)
inline size_t copy_bytes(const Sim *s, int64_t nodes, int64_t parts)
{
    return s->fp64 ? (size_t) nodes * 48 + (size_t) parts * 32 : (size_t) nodes * 32 + (size_t) parts * 16;
}
constexpr double DD_SLACK = 0.005;   // a cut may miss its target by this share of a domain

struct DD {
    Sim *own = nullptr;
    nbody_comm comm{};
    int T = LET_MAX_T;
    float *fr_send = nullptr, *fr_all = nullptr;
    uint32_t *box_all = nullptr;     // [6W] gathered bounding boxes (fitted key domain)
    uint32_t *box_mine = nullptr;    // [6] neutral box of a domain without particles
    unsigned char *exp_buf = nullptr;   // [W] segments of exp_cap bytes: the copy for each domain
    size_t exp_cap = 0;
    int32_t *cnt_dev = nullptr;      // [4W]: (nodes, records) per destination, then per origin
    int32_t *cnt_host = nullptr;     // pinned copy
    unsigned char *recv_buf = nullptr;
    size_t recv_cap = 0;
    hipEvent_t ev_counts = nullptr;
    int early_permille = 0;          // share of the groups walked before the copies arrive (NBODY_DD_EARLY_PERMILLE)
    // phase timing of the last step (nbody_dd_set_phase_timing): events on the stream + host clock
    bool phase_timing = false, phases_valid = false;
    hipEvent_t pev[9] = {};
    double host_wait_ms = 0.0, host_total_ms = 0.0;
    size_t bytes_sent = 0, bytes_received = 0;
    // re-balancing scratch
    uint64_t *q_dev = nullptr;       // query keys
    int32_t *lb_dev = nullptr;       // their positions in the sorted keys
    uint32_t *hist_dev = nullptr, *hist_all = nullptr;
    size_t hist_words = 0;
    unsigned char *mig_send = nullptr, *mig_recv = nullptr;
    size_t mig_send_cap = 0, mig_recv_cap = 0;
    std::vector<uint64_t> splitters;
    nbody_dd_stats stats{};
    std::string err;
};

int dfail(DD *d, int code, const std::string &msg)
{
    d->err = msg;
    return code;
}

#define DD_HIP(d, call)                                                               \
    do {                                                                              \
        hipError_t e_ = (call);                                                       \
        if (e_ != hipSuccess)                                                         \
            return dfail(d, NBODY_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

#define DD_NB(d, call)                                                                \
    do {                                                                              \
        int rc_ = (call);                                                             \
        if (rc_ != NBODY_OK)                                                          \
            return dfail(d, rc_, std::string(#call) + ": " + nbody_last_error(reinterpret_cast<nbody_t *>((d)->own))); \
    } while (0)

#define DD_COMM(d, call)                                                              \
    do {                                                                              \
        if ((call) != 0) return dfail(d, NBODY_ERR_STATE, std::string("collective failed: ") + #call); \
    } while (0)

struct Guard {
    int prev = -1;
    explicit Guard(int dev) { (void) hipGetDevice(&prev); if (prev != dev) (void) hipSetDevice(dev); }
    ~Guard() { if (prev >= 0) (void) hipSetDevice(prev); }
};

nbody_t *H(Sim *s) { return reinterpret_cast<nbody_t *>(s); }

int key_bits(const Sim *s) { return s->key_bytes == 4 ? 30 : 63; }

int ensure_export_capacity(DD *d, size_t cap_bytes)
{
    cap_bytes = (cap_bytes + 255) / 256 * 256;
    if (cap_bytes <= d->exp_cap) return NBODY_OK;
    if (d->exp_buf) DD_HIP(d, hipFree(d->exp_buf));
    d->exp_buf = nullptr;
    d->exp_cap = 0;
    DD_HIP(d, hipMalloc((void **) &d->exp_buf, (size_t) d->comm.size * cap_bytes));
    d->exp_cap = cap_bytes;
    return NBODY_OK;
}

// Exports the copies, exchanges their sizes and starts the sizes on their way to the host.
int export_and_count(DD *d, bool have)
{
    Sim *s = d->own;
    const int W = d->comm.size, me = d->comm.rank;
    if (have)
        DD_NB(d, nbody_let_export_tree_dev(H(s), d->fr_all, W, d->T, me, d->exp_buf, d->exp_cap, d->cnt_dev));
    else
        DD_HIP(d, hipMemsetAsync(d->cnt_dev, 0, (size_t) 2 * W * sizeof(int32_t), s->stream));
    std::vector<size_t> eight(W, 2 * sizeof(int32_t)), off(W);
    for (int r = 0; r < W; ++r) off[r] = (size_t) r * 2 * sizeof(int32_t);
    DD_COMM(d, d->comm.all_to_all(d->comm.ctx, d->cnt_dev, eight.data(), off.data(), d->cnt_dev + 2 * W,
                                  eight.data(), off.data(), (void *) s->stream));
    DD_HIP(d, hipMemcpyAsync(d->cnt_host, d->cnt_dev, (size_t) 4 * W * sizeof(int32_t),
                             hipMemcpyDeviceToHost, s->stream));
    DD_HIP(d, hipEventRecord(d->ev_counts, s->stream));
    return NBODY_OK;
}

// Gathers `words` uint32 per rank and returns their element-wise sum over the ranks.
int gather_sum(DD *d, size_t words, std::vector<uint64_t> &sum)
{
    Sim *s = d->own;
    const int W = d->comm.size;
    DD_COMM(d, d->comm.all_gather(d->comm.ctx, d->hist_dev, d->hist_all, words * sizeof(uint32_t),
                                  (void *) s->stream));
    std::vector<uint32_t> all(words * W);
    DD_HIP(d, hipMemcpyAsync(all.data(), d->hist_all, all.size() * sizeof(uint32_t), hipMemcpyDeviceToHost,
                             s->stream));
    DD_HIP(d, hipStreamSynchronize(s->stream));
    sum.assign(words, 0);
    for (int r = 0; r < W; ++r)
        for (size_t j = 0; j < words; ++j) sum[j] += all[(size_t) r * words + j];
    return NBODY_OK;
}

// Positions of `nq` query keys in the own sorted keys, on the host.
int lower_bounds(DD *d, const std::vector<uint64_t> &q, std::vector<int32_t> &lb)
{
    Sim *s = d->own;
    lb.assign(q.size(), 0);
    if (s->n < 1 || !s->spawned) return NBODY_OK;   // no particles: every position is 0
    DD_HIP(d, hipMemcpyAsync(d->q_dev, q.data(), q.size() * sizeof(uint64_t), hipMemcpyHostToDevice, s->stream));
    DD_HIP(d, launch_key_lower_bound(s, d->q_dev, (int) q.size(), d->lb_dev));
    DD_HIP(d, hipMemcpyAsync(lb.data(), d->lb_dev, q.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s->stream));
    DD_HIP(d, hipStreamSynchronize(s->stream));
    return NBODY_OK;
}

// Sim::bbox_hook of a handle with a fitted key domain: the box of all ranks' particles.
int dd_bbox_hook(void *ctx, uint32_t *box_dev, hipStream_t stream)
{
    DD *d = static_cast<DD *>(ctx);
    const int W = d->comm.size;
    if (W == 1) return 0;
    if (d->comm.all_gather(d->comm.ctx, box_dev, d->box_all, 6 * sizeof(uint32_t), (void *) stream) != 0) return 1;
    return launch_bbox_merge(d->box_all, W, box_dev, stream) == hipSuccess ? 0 : 1;
}

// A domain without particles runs no kick-drift, but the others wait for its (neutral) box.
int empty_domain_bbox(DD *d)
{
    Sim *s = d->own;
    if (!s->keybox) return NBODY_OK;
    const uint32_t neutral[6] = {0x7f7fffffu, 0x7f7fffffu, 0x7f7fffffu, 0u, 0u, 0u};
    DD_HIP(d, hipMemcpyAsync(d->box_mine, neutral, sizeof neutral, hipMemcpyHostToDevice, s->stream));
    if (dd_bbox_hook(d, d->box_mine, s->stream) != 0) return dfail(d, NBODY_ERR_STATE, "collective failed: bounding boxes");
    // the cube every other rank draws from the merged box: keep this rank's copy in step
    DD_HIP(d, hipMemcpyAsync(reinterpret_cast<uint32_t *>(s->keybox + 5), d->box_mine, sizeof neutral, hipMemcpyDeviceToDevice, s->stream));
    DD_HIP(d, launch_keyfit_only(s));
    return NBODY_OK;
}

int ctz_edge(uint64_t j, int width)   // trailing zeros of an edge index; the ends count as coarsest
{
    if (j == 0 || j == (1ull << width)) return 64;
    return __builtin_ctzll(j);
}

}  // namespace

extern "C" {

void nbody_dd_choose_edges(const uint64_t *hist, int32_t nbins, int32_t parts, uint64_t slack,
                           int32_t *edge, int32_t *bin)
{
    std::vector<uint64_t> P((size_t) nbins + 1, 0);
    for (int j = 0; j < nbins; ++j) P[j + 1] = P[j] + hist[j];
    const uint64_t total = P[nbins];
    int width = 0;
    while ((1 << width) < nbins) ++width;
    for (int k = 1; k < parts; ++k) {
        const uint64_t target = (uint64_t) ((unsigned __int128) total * (unsigned) k / (unsigned) parts);
        int best = -1, best_tz = -1;
        uint64_t best_dist = 0;
        for (int j = 0; j <= nbins; ++j) {
            const uint64_t dist = P[j] > target ? P[j] - target : target - P[j];
            if (dist > slack) continue;
            const int tz = ctz_edge((uint64_t) j, width);
            if (tz > best_tz || (tz == best_tz && dist < best_dist)) { best = j; best_tz = tz; best_dist = dist; }
        }
        edge[k - 1] = best;
        // the bin that holds the target: P[b] <= target < P[b + 1]
        int b = (int) (std::upper_bound(P.begin(), P.end(), target) - P.begin()) - 1;
        bin[k - 1] = std::max(0, std::min(nbins - 1, b));
    }
}

int32_t nbody_dd_refine_edge(const uint64_t *hist, int32_t nbins, uint64_t base_count, uint64_t target,
                             uint64_t slack, uint64_t base_edge, int32_t edge_bits)
{
    uint64_t cum = base_count, best_dist = ~0ull;
    int best_j = 0, best_tz = -1;
    bool in_slack = false;
    for (int j = 0; j <= nbins; ++j) {
        const uint64_t dist = cum > target ? cum - target : target - cum;
        const bool ok = dist <= slack;
        const int tz = ctz_edge(base_edge + (uint64_t) j, edge_bits);
        // inside the slack the coarsest edge wins; with no edge inside it, the nearest
        if (ok ? (!in_slack || tz > best_tz || (tz == best_tz && dist < best_dist))
               : (!in_slack && dist < best_dist)) {
            best_j = j; best_dist = dist; best_tz = tz; in_slack = in_slack || ok;
        }
        if (j < nbins) cum += hist[j];
    }
    return best_j;
}

int nbody_sample_ic_range(int kind, int64_t first, int32_t count, int64_t n_total, uint64_t seed,
                          float domain_size, float gravity, float p0, float p1, float *x, float *y,
                          float *z, float *vx, float *vy, float *vz, float *m)
{
    if (count < 0 || first < 0 || n_total < 1 || first + count > n_total || !x || !y || !z || !m)
        return NBODY_ERR_INVALID;
    return sample_ic_range(kind, first, count, n_total, seed, domain_size, gravity, p0, p1, x, y, z, vx,
                           vy, vz, m);
}

int nbody_dd_create(nbody_t *own, const nbody_comm *comm, nbody_dd_t **out)
{
    if (!own || !comm || !out || comm->size < 1 || comm->size > LET_MAX_BOXES || comm->rank < 0 ||
        comm->rank >= comm->size || (comm->size > 1 && (!comm->all_gather || !comm->all_to_all)))
        return NBODY_ERR_INVALID;
    *out = nullptr;
    Sim *s = reinterpret_cast<Sim *>(own);
    DD *d = new (std::nothrow) DD();
    if (!d) return NBODY_ERR_INVALID;
    d->own = s;
    d->comm = *comm;
    if (const char *e = getenv("NBODY_DD_EARLY_PERMILLE")) d->early_permille = std::max(0, std::min(1000, atoi(e)));
    const int W = comm->size;
    Guard g(s->device);
    auto A = [&](void **p, size_t bytes) { return hipMalloc(p, std::max<size_t>(bytes, 16)); };
    hipError_t e = A((void **) &d->fr_send, (size_t) d->T * LET_REC * sizeof(float));
    if (e == hipSuccess) e = A((void **) &d->fr_all, (size_t) W * d->T * LET_REC * sizeof(float));
    if (e == hipSuccess) e = A((void **) &d->box_all, (size_t) 6 * W * sizeof(uint32_t));
    if (e == hipSuccess) e = A((void **) &d->box_mine, 6 * sizeof(uint32_t));
    if (e == hipSuccess) e = A((void **) &d->cnt_dev, (size_t) 4 * W * sizeof(int32_t));
    if (e == hipSuccess) e = hipHostMalloc((void **) &d->cnt_host, (size_t) 4 * W * sizeof(int32_t));
    if (e == hipSuccess) e = hipEventCreateWithFlags(&d->ev_counts, hipEventDisableTiming);
    // one round of histograms per cut at most: (W - 1) x 4097 queries
    const size_t nq = (size_t) std::max(1, W - 1) * (DD_BINS + 1) + W + 1;
    d->hist_words = (size_t) std::max(1, W - 1) * DD_BINS + W;
    if (e == hipSuccess) e = A((void **) &d->q_dev, nq * sizeof(uint64_t));
    if (e == hipSuccess) e = A((void **) &d->lb_dev, nq * sizeof(int32_t));
    if (e == hipSuccess) e = A((void **) &d->hist_dev, d->hist_words * sizeof(uint32_t));
    if (e == hipSuccess) e = A((void **) &d->hist_all, d->hist_words * W * sizeof(uint32_t));
    if (e != hipSuccess) { nbody_dd_destroy(reinterpret_cast<nbody_dd_t *>(d)); return NBODY_ERR_HIP; }
    if (W > 1 && ensure_export_capacity(d, (size_t) std::max(4096, s->cap / 4) * 24) != NBODY_OK) {
        nbody_dd_destroy(reinterpret_cast<nbody_dd_t *>(d));
        return NBODY_ERR_HIP;
    }
    d->splitters.assign((size_t) W + 1, 0);
    d->splitters[W] = 1ull << key_bits(s);
    if (s->keybox) {
        // Fitted key domain: all ranks must encode relative to one cube.  From here on the cube
        // is decided from the union of the ranks' bounding boxes; the cube this handle may have
        // drawn for itself is dropped, so that the first collective decision starts from the
        // same state everywhere (particles come through nbody_dd_spawn, which re-encodes them).
        if (s->spawned && W > 1) {
            nbody_dd_destroy(reinterpret_cast<nbody_dd_t *>(d));
            return NBODY_ERR_STATE;
        }
        const float zero = 0.f;
        if (hipMemcpy(s->keybox + 4, &zero, sizeof zero, hipMemcpyHostToDevice) != hipSuccess) {
            nbody_dd_destroy(reinterpret_cast<nbody_dd_t *>(d));
            return NBODY_ERR_HIP;
        }
        s->bbox_hook = dd_bbox_hook;
        s->bbox_hook_ctx = d;
    }
    // update() = build only: forces and kick are driven from here
    if (nbody_set_traversal_range(own, 0, 0) != NBODY_OK) {
        nbody_dd_destroy(reinterpret_cast<nbody_dd_t *>(d));
        return NBODY_ERR_STATE;
    }
    *out = reinterpret_cast<nbody_dd_t *>(d);
    return NBODY_OK;
}

void nbody_dd_destroy(nbody_dd_t *h)
{
    if (!h) return;
    DD *d = reinterpret_cast<DD *>(h);
    Guard g(d->own->device);
    (void) hipStreamSynchronize(d->own->stream);
    auto F = [](void *p) { if (p) (void) hipFree(p); };
    if (d->own->bbox_hook_ctx == d) { d->own->bbox_hook = nullptr; d->own->bbox_hook_ctx = nullptr; }
    F(d->box_all); F(d->box_mine);
    F(d->fr_send); F(d->fr_all); F(d->exp_buf); F(d->cnt_dev); F(d->recv_buf); F(d->q_dev); F(d->lb_dev);
    F(d->hist_dev); F(d->hist_all); F(d->mig_send); F(d->mig_recv);
    if (d->cnt_host) (void) hipHostFree(d->cnt_host);
    if (d->ev_counts) (void) hipEventDestroy(d->ev_counts);
    for (hipEvent_t e : d->pev) if (e) (void) hipEventDestroy(e);
    delete d;
}

const char *nbody_dd_last_error(nbody_dd_t *h)
{
    return h ? reinterpret_cast<DD *>(h)->err.c_str() : "null handle";
}

int nbody_dd_step(nbody_dd_t *h)
{
    if (!h) return NBODY_ERR_INVALID;
    DD *d = reinterpret_cast<DD *>(h);
    Sim *s = d->own;
    Guard g(s->device);
    const int W = d->comm.size, me = d->comm.rank;
    // A domain may be empty (more ranks than occupied key ranges): it still takes part in
    // every collective, with an empty frontier and no copies.
    const bool have = s->spawned && s->n > 0;
    const bool pt = d->phase_timing;
    const auto host_t0 = std::chrono::steady_clock::now();
    auto mark = [&](int k) -> hipError_t { return pt ? hipEventRecord(d->pev[k], s->stream) : hipSuccess; };
    d->phases_valid = false;
    DD_HIP(d, mark(0));
    if (have) DD_NB(d, nbody_update(H(s)));
    else if (W > 1) { int rce = empty_domain_bbox(d); if (rce) return rce; }
    DD_HIP(d, mark(1));
    if (W == 1) {
        if (have) {
            if (s->fp64) DD_NB(d, nbody_traverse_forest(H(s), 0, 0, -1));
            else DD_NB(d, nbody_traverse_from(H(s), H(s), 0));
            DD_NB(d, nbody_finish_step(H(s)));
        }
        d->stats.records_sent = d->stats.records_received = 0;
        d->stats.nodes_sent = d->stats.nodes_received = 0;
        return NBODY_OK;
    }
    if (have) {
        DD_NB(d, nbody_let_frontier_dev(H(s), d->fr_send, d->T));
    } else {   // empty boxes
        std::vector<float> empty((size_t) d->T * LET_REC, 0.f);
        for (int t = 0; t < d->T; ++t) empty[(size_t) t * LET_REC] = empty[(size_t) t * LET_REC + 1] = empty[(size_t) t * LET_REC + 2] = 3.0e38f;
        DD_HIP(d, hipMemcpyAsync(d->fr_send, empty.data(), empty.size() * sizeof(float), hipMemcpyHostToDevice, s->stream));
        DD_HIP(d, hipStreamSynchronize(s->stream));
    }
    DD_COMM(d, d->comm.all_gather(d->comm.ctx, d->fr_send, d->fr_all, (size_t) d->T * LET_REC * sizeof(float),
                                  (void *) s->stream));
    DD_HIP(d, mark(2));
    int rc = export_and_count(d, have);
    if (rc) return rc;
    DD_HIP(d, mark(3));
    // While the sizes travel to the host and the copies between the GPUs, the first groups are
    // walked against the own tree alone; they get the imported nodes in a pass of their own at
    // the end.  The rest of the groups -- nearly all -- see own and imported nodes in ONE walk.
    const int ngroups = have ? (s->n + s->group - 1) / s->group : 0;
    const int early = have ? std::min(ngroups, (int) ((long long) ngroups * d->early_permille / 1000)) : 0;
    if (early > 0) DD_NB(d, nbody_traverse_forest(H(s), 0, 0, early));
    DD_HIP(d, mark(4));
    const auto wait_t0 = std::chrono::steady_clock::now();
    DD_HIP(d, hipEventSynchronize(d->ev_counts));   // the step's one host wait
    d->host_wait_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wait_t0).count();
    size_t need = 0;
    for (int r = 0; r < W; ++r)
        need = std::max(need, copy_bytes(s, d->cnt_host[2 * r], d->cnt_host[2 * r + 1]));
    if (need > d->exp_cap) {
        // A copy is larger than a segment of the buffer (first steps, or the domains have
        // changed).  The sizes are the true ones either way and the peers have them already,
        // so this is a local matter: grow the buffer and export again.
        DD_HIP(d, hipStreamSynchronize(s->stream));
        rc = ensure_export_capacity(d, need + need / 4 + 4096);
        if (rc) return rc;
        DD_NB(d, nbody_let_export_tree_dev(H(s), d->fr_all, W, d->T, me, d->exp_buf, d->exp_cap, d->cnt_dev));
    }
    size_t total_recv = 0;
    int64_t rs = 0, rr = 0, ns = 0, nr = 0;
    std::vector<size_t> sb(W), so(W), rb(W), ro(W), off;
    std::vector<int32_t> cn, cr;
    for (int r = 0; r < W; ++r) {
        const bool other = r != me;
        const int32_t snn = other ? d->cnt_host[2 * r] : 0, snr = other ? d->cnt_host[2 * r + 1] : 0;
        const int32_t rnn = other ? d->cnt_host[2 * W + 2 * r] : 0, rnr = other ? d->cnt_host[2 * W + 2 * r + 1] : 0;
        sb[r] = copy_bytes(s, snn, snr);
        so[r] = (size_t) r * d->exp_cap;
        rb[r] = copy_bytes(s, rnn, rnr);
        ro[r] = total_recv;
        if (other) { off.push_back(total_recv); cn.push_back(rnn); cr.push_back(rnr); }
        total_recv += rb[r];
        rs += snr; ns += snn; rr += rnr; nr += rnn;
    }
    d->stats.records_sent = rs; d->stats.records_received = rr;
    d->stats.nodes_sent = ns; d->stats.nodes_received = nr;
    d->stats.frontier_records = d->T;
    d->stats.export_capacity = (int32_t) std::min<size_t>(d->exp_cap / 16, 0x7fffffff);
    if (total_recv > d->recv_cap) {
        if (d->recv_buf) DD_HIP(d, hipFree(d->recv_buf));   // implicit device synchronisation
        d->recv_buf = nullptr;
        d->recv_cap = 0;
        const size_t cap = total_recv + total_recv / 4 + 4096;
        DD_HIP(d, hipMalloc((void **) &d->recv_buf, cap));
        d->recv_cap = cap;
    }
    if (!d->recv_buf) {   // nothing to receive, ever: the collective still wants a pointer
        DD_HIP(d, hipMalloc((void **) &d->recv_buf, 4096));
        d->recv_cap = 4096;
    }
    d->bytes_received = total_recv;
    d->bytes_sent = 0;
    for (int r = 0; r < W; ++r) d->bytes_sent += sb[r];
    DD_HIP(d, mark(5));
    DD_COMM(d, d->comm.all_to_all(d->comm.ctx, d->exp_buf, sb.data(), so.data(), d->recv_buf, rb.data(),
                                  ro.data(), (void *) s->stream));
    DD_HIP(d, mark(6));
    if (have) {
        DD_NB(d, nbody_import_trees_dev(H(s), d->recv_buf, (int32_t) cn.size(), off.data(), cn.data(), cr.data()));
        DD_HIP(d, mark(7));
        // both passes are the last word on their groups' accelerations: the kick rides along
        // (float handles; a double handle's kick is a launch of its own after the walks)
        const bool fuse = !s->fp64;
        DD_NB(d, traverse_forest(s, 1, early, ngroups, fuse));          // own tree and copies, one walk
        if (early > 0) {
            if (s->imp_nroots > 0) DD_NB(d, traverse_forest(s, 2, 0, early, fuse));   // the copies alone, added
            else if (fuse) {   // nobody sent anything: the first groups still want their kick
                const int c = s->cur;
                DD_HIP(d, launch_vel_kick_raw(std::min(early * s->group, s->n), s->p.dt, 1.0f - s->p.velocity_dampening,
                                              s->vhx[c], s->vhy[c], s->vhz[c], s->ax, s->ay, s->az, s->vx[c], s->vy[c],
                                              s->vz[c], s->stream));
            }
        }
        if (!fuse) DD_HIP(d, launch64_vel_kick(s));
        s->pending_vel = false;
    } else {
        DD_HIP(d, mark(7));
    }
    DD_HIP(d, mark(8));
    d->host_total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host_t0).count();
    d->phases_valid = pt;
    return NBODY_OK;
}

int nbody_dd_set_phase_timing(nbody_dd_t *h, int32_t on)
{
    if (!h) return NBODY_ERR_INVALID;
    DD *d = reinterpret_cast<DD *>(h);
    Guard g(d->own->device);
    if (on && !d->pev[0])
        for (hipEvent_t &e : d->pev) DD_HIP(d, hipEventCreate(&e));
    d->phase_timing = on != 0;
    d->phases_valid = false;
    return NBODY_OK;
}

int nbody_dd_get_phases(nbody_dd_t *h, nbody_dd_phases *out)
{
    if (!h || !out) return NBODY_ERR_INVALID;
    DD *d = reinterpret_cast<DD *>(h);
    if (!d->phases_valid) return dfail(d, NBODY_ERR_STATE, "no step has been timed (nbody_dd_set_phase_timing, then nbody_dd_step)");
    Guard g(d->own->device);
    DD_HIP(d, hipEventSynchronize(d->pev[8]));
    float ms[8];
    for (int k = 0; k < 8; ++k) DD_HIP(d, hipEventElapsedTime(&ms[k], d->pev[k], d->pev[k + 1]));
    out->ms_build = ms[0]; out->ms_frontier = ms[1]; out->ms_export = ms[2]; out->ms_early_traversal = ms[3];
    out->ms_sizing = ms[4]; out->ms_exchange = ms[5]; out->ms_graft = ms[6]; out->ms_traversal = ms[7];
    DD_HIP(d, hipEventElapsedTime(&out->ms_total, d->pev[0], d->pev[8]));
    out->ms_host_wait = (float) d->host_wait_ms;
    out->ms_host_total = (float) d->host_total_ms;
    out->bytes_frontier = (int64_t) d->T * LET_REC * sizeof(float);
    out->bytes_sent = (int64_t) d->bytes_sent;
    out->bytes_received = (int64_t) d->bytes_received;
    return NBODY_OK;
}

int nbody_dd_rebalance(nbody_dd_t *h)
{
    if (!h) return NBODY_ERR_INVALID;
    DD *d = reinterpret_cast<DD *>(h);
    Sim *s = d->own;
    Guard g(s->device);
    const int W = d->comm.size, me = d->comm.rank;
    if (s->pending_vel) return dfail(d, NBODY_ERR_STATE, "rebalance between update and finish_step");
    const int KB = key_bits(s);
    d->stats.migrated_out = 0;
    d->stats.n_local = s->spawned ? s->n : 0;
    if (W == 1) return NBODY_OK;
    const int n_local = s->spawned ? s->n : 0;
    // ---- first histogram: 4096 bins of the top 12 key bits ----
    std::vector<uint64_t> q(DD_BINS + 1);
    for (int j = 0; j <= DD_BINS; ++j) q[j] = (uint64_t) j << (KB - DD_BIN_BITS);
    std::vector<int32_t> lb;
    int rc = lower_bounds(d, q, lb);
    if (rc) return rc;
    std::vector<uint32_t> local(d->hist_words, 0);
    for (int j = 0; j < DD_BINS; ++j) local[j] = (uint32_t) (lb[j + 1] - lb[j]);
    DD_HIP(d, hipMemcpyAsync(d->hist_dev, local.data(), DD_BINS * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
    std::vector<uint64_t> hist;
    rc = gather_sum(d, DD_BINS, hist);
    if (rc) return rc;
    uint64_t total = 0;
    std::vector<uint64_t> P1(DD_BINS + 1, 0);
    for (int j = 0; j < DD_BINS; ++j) { P1[j + 1] = P1[j] + hist[j]; total += hist[j]; }
    if (total == 0) return dfail(d, NBODY_ERR_STATE, "rebalance: no particles anywhere");
    const uint64_t slack = (uint64_t) (DD_SLACK * (double) total / W);
    std::vector<int32_t> edge(W - 1), bin(W - 1);
    nbody_dd_choose_edges(hist.data(), DD_BINS, W, slack, edge.data(), bin.data());
    std::vector<uint64_t> split((size_t) W + 1, 0);
    split[W] = 1ull << KB;
    struct Cut { int k; uint64_t prefix, base; };   // the bin (of `bits` key bits) that holds cut k's target
    std::vector<Cut> pending;
    for (int k = 1; k < W; ++k) {
        if (edge[k - 1] >= 0) split[k] = (uint64_t) edge[k - 1] << (KB - DD_BIN_BITS);
        else pending.push_back(Cut{k, (uint64_t) bin[k - 1], P1[bin[k - 1]]});
    }
    // ---- further rounds: the next (up to) 12 key bits of the bin each open cut falls in ----
    for (int bits = DD_BIN_BITS; !pending.empty(); ) {
        const int nb = std::min(DD_BIN_BITS, KB - bits);
        const int nbins = 1 << nb, sh = KB - bits - nb;
        const int np = (int) pending.size();
        q.assign((size_t) np * (nbins + 1), 0);
        for (int i = 0; i < np; ++i)
            for (int j = 0; j <= nbins; ++j)
                q[(size_t) i * (nbins + 1) + j] = ((pending[i].prefix << nb) + (uint64_t) j) << sh;
        rc = lower_bounds(d, q, lb);
        if (rc) return rc;
        for (int i = 0; i < np; ++i)
            for (int j = 0; j < nbins; ++j)
                local[(size_t) i * nbins + j] = (uint32_t) (lb[(size_t) i * (nbins + 1) + j + 1] - lb[(size_t) i * (nbins + 1) + j]);
        DD_HIP(d, hipMemcpyAsync(d->hist_dev, local.data(), (size_t) np * nbins * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
        rc = gather_sum(d, (size_t) np * nbins, hist);
        if (rc) return rc;
        std::vector<Cut> next;
        for (int i = 0; i < np; ++i) {
            const Cut c = pending[i];
            const uint64_t *hi = &hist[(size_t) i * nbins];
            const uint64_t target = (uint64_t) ((unsigned __int128) total * (unsigned) c.k / (unsigned) W);
            const int j = nbody_dd_refine_edge(hi, nbins, c.base, target, slack, c.prefix << nb, bits + nb);
            uint64_t cum = c.base;
            for (int t = 0; t < j; ++t) cum += hi[t];
            const uint64_t dist = cum > target ? cum - target : target - cum;
            if (dist <= slack || bits + nb == KB) {
                split[c.k] = ((c.prefix << nb) + (uint64_t) j) << sh;
            } else {   // descend into the sub-bin that holds the target
                uint64_t before = c.base;
                int b = 0;
                while (b + 1 < nbins && before + hi[b] <= target) { before += hi[b]; ++b; }
                next.push_back(Cut{c.k, (c.prefix << nb) + (uint64_t) b, before});
            }
        }
        pending.swap(next);
        bits += nb;
    }
    for (int k = 1; k <= W; ++k) split[k] = std::max(split[k], split[k - 1]);
    d->splitters = split;
    // ---- migration: the destinations of the sorted particles are contiguous stretches ----
    std::vector<uint64_t> cuts(split.begin(), split.end());
    rc = lower_bounds(d, cuts, lb);
    if (rc) return rc;
    lb[0] = 0;
    lb[W] = n_local;
    // every rank's send counts, and with them the one mass all of its particles have (0: they
    // differ or are unknown; all ones: it has no particles) -- if every rank names the same mass the
    // whole set has it, whoever holds which particle afterwards, and the handle goes on leaving
    // the mass out of its own particles' pair arithmetic (k_traverse, UNI)
    std::vector<uint32_t> sc((size_t) W + 1);
    for (int r = 0; r < W; ++r) sc[r] = (uint32_t) (lb[r + 1] - lb[r]);
    const float own_m0 = (s->fp64 || (s->p.flags & NBODY_FLAG_NO_UNIFORM_MASS)) ? 0.f : s->uniform_mass;
    sc[W] = n_local > 0 ? __builtin_bit_cast(uint32_t, own_m0) : 0xffffffffu;
    DD_HIP(d, hipMemcpyAsync(d->hist_dev, sc.data(), (W + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
    DD_COMM(d, d->comm.all_gather(d->comm.ctx, d->hist_dev, d->hist_all, (W + 1) * sizeof(uint32_t), (void *) s->stream));
    std::vector<uint32_t> mat((size_t) W * (W + 1));
    DD_HIP(d, hipMemcpyAsync(mat.data(), d->hist_all, mat.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    DD_HIP(d, hipStreamSynchronize(s->stream));
    uint32_t common = 0xffffffffu;   // bit pattern of the set's one mass, 0 = there is none
    for (int r = 0; r < W; ++r) {
        const uint32_t v = mat[(size_t) r * (W + 1) + W];
        if (v == 0xffffffffu) continue;
        common = (common == 0xffffffffu || common == v) ? v : 0u;
    }
    const float global_m0 = (common == 0xffffffffu) ? 0.f : __builtin_bit_cast(float, common);
    size_t total_recv = 0;
    const size_t mb = mig_bytes(s);
    std::vector<size_t> sb(W), so(W), rb(W), ro(W);
    for (int r = 0; r < W; ++r) {
        sb[r] = (size_t) sc[r] * mb;
        so[r] = (size_t) lb[r] * mb;
        const size_t cnt = mat[(size_t) r * (W + 1) + me];
        rb[r] = cnt * mb;
        ro[r] = total_recv * mb;
        total_recv += cnt;
    }
    // (a domain that would overflow its handle still takes part in the exchange, into a buffer
    // of the full size, and raises its error afterwards without installing anything)
    if (!d->mig_send || (size_t) n_local > d->mig_send_cap) {
        if (d->mig_send) DD_HIP(d, hipFree(d->mig_send));
        d->mig_send = nullptr;
        d->mig_send_cap = 0;
        DD_HIP(d, hipMalloc((void **) &d->mig_send, std::max<size_t>((size_t) s->cap, n_local) * mb));
        d->mig_send_cap = std::max<size_t>((size_t) s->cap, n_local);
    }
    if (!d->mig_recv || total_recv > d->mig_recv_cap) {
        if (d->mig_recv) DD_HIP(d, hipFree(d->mig_recv));
        d->mig_recv = nullptr;
        d->mig_recv_cap = 0;
        const size_t cap = std::max<size_t>((size_t) s->cap, total_recv);
        DD_HIP(d, hipMalloc((void **) &d->mig_recv, cap * mb));
        d->mig_recv_cap = cap;
    }
    if (n_local > 0) DD_HIP(d, launch_pack_particles(s, d->mig_send));
    DD_COMM(d, d->comm.all_to_all(d->comm.ctx, d->mig_send, sb.data(), so.data(), d->mig_recv, rb.data(),
                                  ro.data(), (void *) s->stream));
    d->stats.migrated_out = (int64_t) n_local - (int64_t) sc[me];
    if (total_recv > (size_t) s->cap) {
        DD_HIP(d, hipStreamSynchronize(s->stream));
        return dfail(d, NBODY_ERR_CAPACITY, "rebalance: the domain would hold more particles than num_points");
    }
    if (total_recv == 0) {
        int rce = empty_domain_bbox(d);   // the others re-encode what they received
        if (rce) return rce;
        DD_HIP(d, hipStreamSynchronize(s->stream));
        s->n = 0;
        s->spawned = false;
        d->stats.n_local = 0;
        return NBODY_OK;
    }
    DD_HIP(d, launch_unpack_particles(s, d->mig_recv, (int) total_recv));
    if (install_particles_dev(s, (int) total_recv) != NBODY_OK)
        return dfail(d, NBODY_ERR_STATE, std::string("rebalance: ") + nbody_last_error(H(s)));
    DD_NB(d, nbody_sync(H(s)));
    // (install_particles_dev forgot the mass, these particles came from other ranks: but they all
    // have the one mass the whole set has)
    if (global_m0 > 0.f) s->uniform_mass = global_m0;
    d->stats.n_local = (int64_t) total_recv;
    return NBODY_OK;
}

int nbody_dd_spawn(nbody_dd_t *h, int kind, int64_t n_total, float p0, float p1)
{
    if (!h) return NBODY_ERR_INVALID;
    DD *d = reinterpret_cast<DD *>(h);
    Sim *s = d->own;
    Guard g(s->device);
    const int W = d->comm.size, me = d->comm.rank;
    if (n_total < W) return dfail(d, NBODY_ERR_INVALID, "spawn: fewer particles than ranks");
    const int64_t first = n_total * me / W, last = n_total * (me + 1) / W;
    const int count = (int) (last - first);
    if (count > s->cap) return dfail(d, NBODY_ERR_CAPACITY, "spawn: num_points is smaller than n_total / ranks");
    int rc = spawn_range(s, kind, first, count, n_total, p0, p1);   // samples in the handle's precision
    if (rc == NBODY_ERR_INVALID) return dfail(d, rc, "spawn: unknown initial condition kind");
    if (rc) return dfail(d, rc, std::string("spawn: ") + nbody_last_error(H(s)));
    DD_NB(d, nbody_set_traversal_range(H(s), 0, 0));
    return nbody_dd_rebalance(h);
}

int nbody_dd_get_stats(nbody_dd_t *h, nbody_dd_stats *out, uint64_t *splitters)
{
    if (!h) return NBODY_ERR_INVALID;
    DD *d = reinterpret_cast<DD *>(h);
    d->stats.n_local = d->own->spawned ? d->own->n : 0;
    if (out) *out = d->stats;
    if (splitters) std::memcpy(splitters, d->splitters.data(), d->splitters.size() * sizeof(uint64_t));
    return NBODY_OK;
}

}  // extern "C"

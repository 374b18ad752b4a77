// Multi-GPU Barnes-Hut from a C++ host: one process per GPU, the library's domain
// decomposition (nbody_dd_*, include/nbody_c.h) over RCCL.  The reference is single-GPU
// (its main loop is src/main.cpp + Renderer::run, src/renderer.cpp:56-67); this is what a
// headless multi-GPU caller of the same Simulation surface looks like.
//
//   usage: dd_rccl_main <rank> <nranks> <id_file> [sphere|plummer] [particles_total] [steps]
// Rank 0 writes the RCCL unique id to <id_file>; the other ranks wait for it.  Each process
// uses GPU (rank % visible devices).  With nranks = 1 no file is needed ("-").
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include "nbody_c.h"

namespace {

struct Rccl {
    ncclComm_t comm = nullptr;
    int size = 1;
};

// nbody_comm::all_gather -- one ncclAllGather of bytes on the library's stream
int rccl_all_gather(void *ctx, const void *send_dev, void *recv_dev, size_t bytes, void *stream)
{
    Rccl *r = static_cast<Rccl *>(ctx);
    return ncclAllGather(send_dev, recv_dev, bytes, ncclChar, r->comm, (hipStream_t) stream) != ncclSuccess;
}

// nbody_comm::all_to_all -- grouped point-to-point transfers (RCCL pairs them up, self included)
int rccl_all_to_all(void *ctx, const void *send_dev, const size_t *send_bytes, const size_t *send_off,
                    void *recv_dev, const size_t *recv_bytes, const size_t *recv_off, void *stream)
{
    Rccl *r = static_cast<Rccl *>(ctx);
    bool ok = ncclGroupStart() == ncclSuccess;
    for (int p = 0; p < r->size && ok; ++p) {
        if (send_bytes[p])
            ok = ncclSend(static_cast<const char *>(send_dev) + send_off[p], send_bytes[p], ncclChar, p,
                          r->comm, (hipStream_t) stream) == ncclSuccess;
        if (ok && recv_bytes[p])
            ok = ncclRecv(static_cast<char *>(recv_dev) + recv_off[p], recv_bytes[p], ncclChar, p, r->comm,
                          (hipStream_t) stream) == ncclSuccess;
    }
    return !(ncclGroupEnd() == ncclSuccess && ok);
}

bool exchange_id(ncclUniqueId *id, int rank, const char *path)
{
    if (rank == 0) {
        if (ncclGetUniqueId(id) != ncclSuccess) return false;
        if (!strcmp(path, "-")) return true;
        std::string tmp = std::string(path) + ".tmp";
        FILE *f = fopen(tmp.c_str(), "wb");
        if (!f) return false;
        fwrite(id, sizeof *id, 1, f);
        fclose(f);
        return rename(tmp.c_str(), path) == 0;
    }
    for (int tries = 0; tries < 600; ++tries) {
        if (FILE *f = fopen(path, "rb")) {
            const size_t got = fread(id, sizeof *id, 1, f);
            fclose(f);
            if (got == 1) return true;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
    return false;
}

}  // namespace

int main(int argc, char **argv)
{
    if (argc < 4) {
        fprintf(stderr, "usage: %s <rank> <nranks> <id_file> [sphere|plummer] [particles_total] [steps]\n", argv[0]);
        return 2;
    }
    const int rank = atoi(argv[1]), nranks = atoi(argv[2]);
    const bool plummer = argc > 4 && !strcmp(argv[4], "plummer");
    const long long n_total = argc > 5 ? atoll(argv[5]) : 1 << 20;
    const int steps = argc > 6 ? atoi(argv[6]) : 5;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { fprintf(stderr, "no GPU\n"); return 1; }
    const int dev = rank % ndev;
    if (hipSetDevice(dev) != hipSuccess) return 1;

    Rccl rc;
    rc.size = nranks;
    ncclUniqueId id;
    if (!exchange_id(&id, rank, argv[3])) { fprintf(stderr, "rank %d: no RCCL id\n", rank); return 1; }
    if (ncclCommInitRank(&rc.comm, nranks, id, rank) != ncclSuccess) { fprintf(stderr, "ncclCommInitRank failed\n"); return 1; }

    nbody_params p;
    nbody_default_params(&p);
    p.num_points = (int32_t) (n_total / nranks * 5 / 4 + 4096);   // head-room for the key ranges
    p.theta = 0.6f;
    p.dt = 1e-6f;
    p.softening_factor = plummer ? 1e-3f : 0.1f;
    p.morton_bits = 21;
    p.device_id = dev;
    nbody_t *sim = nullptr;
    if (nbody_create(&p, &sim) != NBODY_OK) { fprintf(stderr, "nbody_create failed\n"); return 1; }

    nbody_comm comm;
    comm.rank = rank;
    comm.size = nranks;
    comm.ctx = &rc;
    comm.all_gather = rccl_all_gather;
    comm.all_to_all = rccl_all_to_all;
    nbody_dd_t *dd = nullptr;
    int st = nbody_dd_create(sim, &comm, &dd);
    if (st == NBODY_OK) st = nbody_dd_spawn(dd, plummer ? 3 : 1, n_total, 0.1f, 1.0f);
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < steps && st == NBODY_OK; ++i) st = nbody_dd_step(dd);
    if (st == NBODY_OK) st = nbody_sync(sim);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (st != NBODY_OK) {
        fprintf(stderr, "rank %d: status %d: %s\n", rank, st, dd ? nbody_dd_last_error(dd) : nbody_last_error(sim));
        return 1;
    }
    nbody_dd_stats s;
    nbody_dd_get_stats(dd, &s, nullptr);
    int32_t nodes = 0;
    nbody_get_num_octree_nodes(sim, &nodes);
    printf("rank=%d/%d particles=%lld octree_nodes=%d records_sent=%lld records_received=%lld ms_per_step=%.3f\n",
           rank, nranks, (long long) s.n_local, nodes, (long long) s.records_sent,
           (long long) s.records_received, ms / (steps > 0 ? steps : 1));
    nbody_dd_destroy(dd);
    nbody_destroy(sim);
    ncclCommDestroy(rc.comm);
    return 0;
}

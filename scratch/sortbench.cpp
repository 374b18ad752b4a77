// Sort microbenchmark: times nb::launch_sort on a handle's key buffer.
//   sortbench <n> <bits 10|21> <sorted 0|1>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include <algorithm>
#include "../nbody_amd/csrc/nbody_internal.hpp"
int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 1 << 21;
    const int bits = argc > 2 ? atoi(argv[2]) : 21;
    const int sorted = argc > 3 ? atoi(argv[3]) : 1;
    nbody_params p; nbody_default_params(&p);
    p.num_points = n; p.morton_bits = bits;
    nbody_t *h = nullptr;
    if (nbody_create(&p, &h) != NBODY_OK) return 1;
    nb::Sim *s = reinterpret_cast<nb::Sim *>(h);
    std::mt19937_64 rng(1);
    const int kb = 3 * bits;
    std::vector<uint64_t> k64(n);
    for (auto &k : k64) k = rng() >> (64 - kb);
    if (sorted) std::sort(k64.begin(), k64.end());
    std::vector<uint32_t> k32(n);
    for (int i = 0; i < n; ++i) k32[i] = (uint32_t) k64[i];
    const void *src = bits == 10 ? (const void *) k32.data() : (const void *) k64.data();
    const size_t kbytes = (size_t) n * s->key_bytes;
    void *d_src; hipMalloc(&d_src, kbytes); hipMemcpy(d_src, src, kbytes, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    s->n = n;
    float best = 1e9, sum = 0;
    const int reps = 30;
    for (int r = 0; r < reps + 3; ++r) {
        hipMemcpyAsync(s->keys[0], d_src, kbytes, hipMemcpyDeviceToDevice, s->stream);
        hipEventRecord(e0, s->stream);
        nb::launch_sort(s);
        hipEventRecord(e1, s->stream);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r >= 3) { best = std::min(best, ms); sum += ms; }
    }
    printf("n=%d bits=%d sorted=%d sort mean %.1f us best %.1f us (%d passes)\n", n, bits, sorted, sum / reps * 1e3, best * 1e3, bits == 10 ? 4 : 8);
    nbody_destroy(h);
    return 0;
}

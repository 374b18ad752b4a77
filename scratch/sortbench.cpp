// Sort microbenchmark: times nb::launch_sort on a handle's key buffer and checks the result
// against std::stable_sort.
//   sortbench <n> <bits 10|21> <order 0 random | 1 sorted | 2 sorted, then 1 key in 8 moved a little>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include <numeric>
#include <algorithm>
#include "../nbody_amd/csrc/nbody_internal.hpp"
int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 1 << 21;
    const int bits = argc > 2 ? atoi(argv[2]) : 21;
    const int order = argc > 3 ? atoi(argv[3]) : 1;
    nbody_params p; nbody_default_params(&p);
    p.num_points = n; p.morton_bits = bits;
    nbody_t *h = nullptr;
    if (nbody_create(&p, &h) != NBODY_OK) return 1;
    nb::Sim *s = reinterpret_cast<nb::Sim *>(h);
    std::mt19937_64 rng(1);
    const int kb = 3 * bits;
    std::vector<uint64_t> k64(n);
    // keys of a system that fills a fifth of the domain per axis: the top bits are nearly constant
    for (auto &k : k64) k = (rng() >> (64 - kb + 7)) | ((uint64_t) 0x38 << (kb - 7));
    if (order) std::sort(k64.begin(), k64.end());
    if (order == 2)
        for (int i = 0; i < n; i += 8) k64[i] ^= (rng() & 0xffffffull) << (kb > 40 ? kb - 40 - 8 : 0);
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
    // check: permutation and keys equal the stable sort's
    std::vector<uint32_t> perm(n), ref(n);
    hipMemcpy(perm.data(), s->vals[s->sorted_buf], (size_t) n * 4, hipMemcpyDeviceToHost);
    std::iota(ref.begin(), ref.end(), 0u);
    if (bits == 10) std::stable_sort(ref.begin(), ref.end(), [&](uint32_t a, uint32_t b) { return k32[a] < k32[b]; });
    else std::stable_sort(ref.begin(), ref.end(), [&](uint32_t a, uint32_t b) { return k64[a] < k64[b]; });
    size_t bad = 0;
    for (int i = 0; i < n; ++i) bad += perm[i] != ref[i];
    std::vector<uint64_t> ko(n);
    if (bits == 10) { std::vector<uint32_t> t(n); hipMemcpy(t.data(), s->keys[s->sorted_buf], kbytes, hipMemcpyDeviceToHost); for (int i = 0; i < n; ++i) ko[i] = t[i]; }
    else hipMemcpy(ko.data(), s->keys[s->sorted_buf], kbytes, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) bad += ko[i] != (bits == 10 ? (uint64_t) k32[ref[i]] : k64[ref[i]]);
    uint32_t err = 0;
    hipMemcpy(&err, &s->counts->err, 4, hipMemcpyDeviceToHost);
    printf("n=%d bits=%d order=%d sort mean %.1f us best %.1f us  mismatches %zu  dev_err %u\n", n, bits, order,
           sum / reps * 1e3, best * 1e3, bad, err);
    nbody_destroy(h);
    return bad ? 2 : 0;
}

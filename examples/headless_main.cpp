// Headless counterpart of the reference's main loop (src/main.cpp + Renderer::start_simulation
// and Renderer::run, src/renderer.cpp:56-67,295-342): construct with the renderer's default
// parameters, spawn, then update() + get_num_octree_nodes() per frame.
//   usage: headless_main [uniform|sphere|disk] [num_points] [steps] [morton_bits]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "nbody_simulation.hpp"

int main(int argc, char **argv)
{
    const char *spawner = argc > 1 ? argv[1] : "disk";
    const int n = argc > 2 ? atoi(argv[2]) : (1 << 18);
    const int steps = argc > 3 ? atoi(argv[3]) : 10;
    const int bits = argc > 4 ? atoi(argv[4]) : 21;

    Simulation<float>::Params p;
    p.num_points = n;
    p.max_num_codes_per_leaf = 32;
    p.theta = 0.75f;
    p.gravity = 1.0f;
    p.velocity_dampening = 0.0f;
    p.num_steps_validator = 0;
    p.mem_traversal_queues = (size_t) (1 << 20) * 50;
    p.traversal_group_size = 64;
    p.domain_size = 5.0f;
    if (!strcmp(spawner, "disk")) { p.dt = 1e-4f; p.softening_factor = 0.001f; }
    else if (!strcmp(spawner, "uniform")) { p.dt = 1e-6f; p.softening_factor = 0.01f; }
    else { p.dt = 1e-6f; p.softening_factor = 0.1f; }

    try {
        Simulation<float> sim(p, bits);
        if (!strcmp(spawner, "disk")) sim.spawn_disk();
        else if (!strcmp(spawner, "uniform")) sim.spawn_uniform();
        else sim.spawn_sphere();
        int nodes = sim.get_num_octree_nodes();
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < steps; ++i) {
            sim.update();
            nodes = sim.get_num_octree_nodes();  // synchronises, like the reference's D2H read
        }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        std::vector<float> x(n);
        if (nbody_get_state(sim.handle(), x.data(), nullptr, nullptr, nullptr, nullptr, nullptr,
                            nullptr, nullptr, nullptr, nullptr, nullptr) != NBODY_OK)
            return 2;
        double sx = 0;
        for (float v : x) sx += v;
        printf("spawner=%s n=%d steps=%d octree_nodes=%d ms_per_step=%.3f mean_x=%.6f\n", spawner, n,
               steps, nodes, ms / (steps > 0 ? steps : 1), sx / n);
    } catch (const std::exception &e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}

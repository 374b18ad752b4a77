// Headless counterpart of the reference's main loop (src/main.cpp + Renderer::start_simulation
// and Renderer::run, src/renderer.cpp:56-67,295-342): construct with the renderer's default
// parameters, spawn, then update() + get_num_octree_nodes() per frame.
// The reference picks float or double at build time (SIM_PRECISION, include/renderer.hpp:14);
// here it is the fifth argument.
//   usage: headless_main [uniform|sphere|disk] [num_points] [steps] [morton_bits] [float|double]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "nbody_simulation.hpp"

template <typename T>
static int run(const char *spawner, int n, int steps, int bits)
{
    typename Simulation<T>::Params p;
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
        Simulation<T> sim(p, bits);
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
        std::vector<double> x(n);   // exact for a double handle, widened for a float one
        if (nbody_get_state_f64(sim.handle(), x.data(), nullptr, nullptr, nullptr, nullptr, nullptr,
                                nullptr, nullptr, nullptr, nullptr, nullptr) != NBODY_OK)
            return 2;
        double sx = 0;
        for (double v : x) sx += v;
        printf("spawner=%s n=%d steps=%d precision=%s octree_nodes=%d ms_per_step=%.3f mean_x=%.9f\n",
               spawner, n, steps, sizeof(T) == 8 ? "double" : "float", nodes,
               ms / (steps > 0 ? steps : 1), sx / n);
    } catch (const std::exception &e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}

int main(int argc, char **argv)
{
    const char *spawner = argc > 1 ? argv[1] : "disk";
    const int n = argc > 2 ? atoi(argv[2]) : (1 << 18);
    const int steps = argc > 3 ? atoi(argv[3]) : 10;
    const int bits = argc > 4 ? atoi(argv[4]) : 21;
    const bool dbl = argc > 5 && !strcmp(argv[5], "double");
    return dbl ? run<double>(spawner, n, steps, bits) : run<float>(spawner, n, steps, bits);
}

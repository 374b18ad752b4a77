// nbody_simulation.hpp -- header-only C++ mirror of the reference's Simulation<T> class over
// the C-ABI of libnbody_mi355.so.
//
// Same public surface as the reference (include/simulation.hpp:11-47): nested Params with
// the same fields in the same order, Simulation(Params&), spawn_uniform(), spawn_sphere(),
// spawn_disk(), update(), get_num_octree_nodes(), write_validation_history().  The GL
// constructor Simulation(Params&, GLuint[7]) becomes Simulation(Params&, void*[7]): seven
// caller-owned device buffers that receive positions and octree centres of mass / sizes
// after every update; device_view() reads the library's own buffers in place.
// Errors, which the reference silently ignores, throw std::runtime_error.
// T = float (the reference's shipped SIM_PRECISION, include/renderer.hpp:14) or double
// (src/cuda/simulation.cu:255-256): Simulation<double> creates its handle with NBODY_FLAG_FP64.
#ifndef NBODY_SIMULATION_HPP
#define NBODY_SIMULATION_HPP

#include <cstddef>
#include <stdexcept>
#include <string>
#include <type_traits>

#include "nbody_c.h"

template <typename T>
class Simulation {
    static_assert(std::is_same<T, float>::value || std::is_same<T, double>::value,
                  "libnbody_mi355 implements Simulation<float> and Simulation<double>");

public:
    struct Params {
        int num_points;
        int max_num_codes_per_leaf;
        int traversal_group_size;
        size_t mem_traversal_queues;
        float theta;
        float dt;
        float gravity;
        float softening_factor;
        float domain_size;
        float velocity_dampening;
        int num_steps_validator;
    };

    // morton_bits: 10 (30-bit keys) or 21 (the reference's 63-bit keys).
    explicit Simulation(Params &params, int morton_bits = 21, int device_id = -1,
                        unsigned flags = 0)
        : Simulation(params, nullptr, morton_bits, device_id, flags)
    {
    }
    // buffers: [pos.x, pos.y, pos.z, com.x, com.y, com.z, node_size], device pointers to T
    // (include/simulation.hpp:31); see nbody_create_external for sizes.
    Simulation(Params &params, void *buffers[7], int morton_bits = 21, int device_id = -1,
               unsigned flags = 0)
        : params_(params)
    {
        nbody_params p;
        nbody_default_params(&p);
        p.num_points = params.num_points;
        p.max_num_codes_per_leaf = params.max_num_codes_per_leaf;
        p.traversal_group_size = params.traversal_group_size;
        p.mem_traversal_queues = params.mem_traversal_queues;
        p.theta = params.theta;
        p.dt = params.dt;
        p.gravity = params.gravity;
        p.softening_factor = params.softening_factor;
        p.domain_size = params.domain_size;
        p.velocity_dampening = params.velocity_dampening;
        p.num_steps_validator = params.num_steps_validator;
        p.morton_bits = morton_bits;
        p.device_id = device_id;
        p.flags = flags | (std::is_same<T, double>::value ? (unsigned) NBODY_FLAG_FP64 : 0u);
        const int rc = nbody_create_external(&p, buffers, &sim_);
        if (rc != NBODY_OK) throw std::runtime_error("nbody_create failed, status " + std::to_string(rc));
    }
    Simulation(const Simulation &) = delete;
    Simulation &operator=(const Simulation &) = delete;
    ~Simulation() { nbody_destroy(sim_); }

    void spawn_uniform() { check(nbody_spawn_uniform(sim_)); }
    void spawn_sphere() { check(nbody_spawn_sphere(sim_)); }
    void spawn_disk() { check(nbody_spawn_disk(sim_)); }
    void update() { check(nbody_update(sim_)); }
    int get_num_octree_nodes()
    {
        int32_t n = 0;
        check(nbody_get_num_octree_nodes(sim_, &n));
        return n;
    }
    void write_validation_history(const std::string &csv_file_path)
    {
        check(nbody_write_validation_history(sim_, csv_file_path.c_str()));
    }

    // additions over the reference
    nbody_t *handle() { return sim_; }
    // the library's own arrays in place: nbody_dev_view for float, nbody_dev_view_f64 for double
    typename std::conditional<std::is_same<T, double>::value, nbody_dev_view_f64, nbody_dev_view>::type
    device_view()
    {
        if constexpr (std::is_same<T, double>::value) {
            nbody_dev_view_f64 v;
            check(nbody_get_device_ptrs_f64(sim_, &v));
            return v;
        } else {
            nbody_dev_view v;
            check(nbody_get_device_ptrs(sim_, &v));
            return v;
        }
    }

private:
    void check(int rc)
    {
        if (rc != NBODY_OK)
            throw std::runtime_error(std::string("nbody status ") + std::to_string(rc) + ": " +
                                     nbody_last_error(sim_));
    }
    nbody_t *sim_ = nullptr;
    Params params_;
};

#endif

// k_reduce.hip -- on-device reductions of the validation path: system energy and the
// relative acceleration error, and the stateless direct-sum energy of a particle set.
//
// Reference behaviour restated: compute_conserved_quantities + compute_acc_error write one
// value per body and three cub::DeviceReduce::Sum calls reduce them on the GPU
// (src/cuda/validator.cu:38-132,257-287).  Here a kernel reduces while it computes: every
// workgroup leaves one partial sum per quantity (double) in a scratch array, and the
// workgroup that finishes last adds the partials in index order, so the result does not
// depend on the order in which workgroups ran.  Only the final doubles travel to the host.
#include <cstdio>

#include "dev_utils.hpp"
#include "nbody_internal.hpp"

namespace nb {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Workgroup of 256 threads: sums K per-thread doubles, stores the K partials of this
// workgroup, and lets the last workgroup to arrive add all partials in index order into
// out[0..K).  scratch: partials [gridDim.x][K] doubles after a 256-byte header whose first
// word is the arrival counter (left at zero for the next launch).
template <int K>
__device__ __forceinline__ void finish_sums(double (&acc)[K], unsigned char *scratch, double *out)
{
    __shared__ double s_part[4][K];
    __shared__ bool s_last;
    uint32_t *counter = reinterpret_cast<uint32_t *>(scratch);
    double *partials = reinterpret_cast<double *>(scratch + REDUCE_HEADER_BYTES);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const double v = wave_sum_f64(acc[k]);
        if (lane == 0) s_part[w][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < K) {
        const double v = s_part[0][threadIdx.x] + s_part[1][threadIdx.x] + s_part[2][threadIdx.x] +
                         s_part[3][threadIdx.x];
        __hip_atomic_store(&partials[(size_t) blockIdx.x * K + threadIdx.x], v, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // release: this workgroup's partials are visible device-wide before its arrival is
        const uint32_t arrived = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL,
                                                        __HIP_MEMORY_SCOPE_AGENT);
        s_last = arrived == gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    // one wave adds the partials: lane l takes workgroups l, l+64, ... in ascending order,
    // then the lanes are combined in a fixed butterfly
    if (w == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double v = 0.0;
            for (uint32_t b = lane; b < gridDim.x; b += 64)
                v += __hip_atomic_load(&partials[(size_t) b * K + k], __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            v = wave_sum_f64(v);
            if (lane == 0) out[k] = v;
        }
        if (lane == 0)
            __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// out[0] = sum_i m_i v_i^2 / 2, out[1] = -G/2 sum_i m_i pot_i, out[2] = sum_i |a_i - d_i| / |d_i|
// (bodies with |d_i| = 0 skipped; a == nullptr: out[2] = 0).  m is read with stride m_stride
// (1 for a mass array, 4 for the w component of packed (x,y,z,m) records).
template <typename R>
__global__ __launch_bounds__(256) void k_energy_reduce(
    int n, const R *__restrict__ m, int m_stride, const R *__restrict__ vx,
    const R *__restrict__ vy, const R *__restrict__ vz, const R *__restrict__ pot, double G,
    const R *__restrict__ ax, const R *__restrict__ ay, const R *__restrict__ az,
    const R *__restrict__ dx, const R *__restrict__ dy, const R *__restrict__ dz,
    unsigned char *scratch, double *out)
{
    double acc[3] = {0.0, 0.0, 0.0};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const double mi = (double) m[(size_t) i * m_stride];
        const double ux = (double) vx[i], uy = (double) vy[i], uz = (double) vz[i];
        acc[0] += 0.5 * mi * (ux * ux + uy * uy + uz * uz);
        acc[1] -= 0.5 * G * mi * (double) pot[i];
        if (ax) {
            const double ex = (double) ax[i] - (double) dx[i], ey = (double) ay[i] - (double) dy[i],
                         ez = (double) az[i] - (double) dz[i];
            const double dn = sqrt((double) dx[i] * dx[i] + (double) dy[i] * dy[i] + (double) dz[i] * dz[i]);
            if (dn > 0.0) acc[2] += sqrt(ex * ex + ey * ey + ez * ez) / dn;
        }
    }
    finish_sums<3>(acc, scratch, out);
}

// Energy of n bodies by direct summation in one kernel: 256 bodies per workgroup, sources
// staged through LDS 256 at a time in the pair layout, pot_i = sum_{j != i} m_j / sqrt(d^2 +
// eps^2) in float like the force kernels (the j = i term is masked out in the one tile
// that holds it), then out[0] = sum m v^2 / 2 and out[1] = -G/2 sum m_i pot_i in double.
__global__ __launch_bounds__(256) void k_direct_energy(const Float4 *__restrict__ posm,
                                                       const float *__restrict__ vx,
                                                       const float *__restrict__ vy,
                                                       const float *__restrict__ vz, int n,
                                                       float eps2, float gravity,
                                                       unsigned char *scratch, double *out)
{
    __shared__ v4f s_xy[128];
    __shared__ v4f s_zm[128];
    float *xyf = reinterpret_cast<float *>(s_xy);
    float *zmf = reinterpret_cast<float *>(s_zm);
    const int i = blockIdx.x * 256 + threadIdx.x;
    const Float4 me = posm[i < n ? i : 0];
    const v2f px2 = {me.x, me.x}, py2 = {me.y, me.y}, pz2 = {me.z, me.z}, e2 = {eps2, eps2};
    v2f ps2 = {0.f, 0.f};
    for (int t0 = 0; t0 < n; t0 += 256) {
        const int c = min(256, n - t0);
        {
            const int t = (int) threadIdx.x;
            const Float4 s = t < c ? posm[t0 + t] : Float4{0.f, 0.f, 0.f, 0.f};   // massless pad
            const int p = t >> 1, h = t & 1;
            xyf[4 * p + h] = s.x; xyf[4 * p + 2 + h] = s.y;
            zmf[4 * p + h] = s.z; zmf[4 * p + 2 + h] = s.w;
        }
        __syncthreads();
        const int np2 = (c + 1) >> 1;
        auto pair_pot = [&](int j, v2f mass_mask) {
            const v4f a = s_xy[j], b = s_zm[j];
            const v2f ddx = a.xy - px2, ddy = a.zw - py2, ddz = b.xy - pz2;
            const v2f t = __builtin_elementwise_fma(
                ddz, ddz, __builtin_elementwise_fma(ddy, ddy, __builtin_elementwise_fma(ddx, ddx, e2)));
            const v2f r = {__builtin_amdgcn_rsqf(t.x), __builtin_amdgcn_rsqf(t.y)};
            ps2 = __builtin_elementwise_fma(b.zw * mass_mask, r, ps2);
        };
        if (t0 == (int) blockIdx.x * 256) {
            // the tile that holds this workgroup's own bodies: body i skips source i
            const int mine = (int) threadIdx.x >> 1, half = (int) threadIdx.x & 1;
            for (int j = 0; j < np2; ++j) {
                v2f mask = {1.f, 1.f};
                if (j == mine) mask[half] = 0.f;
                pair_pot(j, mask);
            }
        } else {
            const v2f one = {1.f, 1.f};
#pragma unroll 8
            for (int j = 0; j < np2; ++j) pair_pot(j, one);
        }
        __syncthreads();
    }
    double acc[2] = {0.0, 0.0};
    if (i < n) {
        const float ps = ps2.x + ps2.y;
        const double ux = (double) vx[i], uy = (double) vy[i], uz = (double) vz[i];
        acc[0] = 0.5 * (double) me.w * (ux * ux + uy * uy + uz * uz);
        acc[1] = -0.5 * (double) gravity * (double) me.w * (double) ps;
    }
    finish_sums<2>(acc, scratch, out);
}

size_t reduce_scratch_bytes(int blocks, int k)
{
    return REDUCE_HEADER_BYTES + (size_t) blocks * k * sizeof(double);
}

static inline int reduce_grid(int n) { return std::max(1, std::min(REDUCE_MAX_BLOCKS, (n + 255) / 256)); }

template <typename R>
static hipError_t energy_reduce_impl(int n, const R *m, int m_stride, const R *vx, const R *vy, const R *vz,
                                     const R *pot, double G, const R *const a[3], const R *const d[3],
                                     unsigned char *scratch, double *out, hipStream_t st)
{
    hipLaunchKernelGGL(k_energy_reduce<R>, dim3(reduce_grid(n)), dim3(256), 0, st, n, m, m_stride, vx, vy,
                       vz, pot, G, a ? a[0] : nullptr, a ? a[1] : nullptr, a ? a[2] : nullptr,
                       d ? d[0] : nullptr, d ? d[1] : nullptr, d ? d[2] : nullptr, scratch, out);
    return hipGetLastError();
}

hipError_t launch_energy_reduce(int n, const float *m, int m_stride, const float *vx, const float *vy,
                                const float *vz, const float *pot, double G, const float *const a[3],
                                const float *const d[3], unsigned char *scratch, double *out,
                                hipStream_t st)
{
    return energy_reduce_impl<float>(n, m, m_stride, vx, vy, vz, pot, G, a, d, scratch, out, st);
}

hipError_t launch_energy_reduce64(int n, const double *m, int m_stride, const double *vx, const double *vy,
                                  const double *vz, const double *pot, double G, const double *const a[3],
                                  const double *const d[3], unsigned char *scratch, double *out,
                                  hipStream_t st)
{
    return energy_reduce_impl<double>(n, m, m_stride, vx, vy, vz, pot, G, a, d, scratch, out, st);
}

// scratch must hold reduce_scratch_bytes((n + 255) / 256, 2) bytes with a zero first word.
hipError_t launch_direct_energy(const Float4 *posm, const float *vx, const float *vy, const float *vz,
                                int n, float eps2, float gravity, unsigned char *scratch, double *out,
                                hipStream_t st)
{
    hipLaunchKernelGGL(k_direct_energy, dim3((n + 255) / 256), dim3(256), 0, st, posm, vx, vy, vz, n,
                       eps2, gravity, scratch, out);
    return hipGetLastError();
}

}  // namespace nb

// ---------------------------------------------------------------------------
// Stateless C-ABI entry points: total energy of a particle set by direct summation.
// ---------------------------------------------------------------------------
#include <mutex>

namespace {

constexpr int NOALLOC_MAX_BODIES = 4194304;
constexpr int NOALLOC_MAX_BLOCKS = NOALLOC_MAX_BODIES / 256;
constexpr int NOALLOC_MAX_DEVICES = 64;

// Scratch of nbody_direct_energy_dev_noalloc: arrival counter + one (kinetic, potential)
// partial per workgroup.  Every GPU the library runs on holds its own copy of this array for
// the life of the process; host code reaches it only through the address hipGetSymbolAddress
// returns for the current device (the name below is not a device address).
static __device__ __align__(256) unsigned char scratch[nb::REDUCE_HEADER_BYTES + NOALLOC_MAX_BLOCKS * 2 * sizeof(double)];

std::mutex g_scratch_mutex;
unsigned char *g_scratch_ptr[NOALLOC_MAX_DEVICES] = {};

// Address of the current device's copy of `scratch` (looked up once per device).
hipError_t scratch_for_current_device(unsigned char **out)
{
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= NOALLOC_MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void *ptr = nullptr;
        e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!ptr) return hipErrorInvalidSymbol;
        g_scratch_ptr[dev] = static_cast<unsigned char *>(ptr);
    }
    *out = g_scratch_ptr[dev];
    return hipSuccess;
}

}  // namespace

// Looks the scratch address of the current device up ahead of time (nbody_create does, for its
// device), so that no later nbody_direct_energy_dev_noalloc call on that device has anything
// left to resolve -- a call under stream capture must not.
namespace nb {
void noalloc_prepare_current_device()
{
    unsigned char *p = nullptr;
    (void) scratch_for_current_device(&p);
}
}  // namespace nb

namespace {

bool energy_args_ok(const float *posm, const float *vx, const float *vy, const float *vz, int32_t n,
                    float softening_factor, double *out_dev)
{
    return posm && vx && vy && vz && out_dev && n >= 1 && softening_factor > 0.0f;
}

}  // namespace

extern "C" {

int nbody_direct_energy_dev(const float *posm, const float *vx, const float *vy, const float *vz,
                            int32_t n, float softening_factor, float gravity, double *out_dev,
                            void *stream)
{
    if (!energy_args_ok(posm, vx, vy, vz, n, softening_factor, out_dev)) return NBODY_ERR_INVALID;
    hipStream_t st = (hipStream_t) stream;
    unsigned char *buf = nullptr;
    if (hipMalloc((void **) &buf, nb::reduce_scratch_bytes((n + 255) / 256, 2)) != hipSuccess)
        return NBODY_ERR_HIP;
    hipError_t e = hipMemsetAsync(buf, 0, nb::REDUCE_HEADER_BYTES, st);
    if (e == hipSuccess)
        e = nb::launch_direct_energy(reinterpret_cast<const nb::Float4 *>(posm), vx, vy, vz, n,
                                     softening_factor * softening_factor, gravity, buf, out_dev, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);   // the scratch is freed below
    (void) hipFree(buf);
    return e == hipSuccess ? NBODY_OK : NBODY_ERR_HIP;
}

int nbody_direct_energy_dev_noalloc(const float *posm, const float *vx, const float *vy,
                                    const float *vz, int32_t n, float softening_factor,
                                    float gravity, double *out_dev, void *stream)
{
    if (!energy_args_ok(posm, vx, vy, vz, n, softening_factor, out_dev)) return NBODY_ERR_INVALID;
    if (n > NOALLOC_MAX_BODIES) {
        fprintf(stderr, "nbody_direct_energy_dev_noalloc: %d bodies exceed the fixed scratch (max %d); "
                        "use nbody_direct_energy_dev\n", n, NOALLOC_MAX_BODIES);
        return NBODY_ERR_CAPACITY;
    }
    unsigned char *ptr = nullptr;
    if (scratch_for_current_device(&ptr) != hipSuccess) return NBODY_ERR_HIP;
    hipStream_t st = (hipStream_t) stream;
    // the arrival counter must start at zero whatever an earlier, possibly failed, call left
    hipError_t e = hipMemsetAsync(ptr, 0, nb::REDUCE_HEADER_BYTES, st);
    if (e == hipSuccess)
        e = nb::launch_direct_energy(reinterpret_cast<const nb::Float4 *>(posm), vx, vy, vz, n,
                                     softening_factor * softening_factor, gravity, ptr, out_dev, st);
    return e == hipSuccess ? NBODY_OK : NBODY_ERR_HIP;
}

}  // extern "C"

// k_let.hip -- kernels of the multi-GPU locally-essential-tree exchange: bounding box of a
// handle's particles, export of the sources a remote domain needs from this handle's tree,
// and unpacking of received (x,y,z,m) records into a source-only handle.
//
// The reference is single-GPU and has none of this; the scheme follows the hybrid
// distributed-octree idea its README cites (Cazalbou et al., PASC'24): every GPU owns a
// domain, builds its own tree, and ships to every other domain only the part of that tree the
// other domain's bodies can ever touch.
#include "dev_utils.hpp"
#include "nbody_internal.hpp"

namespace nb {

// positions are >= 0 (reflecting walls), so float order == unsigned order of the bit patterns
__global__ __launch_bounds__(256) void k_bbox(int n, const Float4 *__restrict__ posm,
                                              uint32_t *__restrict__ box /* lo xyz, hi xyz */)
{
    __shared__ float s[6][4];
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {0.f, 0.f, 0.f};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const Float4 p = posm[i];
        lo[0] = fminf(lo[0], p.x); lo[1] = fminf(lo[1], p.y); lo[2] = fminf(lo[2], p.z);
        hi[0] = fmaxf(hi[0], p.x); hi[1] = fmaxf(hi[1], p.y); hi[2] = fmaxf(hi[2], p.z);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], o, 64));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], o, 64));
        }
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 3; ++k) { s[k][w] = lo[k]; s[3 + k][w] = hi[k]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        float v = s[k][0];
        for (int i = 1; i < 4; ++i) v = k < 3 ? fminf(v, s[k][i]) : fmaxf(v, s[k][i]);
        if (k < 3) atomicMin(&box[k], __float_as_uint(v)); else atomicMax(&box[k], __float_as_uint(v));
    }
}

__device__ __forceinline__ float box_dist_sq(const float *__restrict__ bx, float cx, float cy,
                                             float cz)
{
    const float dx = fmaxf(0.f, fmaxf(bx[0] - cx, cx - bx[3]));
    const float dy = fmaxf(0.f, fmaxf(bx[1] - cy, cy - bx[4]));
    const float dz = fmaxf(0.f, fmaxf(bx[2] - cz, cz - bx[5]));
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

// One thread per (octree node, remote box).  A node is REACHED from the root iff none of its
// ancestors passes the opening criterion with respect to the box, where the criterion is the
// traversal's own  size^2 / d^2 < theta^2  with d = distance from the BOX to the node's centre
// of mass -- never larger than the distance from any body inside the box, so whatever is
// accepted here would be accepted by every group of the remote domain as well.  A reached
// node exports its (com, mass) if it passes, all its particles if it is a leaf, and nothing
// otherwise (its children decide).  No stack, no queue, no inter-thread dependency; output
// slots come from one atomic per workgroup and box.
__global__ __launch_bounds__(256) void k_let_export(
    const Counts *counts, const Float4 *__restrict__ nodeA, const NodeB *__restrict__ nodeB,
    const int32_t *__restrict__ o_parent, const Float4 *__restrict__ posm,
    const float *__restrict__ boxes, float theta, Float4 *__restrict__ out, int cap,
    uint32_t *__restrict__ out_count)
{
    __shared__ uint32_t s_scan[5];
    __shared__ uint32_t s_base;
    const int n_oct = counts->n_oct;
    const int box = blockIdx.y;
    const float *bx = boxes + 6 * box;
    const float th2 = theta * theta;
    const int o = blockIdx.x * 256 + threadIdx.x;
    auto passes = [&](int a) -> bool {
        const Float4 A = nodeA[a];
        const float sz = nodeB[a].size;
        return __fdiv_rn(sz * sz, box_dist_sq(bx, A.x, A.y, A.z)) < th2;
    };
    uint32_t cnt = 0;
    bool leaf = false, whole = false;
    NodeB me{};
    if (o < n_oct) {
        bool reached = true;
        for (int a = o_parent[o]; a >= 0; a = o_parent[a])
            if (passes(a)) { reached = false; break; }
        if (reached) {
            me = nodeB[o];
            leaf = (me.meta & NODE_LEAF) != 0u;
            if (leaf) cnt = (uint32_t) me.count;
            else if (passes(o)) { whole = true; cnt = 1; }
        }
    }
    uint32_t tot;
    const uint32_t excl = block_excl_scan_u32<4>(cnt, s_scan, &tot);
    if (threadIdx.x == 0) s_base = tot ? atomicAdd(&out_count[box], tot) : 0u;
    __syncthreads();
    if (cnt == 0) return;
    const size_t pos = (size_t) s_base + excl;
    if (pos + cnt > (size_t) cap) return;  // over capacity: the host sees out_count > cap
    Float4 *dst = out + (size_t) box * cap + pos;
    if (whole) dst[0] = nodeA[o];
    else
        for (int k = 0; k < me.count; ++k) dst[k] = posm[me.first + k];
}

// (x,y,z,m) records -> the SoA arrays of a source-only handle.
__global__ __launch_bounds__(256) void k_unpack_sources(int n, const Float4 *__restrict__ src,
                                                        float *__restrict__ x, float *__restrict__ y,
                                                        float *__restrict__ z, float *__restrict__ m,
                                                        uint32_t *__restrict__ id)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const Float4 p = src[i];
    x[i] = p.x; y[i] = p.y; z[i] = p.z; m[i] = p.w;
    id[i] = (uint32_t) i;
}

hipError_t launch_bbox(Sim *s, uint32_t *box_dev)
{
    hipLaunchKernelGGL(k_bbox, dim3(std::min((s->n + 255) / 256, 1024)), dim3(256), 0, s->stream,
                       s->n, s->posm, box_dev);
    return hipGetLastError();
}

hipError_t launch_let_export(Sim *s, const float *boxes_dev, int nboxes, Float4 *out, int cap,
                             uint32_t *out_count)
{
    // grid.x from the host-known upper bound of the node count (2n); surplus threads exit
    const int gx = (2 * s->n + 255) / 256;
    hipLaunchKernelGGL(k_let_export, dim3(gx, nboxes), dim3(256), 0, s->stream, s->counts,
                       s->nodeA, s->nodeB, s->o_parent, s->posm, boxes_dev, s->p.theta, out, cap,
                       out_count);
    return hipGetLastError();
}

hipError_t launch_unpack_sources(Sim *s, const Float4 *src, int n)
{
    const int c = s->cur;
    hipLaunchKernelGGL(k_unpack_sources, dim3((n + 255) / 256), dim3(256), 0, s->stream, n, src,
                       s->x[c], s->y[c], s->z[c], s->m[c], s->id[c]);
    return hipGetLastError();
}

}  // namespace nb

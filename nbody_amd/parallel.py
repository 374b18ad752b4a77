"""Multi-GPU choreography: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on MI355X; "gloo" in the CPU tests).  The reference is single-GPU; these protocols are
the two exchange steps the step path has when it is spread over the GPUs of a node:

* ShardedTraversal -- Barnes-Hut step.  Every rank holds all particles and builds the same
  tree (sort + build are a few % of a step); the traversal, ~90 % of a step, is split into
  Morton-contiguous ranges of groups, one per rank; ONE all-gather per acceleration
  component per step returns the full acceleration arrays to everybody, then each rank
  applies the velocity kick to all particles.  Results are bit-identical to a single-GPU
  step because tree, groups and per-group arithmetic are the same.
* ring_direct_sum -- O(N^2) validation path.  Bodies are block-partitioned; the (x,y,z,m)
  tiles travel round a ring (P-1 send/recv steps, double buffered: the next tile is in
  flight while the current one is being summed), each rank accumulating the accelerations of
  its own bodies.  No reduction is needed.

The compute itself is injected (`backend` / `tile_kernel`), so the same choreography runs on
GPUs through libnbody_mi355.so and in the CPU tests through the host oracle.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def split_groups(ngroups: int, world: int, rank: int):
    """Contiguous, equal-sized (ceil) ranges of groups; trailing ranks may get fewer/none."""
    chunk = (ngroups + world - 1) // world
    g0 = min(rank * chunk, ngroups)
    g1 = min(g0 + chunk, ngroups)
    return g0, g1, chunk


class ShardedTraversal:
    """backend must provide:
        n, group                      particle count, bodies per group
        set_traversal_range(g0, g1)   restrict update()'s traversal
        update()                      kick-drift, sort, build, traversal of the own range
        wait()                        own accelerations are complete and visible to torch
        acc_tensors()                 3 torch tensors (ax, ay, az) of n floats, zero-copy views
        finish_step()                 velocity kick for all particles
    """

    def __init__(self, backend, process_group=None):
        self.b = backend
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        n, g = backend.n, backend.group
        ngroups = (n + g - 1) // g
        self.g0, self.g1, chunk = split_groups(ngroups, self.world, self.rank)
        self.chunk_bodies = chunk * g
        self.even = self.chunk_bodies * self.world == n
        backend.set_traversal_range(self.g0, self.g1)
        self._pad = None

    def step(self):
        b = self.b
        b.update()
        b.wait()
        acc = b.acc_tensors()
        lo = self.g0 * b.group
        if self.even:
            for a in acc:  # in place: every rank's slice sits at rank * chunk_bodies
                dist.all_gather_into_tensor(a, a[lo:lo + self.chunk_bodies], group=self.pg)
        else:
            hi = min(self.g1 * b.group, b.n)
            if self._pad is None:
                self._pad = (torch.zeros(self.chunk_bodies * self.world, dtype=acc[0].dtype,
                                         device=acc[0].device),
                             torch.zeros(self.chunk_bodies, dtype=acc[0].dtype, device=acc[0].device))
            full, mine = self._pad
            for a in acc:
                mine.zero_()
                mine[:hi - lo] = a[lo:hi]
                dist.all_gather_into_tensor(full, mine, group=self.pg)
                a.copy_(full[:b.n])
        self._after_collective(acc[0])
        b.finish_step()

    @staticmethod
    def _after_collective(t):
        if t.is_cuda:  # the library works on its own stream: finish RCCL before the kick
            torch.cuda.current_stream(t.device).synchronize()


def ring_direct_sum(local_posm: torch.Tensor, tile_kernel, process_group=None):
    """local_posm: (n_local, 4) float32 tensor of this rank's bodies (x, y, z, m); all ranks
    hold the same n_local.  tile_kernel(targets, sources, accumulate) adds the accelerations
    exerted by `sources` on `targets` into its own output (accumulate=False on the first
    call).  After P tiles every rank has the full direct sum for its bodies."""
    world = dist.get_world_size(process_group)
    rank = dist.get_rank(process_group)
    local_posm = local_posm.contiguous()
    bufs = [torch.empty_like(local_posm), torch.empty_like(local_posm)]
    to_global = (lambda r: dist.get_global_rank(process_group, r)) if process_group else (lambda r: r)
    right, left = to_global((rank + 1) % world), to_global((rank - 1) % world)
    cur = local_posm
    for step in range(world):
        reqs, dst = [], None
        if step + 1 < world:  # pass the tile on while it is being summed
            dst = bufs[step % 2]
            reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend, cur, right, group=process_group),
                                           dist.P2POp(dist.irecv, dst, left, group=process_group)])
        tile_kernel(local_posm, cur, step > 0)
        for r in reqs:
            r.wait()
        if dst is not None:
            if cur.is_cuda:  # tile_kernel must have finished with `cur` before it is reused
                torch.cuda.current_stream(cur.device).synchronize()
            cur = dst


def gpu_tile_kernel(lib, out_ax, out_ay, out_az, softening_factor, gravity):
    """tile_kernel for ring_direct_sum over libnbody_mi355.so: launches nbody_direct_sum_dev
    on torch's current stream; out_* are torch float32 tensors of n_local elements."""
    def kernel(targets, sources, accumulate):
        st = torch.cuda.current_stream(targets.device).cuda_stream
        rc = lib.nbody_direct_sum_dev(targets.data_ptr(), targets.shape[0], sources.data_ptr(),
                                      sources.shape[0], softening_factor, gravity,
                                      out_ax.data_ptr(), out_ay.data_ptr(), out_az.data_ptr(),
                                      1 if accumulate else 0, st)
        if rc != 0:
            raise RuntimeError(f"nbody_direct_sum_dev failed with status {rc}")
    return kernel


class GpuBackend:
    """ShardedTraversal backend over a nbody_amd.Simulation (all particles on every rank)."""

    class _Dev:
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False),
                                             "version": 2}

    def __init__(self, sim, device):
        self.sim = sim
        self.n = sim.params.num_points
        self.group = sim.params.traversal_group_size
        v = sim.device_view()
        self._acc = [torch.as_tensor(self._Dev(p, self.n), device=device) for p in (v.ax, v.ay, v.az)]

    def set_traversal_range(self, g0, g1):
        self.sim.set_traversal_range(g0, g1)

    def update(self):
        self.sim.update()

    def wait(self):
        self.sim.sync()

    def acc_tensors(self):
        return self._acc

    def finish_step(self):
        self.sim.finish_step()

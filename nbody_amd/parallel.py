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

* LetStepper -- Barnes-Hut step with DOMAIN DECOMPOSITION.  Every rank owns a fixed subset of
  the particles (one box of an orthogonal recursive bisection of the initial conditions), builds a tree over
  them only and traverses it; what its bodies need from the other domains arrives as a
  locally-essential tree: an all-gather of the P bounding boxes (24 bytes each), then ONE
  all-to-all of (x,y,z,m) records -- the centres of mass of remote nodes the whole box may
  approximate, and the particles of remote leaves it may not.  A second tree is built over
  the received records and traversed for the own bodies.  Memory, build and traversal all
  scale with N/P; the exchanged volume is a surface term.

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


class LetStepper:
    """backend must provide:
        update()                      kick-drift, sort, build of the own tree (no forces yet)
        bbox()                        6 floats (lo xyz, hi xyz) of the own particles
        let_export(boxes)             boxes: (k, 6) float32 array -> list of k (cnt_i, 4)
                                      float32 torch tensors on the backend's device
        begin_local_forces()          start the own-tree traversal; may return before it is
                                      done, so that it runs while the records are exchanged
        add_remote(records)           (cnt, 4) tensor: build a tree over it, add its attraction
        finish_step()                 velocity kick
        device                        torch device of the exchanged tensors
    """

    def __init__(self, backend, process_group=None):
        self.b = backend
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.last_sent = 0
        self.last_received = 0

    def step(self):
        b, W, me = self.b, self.world, self.rank
        b.update()
        dev = b.device
        box = torch.as_tensor(b.bbox(), dtype=torch.float32).to(dev)
        boxes = torch.empty(W * 6, dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(boxes, box, group=self.pg)
        boxes = boxes.view(W, 6)
        others = [r for r in range(W) if r != me]
        recs = b.let_export(boxes.cpu().numpy()[others]) if others else []
        b.begin_local_forces()
        send_rows = [0] * W
        for r, t in zip(others, recs):
            send_rows[r] = int(t.shape[0])
        sc = torch.tensor(send_rows, dtype=torch.int64, device=dev)
        rc = torch.empty_like(sc)
        dist.all_to_all_single(rc, sc, group=self.pg)
        recv_rows = [int(v) for v in rc.cpu()]
        send = torch.cat(recs) if recs else torch.empty(0, 4, dtype=torch.float32, device=dev)
        recv = torch.empty(sum(recv_rows), 4, dtype=torch.float32, device=dev)
        dist.all_to_all_single(recv, send.contiguous(), recv_rows, send_rows, group=self.pg)
        if recv.is_cuda:  # the library works on its own stream: finish RCCL before the build
            torch.cuda.current_stream(recv.device).synchronize()
        self.last_sent, self.last_received = sum(send_rows), sum(recv_rows)
        b.add_remote(recv)
        b.finish_step()

    def rebalance(self, sample=4096, seed=0, weighted=False):
        """Re-draws the domains and migrates particles (call between steps, every few hundred:
        it goes through host memory).  Every rank contributes `sample` of its positions; all
        ranks bisect the gathered sample identically (orb_cuts), classify their particles and
        exchange them with one all-to-all.  Requires backend.get_particles() -> dict of host
        arrays (x,y,z,vx,vy,vz,m,ax,ay,az float32; id uint32) and backend.set_particles(dict).
        weighted: the cuts equalise WORK instead of particle counts when the backend has
        work_weights() -> one float per particle (order of get_particles()): the number of
        sources its group interacted with in the last force evaluation, own and imported tree.
        A sampled particle stands for n/sample particles of its rank, so its weight is scaled
        by that.  Off by default: on the sphere and Plummer workloads equal counts already
        balance the interaction counts to 3 %, less than the noise of a sampled weighting.
        Returns the number of particles this rank gave away."""
        import numpy as np
        b, W, me = self.b, self.world, self.rank
        work = b.work_weights() if weighted and hasattr(b, "work_weights") else None
        st = b.get_particles()
        n = len(st["x"])
        pos = np.stack([st["x"], st["y"], st["z"]], 1)
        rng = np.random.default_rng(seed * 1000003 + me)
        pick = rng.choice(n, size=sample, replace=n < sample)
        rec4 = np.empty((sample, 4), np.float32)
        rec4[:, :3] = pos[pick]
        rec4[:, 3] = (np.ones(sample) if work is None else np.asarray(work, np.float64)[pick]) * (n / sample)
        mine = torch.from_numpy(rec4.reshape(-1))
        allp = torch.empty(W * sample * 4, dtype=torch.float32)
        if b.device.type == "cuda":     # RCCL moves device memory only
            g = allp.to(b.device)
            dist.all_gather_into_tensor(g, mine.to(b.device), group=self.pg)
            allp = g.cpu()
        else:
            dist.all_gather_into_tensor(allp, mine, group=self.pg)
        allp = allp.numpy().reshape(-1, 4)
        cuts = orb_cuts(allp[:, :3], W, allp[:, 3].astype(np.float64), grid=getattr(b, "grid", None))
        dest = orb_assign(cuts, pos)
        order = np.argsort(dest, kind="stable")
        send_rows = np.bincount(dest, minlength=W).tolist()
        names = ("x", "y", "z", "vx", "vy", "vz", "m", "ax", "ay", "az")
        rec = np.empty((n, 11), np.float32)
        for k, name in enumerate(names):
            rec[:, k] = st[name][order]
        rec[:, 10] = st["id"][order].view(np.float32)          # ids ride along bit for bit
        sc = torch.tensor(send_rows, dtype=torch.int64, device=b.device)
        rc = torch.empty_like(sc)
        dist.all_to_all_single(rc, sc, group=self.pg)
        recv_rows = [int(v) for v in rc.cpu()]
        send = torch.from_numpy(rec).to(b.device)
        recv = torch.empty(sum(recv_rows), 11, dtype=torch.float32, device=b.device)
        dist.all_to_all_single(recv, send, recv_rows, send_rows, group=self.pg)
        got = recv.cpu().numpy()
        new = {name: np.ascontiguousarray(got[:, k]) for k, name in enumerate(names)}
        new["id"] = np.ascontiguousarray(got[:, 10]).view(np.uint32)
        b.set_particles(new)
        return n - send_rows[me]


def snap_to_cells(lo, hi, grid):
    """The coordinate in [lo, hi] that is a boundary of the coarsest Morton cells: grid =
    (domain_size, bits per axis); cell q of an axis is [q, q+1) * domain / (2^bits - 1), the
    quantisation of the library's keys.  A domain cut a hair beside a coarse cell boundary
    leaves a thin slab of the domain's particles on the far side of it; in Morton order they
    form groups that are spread over the whole cut face, whose source lists are several times
    the mean (measured: 27 k against 7 k) and whose waves outlast the traversal.  None (no
    cell boundary inside) leaves the cut where it is."""
    import math
    domain, bits = grid
    scale = ((1 << bits) - 1) / float(domain)
    tlo, thi = math.ceil(lo * scale), math.floor(hi * scale)
    if tlo > thi:
        return None
    for p in range(bits, -1, -1):
        m = -(-tlo >> p) << p          # smallest multiple of 2^p that is >= tlo
        if m <= thi:
            return m / scale
    return None


def orb_cuts(points, parts, weights=None, grid=None, slack=0.005):
    """Orthogonal recursive bisection of `points` (k, 3) into `parts` boxes covering all of
    space: nested tuples (axis, threshold, parts_left, left, right), leaves = None.  Every cut
    goes across the longest axis of the piece at the quantile -- of the point count, or of
    `weights` (k,) when given -- that matches the ranks on either side; with equal weights it
    is the rule of domain_slices.  With grid = (domain_size, bits) a cut moves to the
    coarsest Morton cell boundary within `slack` of the piece's points (snap_to_cells)."""
    import numpy as np
    pts = np.asarray(points, np.float64)
    w = np.ones(len(pts)) if weights is None else np.asarray(weights, np.float64)
    if parts == 1:
        return None
    left = parts // 2
    axis = int(np.argmax(pts.max(0) - pts.min(0))) if len(pts) else 0
    if len(pts) >= 2:
        order = np.argsort(pts[:, axis], kind="stable")
        v = pts[order, axis]
        cw = np.cumsum(w[order])
        # first k points carry (just under) left/parts of the weight; counts when w == 1
        k = int(np.searchsorted(cw, cw[-1] * left / parts, side="right"))
        k = min(max(k, 1), len(v) - 1)
        thr = 0.5 * (v[k - 1] + v[k])
        if grid is not None:
            d = int(slack * len(v))
            snapped = snap_to_cells(v[max(k - 1 - d, 0)], v[min(k + d, len(v) - 1)], grid)
            if snapped is not None:
                thr = snapped
    else:
        thr = float(pts[0, axis]) if len(pts) else 0.0
    m = pts[:, axis] < thr
    return (axis, float(thr), left, orb_cuts(pts[m], left, w[m], grid, slack),
            orb_cuts(pts[~m], parts - left, w[~m], grid, slack))


def orb_assign(cuts, pos, first=0):
    """Domain index of every row of pos (k, 3) under orb_cuts' tree."""
    import numpy as np
    pos = np.asarray(pos)
    out = np.full(len(pos), first, np.int64)
    if cuts is None or len(pos) == 0:
        return out
    axis, thr, left, lo, hi = cuts
    m = pos[:, axis].astype(np.float64) < thr
    out[m] = orb_assign(lo, pos[m], first)
    out[~m] = orb_assign(hi, pos[~m], first + left)
    return out


def domain_slices(x, y, z, world, grid=None, slack=0.005):
    """Index arrays of `world` domains of (nearly) equal particle count with disjoint, tight
    bounding boxes: orthogonal recursive bisection of the given positions (numpy), each cut
    across the longest axis of the piece, in the ratio of the ranks on either side.  This is
    the initial ownership of a LetStepper run.  With grid = (domain_size, bits per axis) a cut
    moves to the coarsest Morton cell boundary within `slack` of the piece's particles
    (snap_to_cells); the counts then differ by up to that fraction."""
    import numpy as np
    pos = np.stack([np.asarray(a, np.float64) for a in (x, y, z)], 1)

    def cut(idx, parts):
        if parts == 1:
            return [np.sort(idx)]
        left = parts // 2
        p = pos[idx]
        axis = int(np.argmax(p.max(0) - p.min(0)))
        k = len(idx) * left // parts
        order = np.argsort(p[:, axis], kind="stable")
        if grid is not None and 1 <= k < len(idx):
            v = p[order, axis]
            d = int(slack * len(v))
            snapped = snap_to_cells(v[max(k - 1 - d, 0)], v[min(k + d, len(v) - 1)], grid)
            if snapped is not None:
                k = int(np.searchsorted(v, snapped, side="left"))   # x < cut goes left
        return cut(idx[order[:k]], left) + cut(idx[order[k:]], parts - left)

    return cut(np.arange(len(pos)), world)


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
        def __init__(self, ptr, n, typestr):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False),
                                             "version": 2}

    def __init__(self, sim, device):
        self.sim = sim
        self.n = sim.params.num_points
        self.group = sim.params.traversal_group_size
        # the library's acceleration arrays, zero-copy: float32, or float64 for a handle
        # created with FLAG_FP64
        v, ts = (sim.device_view_f64(), "<f8") if sim.fp64 else (sim.device_view(), "<f4")
        self._acc = [torch.as_tensor(self._Dev(p, self.n, ts), device=device) for p in (v.ax, v.ay, v.az)]

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


class GpuLetBackend:
    """LetStepper backend over two nbody_amd.Simulation handles on one GPU: `sim` holds the
    own particles, `remote` (created here, grown on demand) the imported records."""

    def __init__(self, sim, device, record_capacity=None, exchange_device=None,
                 remote_codes_per_leaf=None):
        import nbody_amd as nb
        self._nb = nb
        self.sim = sim
        self.gpu = torch.device(device)
        # where the exchanged tensors live: the GPU itself under RCCL; "cpu" lets several
        # ranks share one GPU over gloo (tests)
        self.device = torch.device(exchange_device) if exchange_device else self.gpu
        self.n = sim.params.num_points
        self.grid = (sim.params.domain_size, sim.params.morton_bits)   # cuts snap to its cells
        g = sim.params.traversal_group_size
        sim.set_traversal_range(0, 0)   # update() = build only; forces and kick are driven here
        self.cap = int(record_capacity or max(4096, self.n // 4))
        # Leaves are never approximated, and a far region is represented by a handful of records
        # that would all share one 32-code leaf: small leaves let the opening criterion work on
        # them (import traversal 2.05 -> 1.35 ms at 8 x 2,097,152 particles).
        self.remote_codes_per_leaf = int(remote_codes_per_leaf or
                                         min(8, sim.params.max_num_codes_per_leaf))
        self._buf = None
        self.remote = None
        self._remote_cap = 0

    def update(self):
        self.sim.update()

    def bbox(self):
        return self.sim.get_bbox()

    def let_export(self, boxes):
        k = len(boxes)
        while True:
            if self._buf is None or self._buf.shape[0] < k or self._buf.shape[1] < self.cap:
                self._buf = torch.empty(k, self.cap, 4, dtype=torch.float32, device=self.gpu)
            try:
                counts = self.sim.let_export(boxes, self._buf.data_ptr(), self._buf.shape[1])
                break
            except self._nb.CapacityError as e:
                self.cap = int(max(e.counts) * 1.25) + 1024
                self._buf = None
        return [self._buf[i, :int(c)].to(self.device) for i, c in enumerate(counts)]

    def begin_local_forces(self):
        self.sim.traverse_from(self.sim, accumulate=False)   # queued, not waited for

    def add_remote(self, records):
        cnt = int(records.shape[0])
        records = records.to(self.gpu).contiguous()
        if records.is_cuda:
            torch.cuda.current_stream(self.gpu).synchronize()
        if cnt > self._remote_cap:
            if self.remote is not None:
                self.remote.close()
            self._remote_cap = max(int(cnt * 1.25) + 1024, self.n // 2)
            p = self._nb.Params(**{**self.sim.params.__dict__, "num_points": self._remote_cap,
                                   "max_num_codes_per_leaf": self.remote_codes_per_leaf,
                                   "num_steps_validator": 0, "flags": 0})
            self.remote = self._nb.Simulation(p)
        if self.remote is None:
            return
        self.remote.set_sources_dev(records.data_ptr() if cnt else 0, cnt)
        self.sim.traverse_from(self.remote, accumulate=True)

    def finish_step(self):
        self.sim.finish_step()

    def work_weights(self):
        """Per particle (current order): sources its group met in the own tree plus in the
        imported tree.  Runs both traversals again to read their list lengths one after the
        other; the accelerations come out as they were (same inputs, same sums)."""
        import numpy as np
        self.sim.traverse_from(self.sim, accumulate=False)
        per_group = self.sim.get_list_len().astype(np.float64)
        if self.remote is not None and self.remote.count() > 0:
            self.sim.traverse_from(self.remote, accumulate=True)
            per_group += self.sim.get_list_len()
        n = self.sim.count()
        return np.repeat(per_group, self.sim.params.traversal_group_size)[:n]

    def get_particles(self):
        return self.sim.get_state()

    def set_particles(self, st):
        if len(st["x"]) > self.sim.params.num_points:
            raise RuntimeError(f"domain received {len(st['x'])} particles, handle capacity is "
                               f"{self.sim.params.num_points}: create it with head-room")
        self.sim.set_particles(st["x"], st["y"], st["z"], st["m"], st["vx"], st["vy"], st["vz"],
                               st["ax"], st["ay"], st["az"], st["id"])
        self.n = len(st["x"])
        self.sim.set_traversal_range(0, 0)

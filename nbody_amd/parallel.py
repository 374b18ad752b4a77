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

* DomainDecomposition / LetStepper -- Barnes-Hut step with DOMAIN DECOMPOSITION along the
  Morton curve.  Every rank owns the particles of one contiguous range of Morton keys (ranges
  re-drawn now and then from an all-gathered histogram of key prefixes, particles migrated by
  an all-to-all), builds a tree over them only and traverses it; what its bodies need from the
  other domains arrives as a locally-essential tree: an all-gather of every domain's top-of-
  tree nodes (boxes, centres of mass, masses, sizes), then ONE all-to-all of (x,y,z,m) records
  -- the centres of mass of remote nodes that pass the opening criterion against the nearest
  box of the receiving domain, and the particles of remote leaves that do not.  A second tree
  is built over the received records and traversed for the own bodies.  Memory, build and
  traversal scale with N/P; the exchanged volume is a surface term.  DomainDecomposition
  drives the native implementation (nbody_dd_*, csrc/dd.hip) with torch.distributed
  collectives; LetStepper is the same choreography over an injected backend (CPU tests).

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


DD_BINS = 4096      # histogram bins per refinement round of the key-range re-balancing
DD_BIN_BITS = 12
DD_SLACK = 0.005    # a cut may miss its target by this share of a domain's particles


def morton_splitters(lower_bounds, gather_sum, parts, key_bits, slack_share=DD_SLACK):
    """Key-range splitters of `parts` domains of (nearly) equal particle count: the protocol of
    nbody_dd_rebalance (csrc/dd.hip) with the collectives injected.
      lower_bounds(q)  positions of the uint64 query keys q in this rank's SORTED keys
      gather_sum(a)    element-wise sum over all ranks of the uint64 array a
    The first histogram is over the top 12 key bits (4096 bins) and gives, for every cut, the coarsest
    bin edge within the slack of its target; a cut that no edge meets is refined, round after
    round, with a 4096-bin histogram of the next 12 bits of the bin that holds its target.  Returns
    (splitters uint64[parts + 1], total particle count); splitters[parts] = 2^key_bits."""
    import numpy as np
    import nbody_amd as nb
    q = np.arange(DD_BINS + 1, dtype=np.uint64) << np.uint64(key_bits - DD_BIN_BITS)
    lb = np.asarray(lower_bounds(q), np.int64)
    hist = np.asarray(gather_sum(np.diff(lb).astype(np.uint64)), np.uint64)
    total = int(hist.sum())
    if total == 0:
        raise RuntimeError("re-balance: no particles anywhere")
    slack = int(slack_share * total / parts)
    edge, bin_ = nb.choose_edges(hist, parts, slack)
    P1 = np.concatenate([[0], np.cumsum(hist.astype(object))])
    split = [0] * (parts + 1)
    split[parts] = 1 << key_bits
    pending = []          # (cut, prefix of the bin that holds its target, particles before that bin)
    for k in range(1, parts):
        if edge[k - 1] >= 0:
            split[k] = int(edge[k - 1]) << (key_bits - DD_BIN_BITS)
        else:
            pending.append((k, int(bin_[k - 1]), int(P1[int(bin_[k - 1])])))
    bits = DD_BIN_BITS
    while pending:
        nb_ = min(DD_BIN_BITS, key_bits - bits)
        nbins, sh = 1 << nb_, key_bits - bits - nb_
        q2 = np.array([((prefix << nb_) + j) << sh for _, prefix, _ in pending for j in range(nbins + 1)],
                      dtype=np.uint64)
        lb2 = np.asarray(lower_bounds(q2), np.int64).reshape(len(pending), nbins + 1)
        h2 = np.asarray(gather_sum(np.diff(lb2, axis=1).astype(np.uint64).reshape(-1)), np.uint64)
        h2 = h2.reshape(len(pending), nbins)
        nxt = []
        for (k, prefix, base), h in zip(pending, h2):
            target = total * k // parts
            j = nb.refine_edge(h, base, target, slack, prefix << nb_, bits + nb_)
            cum = base + int(h[:j].sum())
            if abs(cum - target) <= slack or bits + nb_ == key_bits:
                split[k] = ((prefix << nb_) + j) << sh
            else:               # descend into the sub-bin that holds the target
                before, b = base, 0
                while b + 1 < nbins and before + int(h[b]) <= target:
                    before += int(h[b])
                    b += 1
                nxt.append((k, (prefix << nb_) + b, before))
        pending = nxt
        bits += nb_
    for k in range(1, parts + 1):
        split[k] = max(split[k], split[k - 1])
    return np.array(split, dtype=np.uint64), total


class LetStepper:
    """Domain decomposition along the Morton curve with a locally-essential-tree exchange, the
    choreography of nbody_dd_step / nbody_dd_rebalance (csrc/dd.hip) written against an
    injected backend, so that it runs without a GPU (CPU tests: host oracle as the backend,
    gloo collectives).  On GPUs use DomainDecomposition, which drives the native code.
    backend must provide:
        device                        torch device of the exchanged tensors
        key_bits                      30 or 63
        update()                      kick-drift, sort, build of the own tree (no forces yet)
        frontier(T)                   (T, 12) float32 tensor: top-of-tree records (lo xyz, hi xyz,
                                      com xyz, mass, size, count), empty records lo = 3e38
        let_export(frontiers, me)     frontiers (W, T, 12) tensor -> list of W (cnt_r, 4) float32
                                      tensors of (x,y,z,m) records (entry `me` empty)
        begin_local_forces()          own-tree traversal (may return before it is done)
        add_remote(records)           (cnt, 4) tensor: build a tree over it, add its attraction
        finish_step()                 velocity kick
      for rebalance():
        key_lower_bounds(q)           positions of uint64 keys in the own sorted keys
        get_particles()/set_particles(dict)   x,y,z,vx,vy,vz,m,ax,ay,az float32, id uint32,
                                      in the CURRENT (key-sorted) order
    """

    def __init__(self, backend, process_group=None, records_per_domain=32):
        self.b = backend
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.T = records_per_domain
        self.last_sent = 0
        self.last_received = 0
        self.splitters = None

    def step(self):
        b, W, me = self.b, self.world, self.rank
        b.update()
        dev = b.device
        if W == 1:
            b.begin_local_forces()
            b.finish_step()
            return
        fr = b.frontier(self.T).to(dev).contiguous()
        allfr = torch.empty(W * self.T * 12, dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(allfr, fr.view(-1), group=self.pg)
        recs = b.let_export(allfr.view(W, self.T, 12), me)
        b.begin_local_forces()
        send_rows = [0 if r == me else int(t.shape[0]) for r, t in enumerate(recs)]
        sc = torch.tensor(send_rows, dtype=torch.int64, device=dev)
        rc = torch.empty_like(sc)
        dist.all_to_all_single(rc, sc, group=self.pg)
        recv_rows = [int(v) for v in rc.cpu()]
        parts = [t.to(dev) for r, t in enumerate(recs) if r != me and t.shape[0]]
        send = torch.cat(parts) if parts else torch.empty(0, 4, dtype=torch.float32, device=dev)
        recv = torch.empty(sum(recv_rows), 4, dtype=torch.float32, device=dev)
        dist.all_to_all_single(recv, send.contiguous(), recv_rows, send_rows, group=self.pg)
        self.last_sent, self.last_received = sum(send_rows), sum(recv_rows)
        b.add_remote(recv)
        b.finish_step()

    def _gather_sum(self, a):
        import numpy as np
        t = torch.from_numpy(np.ascontiguousarray(a, np.uint64).view(np.int64).copy()).to(self.b.device)
        dist.all_reduce(t, group=self.pg)
        return t.cpu().numpy().view(np.uint64)

    def rebalance(self):
        """Re-draws the domains as contiguous Morton-key ranges of (nearly) equal particle
        counts (morton_splitters) and migrates the particles with one all-to-all of 11-word
        records.  Returns the number of particles this rank gave away."""
        import numpy as np
        b, W, me = self.b, self.world, self.rank
        split, _ = morton_splitters(b.key_lower_bounds, self._gather_sum, W, b.key_bits)
        self.splitters = split
        if W == 1:
            return 0
        st = b.get_particles()
        n = len(st["x"])
        lb = np.asarray(b.key_lower_bounds(split), np.int64)
        lb[0], lb[W] = 0, n
        send_rows = np.diff(lb).tolist()
        names = ("x", "y", "z", "vx", "vy", "vz", "m", "ax", "ay", "az")
        rec = np.empty((n, 11), np.float32)
        for k, name in enumerate(names):
            rec[:, k] = st[name]
        rec[:, 10] = st["id"].astype(np.uint32).view(np.float32)   # ids ride along bit for bit
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


class TorchComm:
    """struct nbody_comm over torch.distributed: the two collectives nbody_dd_* needs, on raw
    device pointers, enqueued on the hipStream_t the library passes (made torch's current
    stream for the call, so RCCL orders itself after the kernels already queued there and the
    kernels queued afterwards wait for it; the host does not wait).  via_host = True stages
    through host memory (gloo, or several ranks sharing one GPU in tests)."""

    class _Raw:
        def __init__(self, ptr, nbytes):
            self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1",
                                             "data": (ptr, False), "version": 2}

    def __init__(self, device, process_group=None, via_host=False):
        import ctypes as C
        import nbody_amd as nb
        self.device = torch.device(device)
        self.pg = process_group
        self.via_host = via_host
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.error = None
        # fault injection for tests of the error path: the all_to_all call with this index
        # (0-based) reports failure instead of communicating; None = never
        self.fail_all_to_all_at = None
        self.calls = {"all_gather": 0, "all_to_all": 0}
        self.bytes = {"all_gather": 0, "all_to_all_sent": 0, "all_to_all_received": 0}
        self._ag = nb.CComm.ALL_GATHER(self._all_gather)
        self._a2a = nb.CComm.ALL_TO_ALL(self._all_to_all)
        self.c = nb.CComm(self.rank, self.size, None, self._ag, self._a2a)

    def _t(self, ptr, nbytes):
        if not nbytes or not ptr:
            return torch.empty(0, dtype=torch.uint8, device=self.device)
        return torch.as_tensor(self._Raw(ptr, nbytes), device=self.device)

    def _stream(self, stream):
        return torch.cuda.stream(torch.cuda.ExternalStream(stream, device=self.device))

    def _all_gather(self, ctx, send, recv, nbytes, stream):
        try:
            self.calls["all_gather"] += 1
            self.bytes["all_gather"] += int(nbytes)
            with self._stream(stream):
                s, r = self._t(send, nbytes), self._t(recv, nbytes * self.size)
                if self.via_host:
                    hr = torch.empty(nbytes * self.size, dtype=torch.uint8)
                    dist.all_gather_into_tensor(hr, s.cpu(), group=self.pg)
                    r.copy_(hr)
                else:
                    dist.all_gather_into_tensor(r, s, group=self.pg)
            return 0
        except Exception as e:      # never let an exception cross the C boundary
            self.error = e
            return 1

    def _all_to_all(self, ctx, send, sbytes, soff, recv, rbytes, roff, stream):
        try:
            W = self.size
            k = self.calls["all_to_all"]
            self.calls["all_to_all"] = k + 1
            if self.fail_all_to_all_at is not None and k == self.fail_all_to_all_at:
                raise RuntimeError(f"injected failure of all_to_all call {k} on rank {self.rank}")
            sb, so = [int(sbytes[r]) for r in range(W)], [int(soff[r]) for r in range(W)]
            rb, ro = [int(rbytes[r]) for r in range(W)], [int(roff[r]) for r in range(W)]
            self.bytes["all_to_all_sent"] += sum(sb)
            self.bytes["all_to_all_received"] += sum(rb)
            with self._stream(stream):
                sbuf = self._t(send, max((o + b for o, b in zip(so, sb)), default=0))
                rbuf = self._t(recv, max((o + b for o, b in zip(ro, rb)), default=0))
                packed_in = all(so[r] == sum(sb[:r]) for r in range(W))
                packed_out = all(ro[r] == sum(rb[:r]) for r in range(W))
                inp = sbuf[:sum(sb)] if packed_in else torch.cat([sbuf[o:o + b] for o, b in zip(so, sb)])
                out = rbuf[:sum(rb)] if packed_out else torch.empty(sum(rb), dtype=torch.uint8, device=self.device)
                if self.via_host:
                    ho = torch.empty(sum(rb), dtype=torch.uint8)
                    dist.all_to_all_single(ho, inp.cpu(), rb, sb, group=self.pg)
                    out.copy_(ho)
                else:
                    dist.all_to_all_single(out, inp.contiguous(), rb, sb, group=self.pg)
                if not packed_out:
                    at = 0
                    for o, b in zip(ro, rb):
                        rbuf[o:o + b].copy_(out[at:at + b])
                        at += b
            return 0
        except Exception as e:
            self.error = e
            return 1


class DomainDecomposition:
    """nbody_dd_* over a nbody_amd.Simulation: the native multi-GPU step (Morton-range
    ownership, frontier all-gather, LET all-to-all) with torch.distributed collectives.
    `sim` holds this rank's particles; create it with head-room (num_points >= the most
    particles the domain may ever hold)."""

    def __init__(self, sim, device, process_group=None, via_host=False):
        import ctypes as C
        import nbody_amd as nb
        self._nb, self._C = nb, C
        self.sim = sim
        self.comm = TorchComm(device, process_group, via_host)
        self._lib = nb.load_library()
        self._h = C.c_void_p()
        rc = self._lib.nbody_dd_create(sim._h, C.byref(self.comm.c), C.byref(self._h))
        if rc != nb.OK:
            raise RuntimeError(f"nbody_dd_create failed with status {rc}")

    def _check(self, rc):
        if rc != self._nb.OK:
            msg = self._lib.nbody_dd_last_error(self._h)
            raise RuntimeError(f"nbody_dd status {rc}: {msg.decode() if msg else ''}"
                               + (f" ({self.comm.error!r})" if self.comm.error else ""))

    def spawn(self, kind, n_total, p0=0.0, p1=0.0):
        self._check(self._lib.nbody_dd_spawn(self._h, kind, n_total, p0, p1))

    def rebalance(self):
        self._check(self._lib.nbody_dd_rebalance(self._h))

    def step(self):
        self._check(self._lib.nbody_dd_step(self._h))

    def stats(self):
        st = self._nb.CDDStats()
        sp = (self._C.c_uint64 * (self.comm.size + 1))()
        self._check(self._lib.nbody_dd_get_stats(self._h, self._C.byref(st), sp))
        out = {k: getattr(st, k) for k, _ in self._nb.CDDStats._fields_}
        out["splitters"] = [int(v) for v in sp]
        return out

    def set_phase_timing(self, on=True):
        self._check(self._lib.nbody_dd_set_phase_timing(self._h, 1 if on else 0))

    def phases(self):
        """Times (ms) and bytes of the last step's phases on this rank (set_phase_timing first)."""
        ph = self._nb.CDDPhases()
        self._check(self._lib.nbody_dd_get_phases(self._h, self._C.byref(ph)))
        return {k: getattr(ph, k) for k, _ in self._nb.CDDPhases._fields_ if k != "pad"}

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.nbody_dd_destroy(self._h)
            self._h = self._C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


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

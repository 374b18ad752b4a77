"""world_size-2/3 gloo tests (CPU) of the multi-GPU choreography in nbody_amd/parallel.py.
The compute is injected: here it is the host oracle, on the GPUs it is libnbody_mi355.so."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
from nbody_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleBackend:
    """ShardedTraversal backend made of oracle stages: replicated state, sharded traversal."""

    def __init__(self, ic, n, group, theta, dt, eps, bits):
        self.n, self.group, self.theta, self.dt, self.eps, self.bits = n, group, theta, dt, eps, bits
        b = O.full_build(ic["x"], ic["y"], ic["z"], ic["m"], 5.0, bits, 32)
        p = b["perm"]
        self.x, self.y, self.z, self.m = b["xs"], b["ys"], b["zs"], b["ms"]
        self.v = [ic[k][p].copy() for k in ("vx", "vy", "vz")]
        self.acc = [torch.zeros(n, dtype=torch.float32) for _ in range(3)]
        self.g0, self.g1 = 0, (n + group - 1) // group

    def set_traversal_range(self, g0, g1):
        self.g0, self.g1 = g0, g1

    def update(self):
        a = [t.numpy() for t in self.acc]
        x, y, z, vh = O.leapfrog_pos(self.dt, 5.0, self.x, self.y, self.z, *self.v, *a)
        b = O.full_build(x, y, z, self.m, 5.0, self.bits, 32)
        p = b["perm"]
        self.x, self.y, self.z, self.m = b["xs"], b["ys"], b["zs"], b["ms"]
        self.vh = [h[p] for h in vh]
        ax, ay, az, _ = O.traverse(b["tree"], b["com"], self.x, self.y, self.z, self.m, self.group,
                                   self.theta, self.eps, 1.0, 5.0)
        lo, hi = self.g0 * self.group, min(self.g1 * self.group, self.n)
        for t, full in zip(self.acc, (ax, ay, az)):
            t.zero_()                      # only the own range is "computed" on this rank
            t[lo:hi] = torch.from_numpy(full[lo:hi])

    def wait(self):
        pass

    def acc_tensors(self):
        return self.acc

    def finish_step(self):
        self.v = [(np.float32(0.5) * a.numpy() * np.float32(self.dt) + h).astype(np.float32)
                  for a, h in zip(self.acc, self.vh)]


def _ic(n):
    import nbody_amd as nb
    return nb.sample_ic(nb.IC_DISK, n, 7)


def _sharded_worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    be = OracleBackend(_ic(n), n, 64, 0.5, 1e-3, 0.01, 21)
    st = parallel.ShardedTraversal(be)
    for _ in range(3):
        st.step()
    if rank == 0:
        q.put((be.x, be.v[0], be.acc[0].numpy().copy(), st.even))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 2048), (2, 1000), (3, 1500)])
def test_sharded_traversal_equals_single_rank(world, n):
    """even split (2048 = 2*16 groups), ragged split (1000), 3 ranks with a partial chunk."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    x, vx, ax, even = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert even == (n % (world * 64) == 0)
    ref = OracleBackend(_ic(n), n, 64, 0.5, 1e-3, 0.01, 21)
    for _ in range(3):
        ref.update()
        ref.finish_step()
    assert np.array_equal(x, ref.x) and np.array_equal(vx, ref.v[0])
    assert np.array_equal(ax, ref.acc[0].numpy())


def _ring_worker(rank, world, port, n_local, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = n_local * world
    ic = _ic(n)
    posm = np.stack([ic["x"], ic["y"], ic["z"], ic["m"]], 1).astype(np.float32)
    mine = torch.from_numpy(posm[rank * n_local:(rank + 1) * n_local].copy())
    acc = [np.zeros(n_local, np.float64) for _ in range(3)]
    seen = []

    def tile_kernel(targets, sources, accumulate):
        t, s = targets.numpy(), sources.numpy()
        seen.append(s[0, 0].item())
        if not accumulate:
            for a in acc:
                a[:] = 0
        # forces of the tile's bodies on my bodies: float64 direct sum, targets placed first
        both = np.concatenate([t, s])
        r = O.direct_f64(both[:, 0], both[:, 1], both[:, 2],
                         np.concatenate([np.zeros(len(t), np.float32), s[:, 3]]), 0.01, 1.0,
                         0, len(t))
        for a, d in zip(acc, r):
            a += d

    parallel.ring_direct_sum(mine, tile_kernel)
    q.put((rank, acc[0].copy(), acc[1].copy(), seen))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ring_direct_sum_equals_full_sum(world):
    n_local = 200
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ring_worker, args=(r, world, port, n_local, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n = n_local * world
    ic = _ic(n)
    rx, ry, _ = O.direct_f64(ic["x"], ic["y"], ic["z"], ic["m"], 0.01, 1.0)
    for rank, ax, ay, seen in res:
        sl = slice(rank * n_local, (rank + 1) * n_local)
        assert np.allclose(ax, rx[sl], rtol=1e-10) and np.allclose(ay, ry[sl], rtol=1e-10)
        # every rank saw every tile exactly once, its own first, then its left neighbours'
        order = [(rank - k) % world for k in range(world)]
        assert seen == [float(ic["x"][o * n_local]) for o in order]


class OracleLetBackend:
    """LetStepper backend made of oracle stages: this rank owns `idx` of the particles."""
    device = torch.device("cpu")

    def __init__(self, ic, idx, group, theta, dt, eps, bits):
        self.group, self.theta, self.dt, self.eps, self.bits = group, theta, dt, eps, bits
        self.x, self.y, self.z, self.m = (ic[k][idx].copy() for k in ("x", "y", "z", "m"))
        self.v = [ic[k][idx].copy() for k in ("vx", "vy", "vz")]
        self.gid = np.asarray(idx).copy()
        self.acc = [np.zeros(len(idx), np.float32) for _ in range(3)]

    def update(self):
        x, y, z, vh = O.leapfrog_pos(self.dt, 5.0, self.x, self.y, self.z, *self.v, *self.acc)
        b = O.full_build(x, y, z, self.m, 5.0, self.bits, 32)
        p = b["perm"]
        self.x, self.y, self.z, self.m = b["xs"], b["ys"], b["zs"], b["ms"]
        self.vh = [h[p] for h in vh]
        self.gid = self.gid[p]
        self.tree = b
        self.acc = None

    def begin_local_forces(self):
        b = self.tree
        ax, ay, az, ll = O.traverse(b["tree"], b["com"], self.x, self.y, self.z, self.m, self.group,
                                    self.theta, self.eps, 1.0, 5.0)
        self.acc = [ax, ay, az]
        self.work = ll.astype(np.float64)

    def work_weights(self):
        return np.repeat(self.work, self.group)[:len(self.x)]

    def bbox(self):
        return np.array([self.x.min(), self.y.min(), self.z.min(),
                         self.x.max(), self.y.max(), self.z.max()], np.float32)

    def let_export(self, boxes):
        b = self.tree
        return [torch.from_numpy(O.let_export(b["tree"], b["com"], self.x, self.y, self.z, self.m,
                                              box, self.theta)) for box in boxes]

    def add_remote(self, records):
        r = records.numpy()
        if len(r) == 0:
            return
        b = O.full_build(r[:, 0], r[:, 1], r[:, 2], r[:, 3], 5.0, self.bits, 32)
        ax, ay, az, ll = O.traverse_from((self.x, self.y, self.z), b["tree"], b["com"], b["xs"],
                                         b["ys"], b["zs"], b["ms"], self.group, self.theta,
                                         self.eps, 1.0, 5.0, acc=self.acc)
        self.acc = [ax, ay, az]
        self.work = self.work + ll

    def finish_step(self):
        self.v = [(np.float32(0.5) * a * np.float32(self.dt) + h).astype(np.float32)
                  for a, h in zip(self.acc, self.vh)]

    def get_particles(self):
        return dict(x=self.x, y=self.y, z=self.z, m=self.m, vx=self.v[0], vy=self.v[1], vz=self.v[2],
                    ax=self.acc[0], ay=self.acc[1], az=self.acc[2], id=self.gid.astype(np.uint32))

    def set_particles(self, st):
        self.x, self.y, self.z, self.m = st["x"], st["y"], st["z"], st["m"]
        self.v = [st["vx"], st["vy"], st["vz"]]
        self.acc = [st["ax"], st["ay"], st["az"]]
        self.gid = st["id"].astype(np.int64)


def _let_worker(rank, world, port, n, q, rebalance_at=-1):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ic = _ic(n)
    idx = parallel.domain_slices(ic["x"], ic["y"], ic["z"], world)[rank]
    be = OracleLetBackend(ic, idx, 64, 0.5, 1e-3, 0.01, 21)
    st = parallel.LetStepper(be)
    sent = []
    moved = 0
    for i in range(3):
        if i == rebalance_at:
            moved = st.rebalance(sample=512, seed=i, weighted=True)
        st.step()
        sent.append((st.last_sent, st.last_received))
    q.put((rank, be.gid, be.x, be.y, be.v[0], be.acc[0], be.acc[1], be.acc[2], sent, moved,
           float(be.work_weights().sum())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n,rebalance_at", [(2, 3000, -1), (3, 2500, -1), (3, 2500, 2)])
def test_let_stepper_tracks_single_rank_and_direct_sum(world, n, rebalance_at):
    """Domain decomposition over gloo ranks: every rank owns n/world particles and imports a
    locally-essential tree from the others.  The forces are not bit-identical to a single
    tree (different trees), so the check is accuracy against a float64 direct sum at the
    final positions plus closeness of the trajectories to the single-rank run."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_let_worker, args=(r, world, port, n, q, rebalance_at))
             for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ic = _ic(n)
    x, y, vx, ax, ay, az = (np.full(n, np.nan, np.float32) for _ in range(6))
    for rank, gid, rx, ry, rvx, rax, ray, raz, sent, moved, work in res:
        x[gid], y[gid], vx[gid], ax[gid], ay[gid], az[gid] = rx, ry, rvx, rax, ray, raz
        # a surface term: every step something is exchanged, far fewer records than particles
        assert all(0 < s < n and 0 < r < n for s, r in sent)
    assert not np.isnan(x).any()                      # every particle owned exactly once
    assert sum(len(r[1]) for r in res) == n
    if rebalance_at >= 0:
        # the disk has moved for two steps: the fresh bisection differs from the initial one
        assert sum(r[9] for r in res) > 0
        # the new cuts equalise work (body-source interactions of the last force evaluation),
        # not particle counts
        works = np.array([r[10] for r in res])
        assert works.max() < 1.25 * works.mean(), works
    ref = OracleBackend(ic, n, 64, 0.5, 1e-3, 0.01, 21)
    for _ in range(3):
        ref.update()
        ref.finish_step()
    # ref is in its own sorted order; map back through a fresh sort of the final positions
    one = OracleLetBackend(ic, np.arange(n), 64, 0.5, 1e-3, 0.01, 21)
    for _ in range(3):
        one.update()
        one.begin_local_forces()
        one.finish_step()
    assert np.array_equal(np.sort(one.x), np.sort(ref.x))   # world=1 LET == plain stepper
    ox = np.empty(n, np.float32); ox[one.gid] = one.x
    ovx = np.empty(n, np.float32); ovx[one.gid] = one.v[0]
    assert np.max(np.abs(x - ox)) < 1e-5
    assert np.max(np.abs(vx - ovx)) < 2e-3 * np.max(np.abs(ovx))
    # accuracy of the last force evaluation against float64 all-pairs at the same positions
    zz = np.empty(n, np.float32); zz[one.gid] = one.z
    d = O.direct_f64(x, y, zz, ic["m"], 0.01, 1.0)
    mean, mx = O.acc_error(ax, ay, az, *d)
    assert mean < 2e-3, mean


def test_orb_cuts_follow_the_weights():
    rng = np.random.default_rng(3)
    pts = rng.uniform(0, 1, (20000, 3)) * np.array([4.0, 1.0, 1.0])      # longest axis: x
    # unweighted: equal counts, as domain_slices cuts them
    c = parallel.orb_cuts(pts, 4)
    cnt = np.bincount(parallel.orb_assign(c, pts), minlength=4)
    assert cnt.max() - cnt.min() <= 4
    # the right half of the box costs nine times as much per particle
    w = np.where(pts[:, 0] > 2.0, 9.0, 1.0)
    c = parallel.orb_cuts(pts, 4, w)
    dom = parallel.orb_assign(c, pts)
    work = np.bincount(dom, weights=w, minlength=4)
    assert work.max() < 1.02 * work.mean()
    assert np.bincount(dom, minlength=4).max() > 2 * np.bincount(dom, minlength=4).min()
    # three parts: 1/3 of the weight left of the first cut
    c3 = parallel.orb_cuts(pts, 3, w)
    work3 = np.bincount(parallel.orb_assign(c3, pts), weights=w, minlength=3)
    assert work3.max() < 1.02 * work3.mean()


def test_cuts_snap_to_coarse_cell_boundaries():
    """A bisection of a centred sphere puts its median a hair beside x = domain/2, the boundary
    of the coarsest Morton cells; with `grid` the cut moves onto it, so that no domain holds
    a thin slab of particles on the far side."""
    S = (1 << 21) - 1
    assert parallel.snap_to_cells(2.4999, 2.5002, (5.0, 21)) == (1 << 20) * 5.0 / S
    assert parallel.snap_to_cells(1.0, 1.0000001, (5.0, 21)) is None          # no boundary inside
    got = parallel.snap_to_cells(2.51, 2.52, (5.0, 21)) * S / 5.0
    assert got == round(got) and int(round(got)) % (1 << 12) == 0            # the roundest in range
    rng = np.random.default_rng(5)
    n = 200000
    r = 0.5 * rng.uniform(0, 1, n) ** (1 / 3)
    ct, ph = rng.uniform(-1, 1, n), rng.uniform(0, 2 * np.pi, n)
    st = np.sqrt(1 - ct * ct)
    x, y, z = 2.5 + r * st * np.cos(ph), 2.5 + r * st * np.sin(ph), 2.5 + r * ct
    half = (1 << 20) * 5.0 / S
    plain = parallel.domain_slices(x, y, z, 8)
    snapped = parallel.domain_slices(x, y, z, 8, grid=(5.0, 21))
    assert sorted(np.concatenate(snapped).tolist()) == list(range(n))       # still a partition
    assert max(len(p) for p in snapped) < 1.02 * n / 8                      # within the slack
    def straddlers(parts):
        k = 0
        for p in parts:
            for a in (x, y, z):
                lo, hi = (a[p] < half).sum(), (a[p] >= half).sum()
                k += min(lo, hi)                                            # the minority side
        return k
    assert straddlers(plain) > 0 and straddlers(snapped) == 0
    # the same rule in the re-balancing cuts
    pts = np.stack([x, y, z], 1)
    dom = parallel.orb_assign(parallel.orb_cuts(pts, 8, grid=(5.0, 21)), pts)
    assert straddlers([np.nonzero(dom == d)[0] for d in range(8)]) == 0

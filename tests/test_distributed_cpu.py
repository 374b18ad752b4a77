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
        self.key_bits = 3 * bits
        self.gid = np.asarray(idx).copy()
        self._install({k: ic[k][idx].copy() for k in ("x", "y", "z", "m", "vx", "vy", "vz")},
                      [np.zeros(len(idx), np.float32) for _ in range(3)])

    def _install(self, st, acc):
        """Sort + build, as the library does after set_particles."""
        b = O.full_build(st["x"], st["y"], st["z"], st["m"], 5.0, self.bits, 32)
        p = b["perm"]
        self.x, self.y, self.z, self.m = b["xs"], b["ys"], b["zs"], b["ms"]
        self.v = [st[k][p] for k in ("vx", "vy", "vz")]
        self.acc = [a[p] for a in acc]
        self.gid = self.gid[p]
        self.tree = b

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

    def frontier(self, T):
        b = self.tree
        rec, _ = O.frontier(b["tree"], b["com"], self.x, self.y, self.z, T)
        return torch.from_numpy(rec)

    def let_export(self, frontiers, me):
        b = self.tree
        fr = frontiers.numpy()
        return [torch.empty(0, 4) if r == me else
                torch.from_numpy(O.let_export(b["tree"], b["com"], self.x, self.y, self.z, self.m,
                                              fr[r][:, :6], self.theta)) for r in range(len(fr))]

    def add_remote(self, records):
        r = records.numpy()
        if len(r) == 0:
            return
        b = O.full_build(r[:, 0], r[:, 1], r[:, 2], r[:, 3], 5.0, self.bits, 8)
        ax, ay, az, ll = O.traverse_from((self.x, self.y, self.z), b["tree"], b["com"], b["xs"],
                                         b["ys"], b["zs"], b["ms"], self.group, self.theta,
                                         self.eps, 1.0, 5.0, acc=self.acc)
        self.acc = [ax, ay, az]
        self.work = self.work + ll

    def finish_step(self):
        self.v = [(np.float32(0.5) * a * np.float32(self.dt) + h).astype(np.float32)
                  for a, h in zip(self.acc, self.vh)]

    def key_lower_bounds(self, q):
        return np.searchsorted(self.tree["codes_sorted"], np.asarray(q, np.uint64), side="left")

    def get_particles(self):
        return dict(x=self.x, y=self.y, z=self.z, m=self.m, vx=self.v[0], vy=self.v[1], vz=self.v[2],
                    ax=self.acc[0], ay=self.acc[1], az=self.acc[2], id=self.gid.astype(np.uint32))

    def set_particles(self, st):
        self.gid = st["id"].astype(np.int64)
        self._install(st, [st["ax"], st["ay"], st["az"]])


def _let_worker(rank, world, port, n, q, rebalance_at=-1):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ic = _ic(n)
    idx = np.arange(n * rank // world, n * (rank + 1) // world)   # an arbitrary first ownership
    be = OracleLetBackend(ic, idx, 64, 0.5, 1e-3, 0.01, 21)
    st = parallel.LetStepper(be, records_per_domain=16)
    moved = st.rebalance()                                        # -> Morton key ranges
    keys0, split0 = be.tree["codes_sorted"].copy(), st.splitters.copy()
    sent = []
    for i in range(3):
        if i == rebalance_at:
            moved += st.rebalance()
        st.step()
        sent.append((st.last_sent, st.last_received))
    q.put((rank, be.gid, be.x, be.y, be.v[0], be.acc[0], be.acc[1], be.acc[2], sent, moved,
           split0, keys0))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n,rebalance_at", [(2, 3000, -1), (3, 2500, -1), (3, 2500, 2)])
def test_let_stepper_tracks_single_rank_and_direct_sum(world, n, rebalance_at):
    """Domain decomposition over gloo ranks: every rank owns one Morton-key range (about
    n/world particles) and imports a locally-essential tree from the others.  The forces are
    not bit-identical to a single tree (different trees), so the check is accuracy against a
    float64 direct sum at the final positions plus closeness of the trajectories to the
    single-rank run."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_let_worker, args=(r, world, port, n, q, rebalance_at))
             for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ic = _ic(n)
    x, y, vx, ax, ay, az = (np.full(n, np.nan, np.float32) for _ in range(6))
    for rank, gid, rx, ry, rvx, rax, ray, raz, sent, moved, split, keys0 in res:
        x[gid], y[gid], vx[gid], ax[gid], ay[gid], az[gid] = rx, ry, rvx, rax, ray, raz
        # a surface term: every step something is exchanged, far fewer records than particles
        assert all(0 < s < n and 0 < r < n for s, r in sent)
        # ownership after the first re-balance = the rank's key range, about n / world particles
        assert np.array_equal(split, res[0][10]) and int(split[0]) == 0 and int(split[-1]) == 1 << 63
        assert np.all(keys0 >= split[rank]) and np.all(keys0 < split[rank + 1])
        # both of a domain's cuts may miss their targets by the slack (or a 24-bit sub-bin)
        assert abs(len(keys0) - n / world) <= 2 * 0.005 * n / world + 8
        assert moved > 0
    assert not np.isnan(x).any()                      # every particle owned exactly once
    assert sum(len(r[1]) for r in res) == n
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


def test_cuts_take_the_coarsest_edge_within_the_slack():
    """nbody_dd_choose_edges / nbody_dd_refine_edge (host code of the library)."""
    import nbody_amd as nb
    # 16 bins, uniform: two parts -> the middle edge (8: three trailing zeros), exactly on target
    edge, bin_ = nb.choose_edges(np.full(16, 100), 2, 0)
    assert list(edge) == [8] and list(bin_) == [8]
    # four parts -> edges 4, 8, 12
    assert list(nb.choose_edges(np.full(16, 100), 4, 0)[0]) == [4, 8, 12]
    # a target a hair beside a coarse edge: inside the slack the coarse edge wins over the nearer one
    h = np.full(16, 100)
    h[7] = 130                                        # cumulative at edge 8 = 830, target = 815
    assert list(nb.choose_edges(h, 2, 20)[0]) == [8]
    assert list(nb.choose_edges(h, 2, 10)[0]) == [-1]           # nothing within 10 of 815
    assert list(nb.choose_edges(h, 2, 10)[1]) == [7]            # target inside bin 7 (700..830)
    # refinement inside bin 7: 8 sub-bins of the 130 particles; target 815 = 700 + 115
    sub = np.array([10, 20, 30, 10, 20, 20, 10, 10])            # cumulative 10 30 60 70 90 110 120 130
    assert nb.refine_edge(sub, 700, 815, 0, 7 * 8, 7) == 6         # nearest (810 or 820): first wins ties
    assert nb.refine_edge(sub, 700, 815, 30, 7 * 8, 7) == 8        # within 30: edges 5..8; edge 64 is the coarsest
    assert nb.refine_edge(sub, 700, 815, 6, 7 * 8, 7) == 6         # within 6: 810 (edge 62) or 820 (edge 63): 62 is coarser
    # all particles in one bin: every cut falls into it
    h = np.zeros(4096, np.uint64)
    h[1234] = 1000
    edge, bin_ = nb.choose_edges(h, 4, 1)
    assert list(edge) == [-1, -1, -1] and list(bin_) == [1234] * 3


def test_morton_splitters_balance_a_clustered_set():
    """The two-round histogram protocol on one 'rank' (the collectives are identities):
    clustered keys, 5 domains, counts within the slack or one sub-bin of their targets."""
    rng = np.random.default_rng(2)
    n = 200000
    pts = np.concatenate([rng.normal(2.5, 0.02, (n // 2, 3)), rng.uniform(0, 5, (n // 2, 3))]).clip(0, 5)
    keys = np.sort(O.morton(pts[:, 0], pts[:, 1], pts[:, 2], 5.0, 21))
    lb = lambda q: np.searchsorted(keys, np.asarray(q, np.uint64), side="left")
    split, total = parallel.morton_splitters(lb, lambda a: a, 5, 63)
    assert total == n and int(split[0]) == 0 and int(split[5]) == 1 << 63
    cum = np.searchsorted(keys, split[1:5], side="left")           # particles before each cut
    assert np.all(np.diff(split.astype(object)) >= 0)
    assert np.abs(cum - n * np.arange(1, 5) / 5).max() <= max(0.005 * n / 5, 64), cum
    # 30-bit keys and a uniform set: the cuts of 8 domains are the octant boundaries
    g = (np.arange(40) + 0.5) * (5.0 / 40)                         # a lattice: equal octants
    u = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    k30 = np.sort(O.morton(u[:, 0], u[:, 1], u[:, 2], 5.0, 10))
    split, _ = parallel.morton_splitters(lambda q: np.searchsorted(k30, np.asarray(q, np.uint64)),
                                         lambda a: a, 8, 30)
    assert [int(v) for v in split] == [j << 27 for j in range(9)]

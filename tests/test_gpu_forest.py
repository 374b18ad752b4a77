"""The locally-essential tree shipped as a tree (nbody_let_export_tree_dev), hung behind the
receiver's own tree (nbody_import_trees_dev) and walked with it in one traversal
(nbody_traverse_forest): what nbody_dd_step does.  Checked on one GPU with the ranks played one
after the other: the exported copy is a valid breadth-first tree whose leaves are exactly the
records of the flat export (itself bit-exact against the oracle, tests/test_gpu_let.py), the
forest walk's interaction lists and forces equal a restatement of the group walk in numpy over
the very arrays that were shipped, and the forces meet the bounds of the single-tree run."""
import numpy as np
import pytest
import torch

import oracle_lib as O
import nbody_amd as nb
from test_gpu_let import DOMAIN, T, _params, _rows_sorted, key_ranges

pytestmark = pytest.mark.gpu

LEAF = 0x80000000


def parse_copy(raw, nn, npart, f64=False):
    """Node and particle records of one exported copy: raw uint8 array of 32 nn + 16 npart bytes
    (float handles), 48 nn + 32 npart (double handles: four doubles, then size/first/count/meta)."""
    if f64:
        rec = raw[:48 * nn].reshape(nn, 48)
        ints = rec[:, 32:].copy().view(np.int32).reshape(nn, 4)
        return {"com": rec[:, :32].copy().view(np.float64).reshape(nn, 4), "size": ints[:, 0].copy().view(np.float32),
                "first": ints[:, 1].copy(), "count": ints[:, 2].copy(), "meta": ints[:, 3].copy().view(np.uint32),
                "parts": raw[48 * nn:48 * nn + 32 * npart].copy().view(np.float64).reshape(npart, 4)}
    nodes = raw[:32 * nn].view(np.float32).reshape(nn, 8)
    ints = raw[:32 * nn].view(np.int32).reshape(nn, 8)
    parts = raw[32 * nn:32 * nn + 16 * npart].view(np.float32).reshape(npart, 4)
    return {"com": nodes[:, :4].copy(), "size": nodes[:, 4].copy(), "first": ints[:, 5].copy(),
            "count": ints[:, 6].copy(), "meta": ints[:, 7].copy().view(np.uint32), "parts": parts.copy()}


class ForestDomains:
    """`world` domains on this GPU, stepped with the calls nbody_dd_step makes."""

    def __init__(self, ic, world, bits, theta, eps=0.01, dt=1e-3, records=T, cap_bytes=1024, group=64, f64=False,
                 flags=0, masses=None):
        self.parts = key_ranges(ic, world, bits)
        self.nb, self.pb = (48, 32) if f64 else (32, 16)       # bytes per node and particle record
        self.records, self.cap = records, cap_bytes           # a small capacity forces a regrow
        self.sims = []
        for r, idx in enumerate(self.parts):
            p = _params(len(idx), bits, theta, eps, dt)
            p.traversal_group_size = group
            p.flags |= flags
            if f64:
                p.flags |= nb.FLAG_FP64
            s = nb.Simulation(p)
            m = ic["m"][idx] if masses is None else np.full(len(idx), masses[r], ic["m"].dtype)   # one mass per domain
            s.set_state(ic["x"][idx], ic["y"][idx], ic["z"][idx], m, ic["vx"][idx],
                        ic["vy"][idx], ic["vz"][idx])
            s.set_traversal_range(0, 0)                       # update() = build only
            self.sims.append(s)

    def step(self, flat=False):
        W, R = len(self.sims), self.records
        for s in self.sims:
            s.update()
        self.frontiers = torch.zeros(W, R, nb.LET_REC, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        for r, s in enumerate(self.sims):
            s.let_frontier_dev(self.frontiers[r].data_ptr(), R)
            s.sync()
        self.copies, self.flat = [], []
        for r, s in enumerate(self.sims):
            while True:
                out = torch.zeros(W, self.cap, dtype=torch.uint8, device="cuda")
                cnt = torch.full((2 * W,), -1, dtype=torch.int32, device="cuda")
                torch.cuda.synchronize()
                s.let_export_tree_dev(self.frontiers.data_ptr(), W, R, r, out.data_ptr(), self.cap, cnt.data_ptr())
                s.sync()
                c = cnt.cpu().numpy().reshape(W, 2)
                need = int((self.nb * c[:, 0] + self.pb * c[:, 1]).max())
                if need <= self.cap:
                    break
                assert not out.any()                          # over capacity: nothing was written
                self.cap = (int(need * 1.25) + 64) // 16 * 16
            assert c[r, 0] == 0 and c[r, 1] == 0
            self.copies.append({q: (out[q, :self.nb * c[q, 0] + self.pb * c[q, 1]].clone(), int(c[q, 0]), int(c[q, 1]))
                                for q in range(W) if q != r})
            if flat:   # the flat export of the same tree against the same frontiers
                fcap = max(int(c[:, 1].max()), 1)
                fo = torch.zeros(W, fcap, 4, dtype=torch.float32, device="cuda")
                fc = torch.zeros(W, dtype=torch.int32, device="cuda")
                torch.cuda.synchronize()
                s.let_export_dev(self.frontiers.data_ptr(), W, R, r, fo.data_ptr(), fcap, fc.data_ptr())
                s.sync()
                fcn = fc.cpu().numpy()
                self.flat.append({q: fo[q, :fcn[q]].cpu().numpy() for q in range(W) if q != r})
        torch.cuda.synchronize()
        for r, s in enumerate(self.sims):
            inc = [self.copies[q][r] for q in range(W) if q != r]
            off, nn, nr, chunks = [], [], [], []
            at = 0
            for raw, a, b in inc:
                off.append(at); nn.append(a); nr.append(b); chunks.append(raw)
                at += raw.numel()
            recv = torch.cat(chunks).contiguous() if chunks else torch.zeros(16, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            s.import_trees_dev(recv.data_ptr(), off, nn, nr)
            s.traverse_forest()
            s.finish_step()
            s.sync()

    def close(self):
        for s in self.sims:
            s.close()


@pytest.mark.parametrize("kind,bits,world,records", [(nb.IC_SPHERE, 21, 2, 32), (nb.IC_DISK, 21, 3, 32),
                                                      (nb.IC_UNIFORM, 10, 4, 4), (nb.IC_PLUMMER, 21, 5, 1)])
def test_exported_copy_is_a_breadth_first_tree_over_the_flat_export(kind, bits, world, records):
    n, theta = 6000, 0.6
    ic = nb.sample_ic(kind, n, 11, p0=0.1, p1=1.0)
    d = ForestDomains(ic, world, bits, theta, records=records)
    d.step(flat=True)
    for r in range(world):
        for q in range(world):
            if q == r:
                continue
            raw, nn, npart = d.copies[r][q]
            t = parse_copy(raw.cpu().numpy(), nn, npart)
            assert nn >= 1
            leaf = (t["meta"] & LEAF) != 0
            inner = np.flatnonzero(~leaf)
            # children: contiguous blocks in the order of their parents, tiling [1, nn)
            assert np.array_equal(t["first"][inner], 1 + np.concatenate([[0], np.cumsum(t["count"][inner])[:-1]]))
            assert 1 + t["count"][inner].sum() == nn
            assert np.all((t["count"][inner] >= 2) & (t["count"][inner] <= 8))
            # leaves: their particle ranges tile [0, npart) in node order
            lv = np.flatnonzero(leaf)
            assert np.array_equal(t["first"][lv], np.concatenate([[0], np.cumsum(t["count"][lv])[:-1]]))
            assert t["count"][lv].sum() == npart
            # the child mask of an inner node marks children that need no test: single-record
            # leaves (nodes that pass for the whole domain); never an inner child
            for o in inner:
                kids = np.arange(t["first"][o], t["first"][o] + t["count"][o])
                may = sum(1 << k for k, c in enumerate(kids) if leaf[c] and t["count"][c] == 1)
                assert int(t["meta"][o]) & ~may == 0
                # a node's mass is its children's (float32 sums in the sender's order)
                assert np.isclose(t["com"][kids, 3].sum(), t["com"][o, 3], rtol=1e-5)
                assert np.all(t["size"][kids] <= t["size"][o] * 0.5)
            # the particle records are exactly the flat export's records
            assert np.array_equal(_rows_sorted(t["parts"]), _rows_sorted(d.flat[r][q]))
    d.close()


def _f32_fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def walk_copy(t, bx, by, bz, theta, domain, f64=False):
    """Sources (indices into t['parts'] and node ids accepted as (com, mass)) a group of bodies
    takes from a copy: a child its parent marks is taken whole without a test; any other node,
    leaf or not, is accepted as one source iff size^2 / min_b |x_b - com|^2 < theta^2 (float32,
    the kernel's operation order), else a leaf is taken whole and an inner node opened."""
    real = np.float64 if f64 else np.float32
    th2 = real(np.float32(theta)) * real(np.float32(theta))
    leaf = (t["meta"] & LEAF) != 0
    parts, coms = [], []
    stack = [(0, False)]
    while stack:
        o, untested = stack.pop()
        if untested:
            parts.extend(range(t["first"][o], t["first"][o] + t["count"][o]))
            continue
        dx, dy, dz = bx - t["com"][o, 0], by - t["com"][o, 1], bz - t["com"][o, 2]
        # (double: the kernel's fused multiply-adds round differently only in the last bit, which
        # decides a test once in about 1e13)
        d2 = dz * dz + (dy * dy + dx * dx) if f64 else _f32_fma(dz, dz, _f32_fma(dy, dy, dx * dx))
        dmin = min(real(3.0) * real(domain) * real(domain), d2.min())
        size = real(t["size"][o])
        with np.errstate(divide="ignore"):
            if (size * size) / dmin < th2:
                coms.append(o)
            elif leaf[o]:
                parts.extend(range(t["first"][o], t["first"][o] + t["count"][o]))
            else:
                stack.extend((t["first"][o] + k, bool((int(t["meta"][o]) >> k) & 1)) for k in range(t["count"][o]))
    return parts, coms


@pytest.mark.parametrize("group,world", [(64, 3), (32, 2)])
def test_forest_walk_equals_numpy_restatement(group, world):
    n, theta, eps, bits = 3000, 0.6, 0.02, 21
    ic = nb.sample_ic(nb.IC_SPHERE, n, 21)
    d = ForestDomains(ic, world, bits, theta, eps, group=group)
    d.step()
    for r, s in enumerate(d.sims):
        st = s.get_state()
        ll = s.get_list_len()
        b = O.full_build(st["x"], st["y"], st["z"], st["m"], DOMAIN, bits, 32)
        ax, ay, az, own_ll = O.traverse_from((st["x"], st["y"], st["z"]), b["tree"], b["com"], b["xs"], b["ys"],
                                             b["zs"], b["ms"], group, theta, eps, 1.0, DOMAIN)
        want = [ax.astype(np.float64), ay.astype(np.float64), az.astype(np.float64)]
        want_ll = own_ll.astype(np.int64).copy()
        copies = [parse_copy(raw.cpu().numpy(), nn, npart)
                  for q in range(world) if q != r for raw, nn, npart in [d.copies[q][r]]]
        nloc = len(st["x"])
        for g in range((nloc + group - 1) // group):
            sl = slice(g * group, min((g + 1) * group, nloc))
            bx, by, bz = st["x"][sl], st["y"][sl], st["z"][sl]
            for t in copies:
                parts, coms = walk_copy(t, bx, by, bz, theta, DOMAIN)
                src = np.concatenate([t["parts"][parts].reshape(-1, 4), t["com"][coms].reshape(-1, 4)]).astype(np.float64)
                want_ll[g] += len(src)
                dx = src[None, :, 0] - bx[:, None].astype(np.float64)
                dy = src[None, :, 1] - by[:, None].astype(np.float64)
                dz = src[None, :, 2] - bz[:, None].astype(np.float64)
                inv = src[None, :, 3] / (dx * dx + dy * dy + dz * dz + eps * eps) ** 1.5
                want[0][sl] += (dx * inv).sum(1); want[1][sl] += (dy * inv).sum(1); want[2][sl] += (dz * inv).sum(1)
        assert np.array_equal(ll, want_ll)
        scale = np.sqrt(want[0] ** 2 + want[1] ** 2 + want[2] ** 2)
        for got, w in zip((st["ax"], st["ay"], st["az"]), want):
            assert np.max(np.abs(got - w) / scale) < 1e-4
    d.close()


@pytest.mark.parametrize("world", [2, 4])
def test_forest_forces_match_direct_sum_and_single_domain(world):
    n, theta, eps = 20000, 0.6, 0.01
    ic = nb.sample_ic(nb.IC_SPHERE, n, 5)
    d = ForestDomains(ic, world, 21, theta, eps)
    d.step()
    x, y, z, m, ax, ay, az = (np.empty(n, np.float32) for _ in range(7))
    for idx, s in zip(d.parts, d.sims):
        st = s.get_state()
        g = idx[st["id"]]
        x[g], y[g], z[g], m[g] = st["x"], st["y"], st["z"], st["m"]
        ax[g], ay[g], az[g] = st["ax"], st["ay"], st["az"]
    d.close()
    ref = O.direct_f64(x, y, z, m, eps, 1.0)
    mean_let, max_let = O.acc_error(ax, ay, az, *ref)
    one = nb.Simulation(_params(n, 21, theta, eps))
    one.set_state(ic["x"], ic["y"], ic["z"], ic["m"], ic["vx"], ic["vy"], ic["vz"])
    one.update()
    s1 = one.get_state()
    p = s1["id"]
    assert np.array_equal(s1["x"], x[p])                  # same drift, same positions
    mean_one, max_one = O.acc_error(s1["ax"], s1["ay"], s1["az"], *[r[p] for r in ref])
    one.close()
    # the box criterion is more conservative than the per-group one: never much worse
    assert mean_let < 1.5 * mean_one and mean_let < 1e-2, (mean_let, mean_one)
    assert max_let < 3 * max_one + 5e-2, (max_let, max_one)


def test_forest_run_tracks_single_domain_run_and_regrows():
    """Five steps of three domains against one handle; the import room starts at nothing and the
    export buffer at 1 KB, so both grow on the way."""
    n, theta, steps = 8192, 0.5, 5
    ic = nb.sample_ic(nb.IC_DISK, n, 9)
    d = ForestDomains(ic, 3, 21, theta, 0.05, 1e-3)
    one = nb.Simulation(_params(n, 21, theta, 0.05, 1e-3))
    one.set_state(ic["x"], ic["y"], ic["z"], ic["m"], ic["vx"], ic["vy"], ic["vz"])
    for _ in range(steps):
        d.step()
        one.update()
    s1 = one.get_state()
    one.close()
    inv = np.empty(n, np.int64)
    inv[s1["id"]] = np.arange(n)
    for idx, s in zip(d.parts, d.sims):
        st = s.get_state()
        g = inv[idx[st["id"]]]
        for k in ("x", "y", "z"):
            assert np.max(np.abs(st[k] - s1[k][g])) < 1e-5
        for k in ("vx", "vy", "vz"):
            assert np.max(np.abs(st[k] - s1[k][g])) < 2e-3 * np.max(np.abs(s1[k]))
    d.close()


@pytest.mark.parametrize("masses,exact", [((2.0 ** -14,) * 3, True), ((1.0 / 20000,) * 3, False),
                                          ((3e-5, 7e-5, 5e-5), False)])
def test_forest_walk_of_domains_with_one_mass_each(masses, exact):
    """A handle whose own particles all have one mass leaves it out of their pair arithmetic, in
    the walk over own tree and copies too; the copies' records keep their masses (a pruned subtree
    travels as one record of its own mass, and another domain may have another mass).  Against
    the general kernel (NBODY_FLAG_NO_UNIFORM_MASS): same lists, the same bits for power-of-two
    masses, a few roundings otherwise."""
    n, world, theta, eps = 20000, 3, 0.6, 0.01
    ic = nb.sample_ic(nb.IC_SPHERE, n, 5)
    res = []
    for general in (True, False):
        d = ForestDomains(ic, world, 21, theta, eps, flags=nb.FLAG_NO_UNIFORM_MASS if general else 0, masses=masses)
        d.step()
        res.append([(s.get_state(), s.get_list_len().copy()) for s in d.sims])
        d.close()
    for (sa, la), (sb, lb) in zip(*res):
        assert np.array_equal(la, lb)
        assert np.array_equal(sa["id"], sb["id"])
        scale = np.sqrt(sa["ax"].astype(np.float64) ** 2 + sa["ay"] ** 2 + sa["az"] ** 2)
        for k in ("ax", "ay", "az"):
            if exact:
                assert np.array_equal(sa[k], sb[k]), k
            else:
                assert np.max(np.abs(sa[k].astype(np.float64) - sb[k]) / scale) < 1e-5, k


def test_forest_without_copies_is_the_plain_traversal():
    n = 8192
    ic = nb.sample_ic(nb.IC_SPHERE, n, 6)
    p = _params(n, 21, 0.5)
    with nb.Simulation(p) as a, nb.Simulation(p) as b:
        for s in (a, b):
            s.set_state(ic["x"], ic["y"], ic["z"], ic["m"])
        a.update()
        b.set_traversal_range(0, 0)
        b.update()
        b.import_trees_dev(0, [], [], [])
        b.traverse_forest()
        b.finish_step()
        sa, sb = a.get_state(), b.get_state()
        for k in ("ax", "ay", "az", "vx", "x"):
            assert np.array_equal(sa[k], sb[k])
        with pytest.raises(RuntimeError):
            b.import_trees_dev(0, [8], [1], [1])          # offsets are multiples of 16


def test_forest_of_double_handles_equals_numpy_restatement():
    """Double handles: copies carry doubles (48-byte nodes, 32-byte particles), the criterion and
    the sums run in double.  Lists exact, forces to 1e-11 against the oracle's double traversal
    of the own tree plus a numpy walk of the shipped copies."""
    n, theta, eps, bits, group, world = 3000, 0.6, 0.02, 21, 64, 3
    ic = nb.sample_ic(nb.IC_SPHERE, n, 21, dtype=np.float64)
    d = ForestDomains(ic, world, bits, theta, eps, group=group, f64=True)
    d.step()
    e2 = float(np.float32(eps)) ** 2                      # the handle keeps its parameters as floats
    for r, s in enumerate(d.sims):
        st = s.get_state()
        assert st["x"].dtype == np.float64
        ll = s.get_list_len()
        b = O.full_build_dp(st["x"], st["y"], st["z"], st["m"], DOMAIN, bits, 32)
        ax, ay, az, own_ll = O.traverse_dp(b["tree"], b["com"], b["xs"], b["ys"], b["zs"], b["ms"], group, theta,
                                           eps, 1.0, DOMAIN)
        want = [np.array(ax, np.float64), np.array(ay, np.float64), np.array(az, np.float64)]
        want_ll = np.array(own_ll, np.int64)
        copies = [parse_copy(raw.cpu().numpy(), nn, npart, True)
                  for q in range(world) if q != r for raw, nn, npart in [d.copies[q][r]]]
        nloc = len(st["x"])
        for g in range((nloc + group - 1) // group):
            sl = slice(g * group, min((g + 1) * group, nloc))
            bx, by, bz = st["x"][sl], st["y"][sl], st["z"][sl]
            for t in copies:
                parts, coms = walk_copy(t, bx, by, bz, theta, DOMAIN, True)
                src = np.concatenate([t["parts"][parts].reshape(-1, 4), t["com"][coms].reshape(-1, 4)])
                want_ll[g] += len(src)
                dx, dy, dz = (src[None, :, k] - c[:, None] for k, c in enumerate((bx, by, bz)))
                inv = src[None, :, 3] / (dx * dx + dy * dy + dz * dz + e2) ** 1.5
                want[0][sl] += (dx * inv).sum(1); want[1][sl] += (dy * inv).sum(1); want[2][sl] += (dz * inv).sum(1)
        assert np.array_equal(ll, want_ll)
        scale = np.sqrt(want[0] ** 2 + want[1] ** 2 + want[2] ** 2)
        for got, w in zip((st["ax"], st["ay"], st["az"]), want):
            assert np.max(np.abs(got - w) / scale) < 1e-11
    # against the all-pairs sum in double: the tree's own error, nothing else
    x, y, z, m, ax, ay, az = (np.empty(n, np.float64) for _ in range(7))
    for idx, s in zip(d.parts, d.sims):
        st = s.get_state()
        g = idx[st["id"]]
        x[g], y[g], z[g], m[g] = st["x"], st["y"], st["z"], st["m"]
        ax[g], ay[g], az[g] = st["ax"], st["ay"], st["az"]
    d.close()
    mean, mx = O.acc_error(ax, ay, az, *O.direct_f64(x, y, z, m, eps, 1.0))
    assert mean < 1e-2 and mx < 0.3

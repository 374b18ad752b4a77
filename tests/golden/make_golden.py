"""Writes tests/golden/*.npz: outputs of the host oracle (oracle/oracle.cpp) for small seeded
inputs, inputs included.  They pin the ORACLE against drift between revisions (and give the GPU
tests a fixed target); they are not outputs of the reference, which cannot be built or run
here and ships no vectors of its own -- parity with the reference stays unpinned.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as O  # noqa: E402


def inputs(kind, n, seed):
    """numpy-only generators, so that the fixtures do not depend on the library's samplers."""
    rng = np.random.default_rng(seed)
    if kind == "clusters":
        c = rng.uniform(1, 4, (5, 3))
        p = (c[rng.integers(0, 5, n)] + rng.normal(0, 0.05, (n, 3))).clip(0, 5)
    else:  # "shell": a thick spherical shell around the domain centre
        d = rng.normal(0, 1, (n, 3))
        p = 2.5 + d / np.linalg.norm(d, axis=1, keepdims=True) * rng.uniform(0.3, 0.5, (n, 1))
    v = rng.normal(0, 0.1, (n, 3))
    m = rng.uniform(0.5, 1.5, n)
    return p.astype(np.float32), v.astype(np.float32), m.astype(np.float32)


CASES = [  # name, kind, n, seed, bits, max_codes, group, theta, eps, dt
    ("clusters_b21_g64", "clusters", 3000, 1, 21, 32, 64, 0.5, 0.01, 1e-3),
    ("shell_b10_g32", "shell", 2500, 2, 10, 8, 32, 0.7, 0.05, 1e-3),
]


def make(case):
    name, kind, n, seed, bits, max_codes, group, theta, eps, dt = case
    p, v, m = inputs(kind, n, seed)
    b = O.full_build(p[:, 0], p[:, 1], p[:, 2], m, 5.0, bits, max_codes)
    ax, ay, az, ll = O.traverse(b["tree"], b["com"], b["xs"], b["ys"], b["zs"], b["ms"], group,
                                theta, eps, 1.0, 5.0)
    sim = O.OracleSim(p[:, 0], p[:, 1], p[:, 2], m, v[:, 0], v[:, 1], v[:, 2], bits=bits,
                      max_codes=max_codes, group=group, theta=theta, dt=dt, eps=eps)
    for _ in range(3):
        sim.step()
    s3 = sim.state()
    out = dict(pos=p, vel=v, mass=m,
               params=np.array([bits, max_codes, group], np.int32),
               fparams=np.array([theta, eps, dt, 5.0, 1.0], np.float32),
               codes_sorted=b["codes_sorted"], perm=b["perm"], unique=b["unique"],
               first_pt=b["first_pt"], leaf_first_code_idx=b["leaf_first_code_idx"],
               list_len=ll, ax=ax, ay=ay, az=az, counts_after_3=np.array(sim.counts()[:3], np.int64))
    for k, a in b["tree"].items():
        if k != "n_oct":
            out["tree_" + k] = a
    for k, a in b["com"].items():
        out["com_" + k] = a
    for k in ("x", "y", "z", "vx", "vy", "vz", "id"):
        out["after3_" + k] = s3[k]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "n_oct", b["tree"]["n_oct"], "interactions", int(ll.sum()))


def make_dp():
    """The oracle's double instantiation on double-precision inputs (not float-representable)."""
    rng = np.random.default_rng(3)
    n, bits, max_codes, group, theta, eps = 2000, 21, 16, 64, 0.6, 0.01
    c = rng.uniform(1, 4, (4, 3))
    p = (c[rng.integers(0, 4, n)] + rng.normal(0, 0.05, (n, 3))).clip(0, 5)
    m = rng.uniform(0.5, 1.5, n)
    b = O.full_build_dp(p[:, 0], p[:, 1], p[:, 2], m, 5.0, bits, max_codes)
    ax, ay, az, ll = O.traverse_dp(b["tree"], b["com"], b["xs"], b["ys"], b["zs"], b["ms"], group,
                                   theta, eps, 1.0, 5.0)
    out = dict(pos=p, mass=m, params=np.array([bits, max_codes, group], np.int32),
               fparams=np.array([theta, eps, 0.0, 5.0, 1.0], np.float32),
               codes_sorted=b["codes_sorted"], perm=b["perm"], leaf_first_code_idx=b["leaf_first_code_idx"],
               list_len=ll, ax=ax, ay=ay, az=az)
    for k, a in b["com"].items():
        out["com_" + k] = a
    for k in ("num_children", "points_begin", "points_end", "level"):
        out["tree_" + k] = b["tree"][k]
    np.savez_compressed(os.path.join(HERE, "double", "clusters_dp_b21_g64.npz"), **out)
    print("clusters_dp_b21_g64 n_oct", b["tree"]["n_oct"], "interactions", int(ll.sum()))


if __name__ == "__main__":
    for c in CASES:
        make(c)
    os.makedirs(os.path.join(HERE, "double"), exist_ok=True)
    make_dp()

import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    with open(os.path.join(GOLD, name + ".json")) as f:
        d = json.load(f)

    def fix(v):
        if isinstance(v, str) and v in ("nan", "inf", "-inf"):
            return float(v)
        if isinstance(v, list):
            return [fix(x) for x in v]
        return v

    d["result"] = {k: fix(v) for k, v in d["result"].items()}
    return d


def gpath(name):
    return os.path.join(GOLD, name)


def args_of(g):
    """decode the harness arguments of a fixture into python values"""
    a = g["args"]
    n = [int(x) for x in str(a["n"]).split(",")]
    out = dict(
        path=gpath(a["l"]), n=n, N=sum(n), Q=len(n),
        seed=int(a.get("d", 0)), dc=int(a.get("dc", 0)), beta=float(a.get("beta", 1.0)),
        crit=float(a.get("e", 5e-6)), lcrit=float(a.get("E", 1e-6)), tmax=int(a.get("t", 100)),
        damp=float(a.get("R", 1.0)), lr=float(a.get("r", 0.2)), init_flag=int(a.get("i", 0)),
    )
    if "eps" in a:
        out["eps"], out["c"] = float(a["eps"]), float(a["c"])
    else:
        out["pa"] = [float(x) for x in str(a["pa"]).split(",")]
        out["cab_upper"] = [float(x) for x in str(a["cab"]).split(",")]
    if "beliefs" in a:
        out["beliefs"] = np.loadtxt(gpath(a["beliefs"]), dtype=np.int32)
    if "nodes" in a:
        out["nodes"] = [int(x) for x in str(a["nodes"]).split(",")]
    out["true_conf"] = np.repeat(np.arange(len(n)), n).astype(np.uint32)
    return out


def best_perm_diff(psi, ref):
    """max |psi - ref| minimised over label permutations (label symmetry)"""
    import itertools
    Q = psi.shape[1]
    if Q > 6:  # Q! is out of reach: match columns by total absolute difference (assignment problem), then measure
        from scipy.optimize import linear_sum_assignment
        cost = np.array([[np.abs(psi[:, a] - ref[:, b]).sum() for a in range(Q)] for b in range(Q)])
        rows, cols = linear_sum_assignment(cost)  # ref column b <- psi column cols[b]
        p = tuple(int(c) for c in cols)
        return np.abs(psi[:, list(p)] - ref).max(), p
    best = None
    for p in itertools.permutations(range(Q)):
        d = np.abs(psi[:, list(p)] - ref).max()
        if best is None or d < best[0]:
            best = (d, p)
    return best


def best_vector_perm(v, ref):
    """permutation p minimising max |v[p] - ref| (exhaustive for small Q, assignment on |v_a - ref_b| above)"""
    import itertools
    Q = len(ref)
    if Q > 6:
        from scipy.optimize import linear_sum_assignment
        cost = np.abs(np.asarray(v)[None, :] - np.asarray(ref)[:, None])
        _, cols = linear_sum_assignment(cost)
        return [int(c) for c in cols]
    return list(min(itertools.permutations(range(Q)), key=lambda p: np.abs(np.asarray(v)[list(p)] - ref).max()))


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.lib()
    return oracle

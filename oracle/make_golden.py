#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — generates tests/golden/*.json from the compiled, untouched reference.

Run in the build container only (needs /root/reference and `make -C oracle ref`):

    python oracle/make_golden.py

Each fixture holds the harness arguments (inputs) and the reference's outputs at 17 significant
digits. Input graphs are data files: the reference's shipped dataset (copied verbatim as a data
fixture) and small synthetic graphs generated here with a fixed numpy seed.
"""
import json
import os
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
REF_BIN = os.path.join(HERE, "_ref", "bp_ref")
REF_DATASET = "/root/reference/dataset/N_1000-Q_2-method_cab_ec-eps_0.1-c_3.0.edgelist"


def run(cmd, **kw):
    argv = [REF_BIN, cmd] + ["%s=%s" % (k, v) for k, v in kw.items()]
    out = subprocess.run(argv, capture_output=True, text=True, check=True).stdout
    return json.loads(out)


def save(name, args, result, note=""):
    for k in ("load_s", "init_s", "converge_s", "fe_s", "entropy_s", "learn_s"):
        result.pop(k, None)
    args = dict(args)
    if "l" in args:
        args["l"] = os.path.basename(args["l"])  # fixtures refer to graphs by file name under tests/golden/
    with open(os.path.join(GOLD, name + ".json"), "w") as f:
        json.dump({"note": note, "args": args, "result": result}, f, separators=(",", ":"))
    print("wrote", name)


def planted_graph(N, Q, c, eps, seed):
    """sparse planted-partition graph, contiguous groups; returns unique undirected pairs (a<b)."""
    rng = np.random.default_rng(seed)
    cin = c * Q / ((Q - 1) * eps + 1)
    cout = eps * cin
    sizes = [N // Q] * Q
    sizes[-1] += N - sum(sizes)
    starts = np.cumsum([0] + sizes)
    pairs = []
    for r in range(Q):
        for s in range(r, Q):
            p = (cin if r == s else cout) / N
            npairs = sizes[r] * (sizes[r] - 1) / 2 if r == s else sizes[r] * sizes[s]
            m = rng.poisson(p * npairs)
            a = rng.integers(starts[r], starts[r + 1], m)
            b = rng.integers(starts[s], starts[s + 1], m)
            pairs.append(np.stack([a, b], 1))
    e = np.concatenate(pairs)
    e = e[e[:, 0] != e[:, 1]]
    e = np.unique(np.sort(e, 1), axis=0)
    return e, cin, cout


def hub_graph(N, Q, seed):
    """small degree-corrected graph with a few hubs (degree >= 50) for the large-degree path."""
    rng = np.random.default_rng(seed)
    theta = np.minimum(rng.pareto(1.5, N) + 1.0, 60.0)
    theta[:3] = [90.0, 75.0, 60.0]  # guaranteed hubs
    grp = np.arange(N) * Q // N
    w = theta / theta.sum()
    m = int(2.2 * N)
    a = rng.choice(N, 6 * m, p=w)
    b = rng.choice(N, 6 * m, p=w)
    keep = (grp[a] == grp[b]) | (rng.random(6 * m) < 0.15)
    e = np.stack([a[keep], b[keep]], 1)[:m]
    e = e[e[:, 0] != e[:, 1]]
    return np.unique(np.sort(e, 1), axis=0)


def write_edgelist(path, e):
    with open(path, "w") as f:
        for a, b in e:
            f.write("%d %d\n" % (a, b))


def gen_q10():
    """Q = 10 (above the reference's Q! overlap search, bp.cpp:784-790: identity labelling only): N = 1000, c = 16, eps = 0.02"""
    e, cin, cout = planted_graph(1000, 10, 16.0, 0.02, 21)
    q10 = os.path.join(GOLD, "q10_n1000.edgelist")
    write_edgelist(q10, e)
    cabu = []
    for r_ in range(10):
        for s_ in range(r_, 10):
            cabu.append(cin if r_ == s_ else cout)
    cabs = ",".join(repr(float(x)) for x in cabu)
    b = dict(l=q10, n=",".join(["100"] * 10), pa=",".join(["0.1"] * 10), cab=cabs, t=2000)
    # seed 1: the asynchronous run reaches the planted fixed point (f = -21.25); seeds 0, 2, 6 end in a fixed point
    # with two groups merged (f = -20.17 ... -20.19) - which basin a run falls into is schedule dependent
    a = dict(b, d=1, e="1e-13", dump="psi")
    save("q10_tight_seed1", a, run("infer", **a), "synthetic planted graph, Q=10")
    a = dict(b, d=1, e="1e-13", mode="learn")
    save("q10_em_expect_seed1", a, run("em_expect", **a))
    a = dict(b, d=1, nodes="0,5,100,999")
    save("q10_node_update_seed1", a, run("node_update", **a))


def main():
    if not os.path.exists(REF_BIN):
        subprocess.check_call(["make", "-C", HERE, "ref"])
    os.makedirs(GOLD, exist_ok=True)
    if sys.argv[1:] == ["q10"]:  # only the Q = 10 fixtures (the others are unchanged)
        return gen_q10()
    c1 = os.path.join(GOLD, "c1_dataset.edgelist")
    shutil.copyfile(REF_DATASET, c1)  # data fixture (the reference's only shipped input)

    save("rng_seed0", {"d": 0}, run("rng", d=0), "SURVEY Appendix E: first draws of std::mt19937(0) through uniform_real_distribution")
    save("rng_seed7", {"d": 7}, run("rng", d=7))

    base = dict(l=c1, n="500,500", pa="0.5,0.5")
    readme = dict(base, cab="3.63,2.36,3.63", t=1000, i=0)
    matched = dict(base, cab="5.4545454545454541,0.54545454545454541,5.4545454545454541", t=1000)
    for sd in (0, 1, 2):
        a = dict(readme, d=sd)
        save("c1_readme_infer_seed%d" % sd, a, run("infer", **a), "README.md:36 command (below detectability: overlap 0.5)")
    a = dict(matched, d=0, e="5e-6")
    save("c1_matched_default_seed0", a, run("infer", **a))
    a = dict(matched, d=0, e="1e-13", dump="psi")
    save("c1_matched_tight_seed0", a, run("infer", **a), "fixed-point golden: psi is N*Q row-major")
    a = dict(matched, d=5, e="1e-13")
    save("c1_matched_tight_seed5", a, run("infer", **a))
    a = dict(matched, d=0, e="1e-13", dump="msg")
    save("c1_matched_tight_seed0_msg", a, run("infer", **a), "msg_in is the reference's in-ordered mmap_[i][l][q]")
    a = dict(matched, d=0, e="1e-12", R="0.5")
    save("c1_matched_damped_seed0", a, run("infer", **a), "dumping_rate 0.5")
    a = dict(matched, d=0, e="1e-12", beta="0.8", dump="psi")
    save("c1_matched_beta08_seed0", a, run("infer", **a), "beta != 1 (small-degree path semantics)")

    dcab = "0.60606060606060608,0.060606060606060608,0.60606060606060608"
    for dc in (1, 2):
        a = dict(base, cab=dcab, t=1000, d=0, dc=dc, e="1e-13", dump="psi", quiet=1)
        save("c1_dc%d_tight_seed0" % dc, a, run("infer", **a), "deg_corr_flag=%d; entropy is NaN in the reference (B11)" % dc)
        a = dict(base, cab=dcab, t=1000, d=0, dc=dc, quiet=1)
        save("c1_dc%d_default_seed0" % dc, a, run("infer", **a))

    # learn mode (README.md:41) and an informative start
    a = dict(base, cab="3.63,2.36,3.63", t=1000, d=0)
    save("c1_readme_learn_seed0", a, run("learn", **a), "README.md:41")
    a = dict(base, cab="5,1,5", t=1000, d=0)
    save("c1_learn_515_seed0", a, run("learn", **a))
    a = dict(base, cab="5,1,5", t=1000, d=3)
    save("c1_learn_515_seed3", a, run("learn", **a))

    # EM expectations on the tight fixed point
    a = dict(base, cab="5.4222500510078531,0.56205351675988069,5.4376514988570044", t=1000, d=0, e="1e-13", mode="learn", dump="psi")
    save("c1_em_expect_seed0", a, run("em_expect", **a), "SURVEY Appendix D EM golden parameters")
    for dc in (1, 2):
        a = dict(base, cab=dcab, t=1000, d=0, e="1e-13", dc=dc, mode="learn", quiet=1)
        save("c1_em_expect_dc%d_seed0" % dc, a, run("em_expect", **a))

    # single node updates from the seeded initial state (no schedule): degree 0, small, the max-degree node
    a = dict(matched, d=0, nodes="0,1,2,3,17,500,999", dump="psi")
    save("c1_node_update_seed0", a, run("node_update", **a), "bp_iter_update_psi on listed nodes, in order, after init_h()")
    a = dict(matched, d=0, nodes="0,1,2,3,17,500,999", large=1, quiet=1)
    save("c1_node_update_large_seed0", a, run("node_update", **a), "same nodes through bp_iter_update_psi_large_degree")
    for dc in (1, 2):
        a = dict(base, cab=dcab, d=0, dc=dc, nodes="0,1,2,3,17,500,999")
        save("c1_node_update_dc%d_seed0" % dc, a, run("node_update", **a))
    a = dict(matched, d=0, e="1e-30", t=3, dump="psi")
    save("c1_three_sweeps_seed0", a, run("converge", **a), "3 asynchronous sweeps (schedule + RNG stream check)")

    # clamped nodes: -i 1 with a beliefs file (bp_conditional skips planted rows)
    beliefs = -np.ones(1000, dtype=int)
    beliefs[:50] = 0
    beliefs[500:550] = 1
    bpath = os.path.join(GOLD, "c1_beliefs.txt")
    np.savetxt(bpath, beliefs, fmt="%d")
    a = dict(matched, d=0, e="1e-13", i=1, beliefs=bpath, dump="psi")
    r = run("infer", **a)
    a["beliefs"] = "c1_beliefs.txt"
    save("c1_planted_i1_seed0", a, r, "-i 1 with 100 clamped nodes")

    # Q=4 synthetic graph (N=400, c=6, eps=0.05): pins Q>2 paths
    e, cin, cout = planted_graph(400, 4, 6.0, 0.05, 11)
    q4 = os.path.join(GOLD, "q4_n400.edgelist")
    write_edgelist(q4, e)
    cabu = []
    for r_ in range(4):
        for s_ in range(r_, 4):
            cabu.append(cin if r_ == s_ else cout)
    cabs = ",".join(repr(float(x)) for x in cabu)
    b4 = dict(l=q4, n="100,100,100,100", pa="0.25,0.25,0.25,0.25", cab=cabs, t=2000)
    a = dict(b4, d=0, e="1e-13", dump="psi")
    save("q4_tight_seed0", a, run("infer", **a), "synthetic planted graph, Q=4")
    a = dict(b4, d=0, e="1e-13", mode="learn")
    save("q4_em_expect_seed0", a, run("em_expect", **a))
    a = dict(b4, d=1, nodes="0,5,100,399")
    save("q4_node_update_seed1", a, run("node_update", **a))
    a = dict(l=q4, n="100,100,100,100", eps="0.05", c="6.0", t=2000, d=0)
    save("q4_epsc_default_seed0", a, run("infer", **a), "--epsilon_c path (blockmodel.cpp:229-272)")
    a = dict(l=q4, n="100,100,100,100", pa="0.25,0.25,0.25,0.25", cab=",".join(repr(float(x)) for x in [7, 1.5, 1, 0.5, 6, 1, 1.2, 8, 0.7, 5]), t=300, d=2, E="1e-7")
    save("q4_learn_seed2", a, run("learn", **a))

    # hub graph: rows with degree >= 50 take the reference's log-domain path
    e = hub_graph(600, 3, 5)
    hg = os.path.join(GOLD, "hub_n600.edgelist")
    write_edgelist(hg, e)
    a = dict(l=hg, n="200,200,200", pa="0.3333333333333333,0.3333333333333333,0.3333333333333333",
             cab="0.05,0.008,0.008,0.05,0.008,0.05", dc=1, t=2000, d=0, e="1e-12", dump="psi", quiet=1)
    save("hub_dc1_tight_seed0", a, run("infer", **a), "power-law graph with hubs, deg_corr_flag=1")
    a = dict(l=hg, n="200,200,200", pa="0.3333333333333333,0.3333333333333333,0.3333333333333333",
             cab="9,1.5,1.5,9,1.5,9", dc=0, t=2000, d=0, e="1e-12", dump="psi", quiet=1)
    save("hub_dc0_tight_seed0", a, run("infer", **a), "same graph, plain SBM")
    # This instance has three BP fixed points and the reference's own result depends on its seed: of seeds 0..40, 26 end at
    # f = -0.83246 (0, 2, 5, 6, 7, ...), 13 at f = -0.84669 (1, 3, 4, 8, 11, ...), 2 at f = -0.82785 (23, 39). All three are
    # "where the reference lands".
    for d in (1, 23):
        a = dict(a, d=d)
        save("hub_dc0_tight_seed%d" % d, a, run("infer", **a), "same graph, plain SBM, another fixed point the reference reaches")
    gen_q10()


if __name__ == "__main__":
    sys.exit(main())

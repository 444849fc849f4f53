"""Debug aid: one instance of tests/test_gpu_fuzz.py::test_random_instance_learning_against_the_oracle, EM step by EM step,
engine against the oracle's synchronous EM (same rules): sweeps of each BP run, free energy, parameters.
usage: python tools/trace_learn_instance.py SEED [max_steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import sbm_bp_amd as S
import oracle as orc
from sbm_bp_amd import synth

seed = int(sys.argv[1])
nmax = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rng = np.random.default_rng(7000 + seed)
Q = int(rng.choice([2, 3, 4, 6]))
N = int(rng.choice([120, 300, 600])) // Q * Q
c = float(rng.choice([4.0, 7.0, 10.0]))
eps = float(rng.choice([0.05, 0.15, 0.3]))
pairs, cin, cout = synth.planted_partition(N, Q, c, eps, 100 + seed)
tc = synth.true_conf(N, Q)
cab0 = synth.cab_matrix(Q, cin * rng.uniform(0.7, 1.3), cout * rng.uniform(0.7, 1.6))
na = np.array(synth.group_sizes(N, Q), dtype=np.uint32)
lr, lcrit, tmax = float(rng.choice([0.2, 0.5])), 1e-6, 60
print(dict(Q=Q, N=N, c=c, eps=eps, lr=lr))
g = S.Graph.from_edges(pairs, N)
og = orc.Graph.from_edges(pairs, N)
bp = S.bp_basic()
bp.init_messages(S.blockmodel_t(g, Q, 0), 0, None, tc, seed)
bp.expand_bp_params(S.bp_blockmodel_state(cab0, na))
bp.set_schedule(0.3, 1)
ob = orc.OracleBP(og, Q, 0)
ob.init_messages(0, None, tc, orc.Rng(seed))
ob.set_params(cab0, na, 1.0)
ob.set_field_mix(0.3)
crit = np.float32(lcrit)
fold, fdiff = 0.0, 1.0
cab, nac = cab0.copy(), na.copy()
for t in range(min(tmax, nmax)):
    if fdiff < float(crit):
        crit = np.float32(float(crit) * 0.1)
    n1, l1 = bp.converge(float(crit), tmax, 1.0)
    n2, l2 = ob.converge_sync(float(crit), tmax, 1.0)
    p1, m1 = bp.get_state()
    p2, m2 = ob.get_state()
    e1, e2 = bp.em_expectations(), ob.em_expect()
    f1, f2 = bp.compute_free_energy(), ob.free_energy(0)[0]
    print("step %2d crit %.0e niter %3d/%3d last %.2e/%.2e |dpsi| %.1e |dmsg| %.1e f %.12f/%.12f |dcab_e| %.1e" % (
        t, float(crit), n1, n2, l1, l2, np.abs(p1 - p2).max(), np.abs(m1 - m2).max(), f1, f2, np.abs(e1[2] - e2[2]).max()))
    fdiff, fold = abs(f2 - fold), f2
    if fdiff < float(crit):
        break
    # the oracle's values drive both (so a divergence shows where it STARTS, not its amplification)
    snap = min(1.0 * N * float(crit), 0.01)
    n_new = nac.astype(np.int64).copy()
    rest = N
    for i in range(Q - 1):
        n_new[i] = int(lr * e2[0][i] + (1 - lr) * nac[i] + snap)
        rest -= n_new[i]
    n_new[-1] = rest
    cab = lr * e2[2] + (1 - lr) * cab
    nac = n_new.astype(np.uint32)
    bp.expand_bp_params(S.bp_blockmodel_state(cab, nac))
    ob.set_params(cab, nac, 1.0)

"""Check, not product code: the oscillating fixture `hub_dc0` at a realistic size. A power-law graph (the C4 generator) under the
PLAIN model (deg_corr_flag 0) is where synchronous sweeps swing. Runs the compiled reference (oracle/_ref/bp_ref, its own
random-sequential schedule, one host core per seed) for four seeds and the engine with nothing but the reference's flags from
the reference's own initial states, then a few fixed schedules, and compares the fixed points reached (overlap, free energy,
largest difference between the marginals). Finding (profiles/r03_powerlaw_dc0_multistability.log): the instance has many BP
fixed points - the reference's four seeds end in four different ones.
  python3 tools/check_powerlaw_dc0.py"""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import sbm_bp_amd as S
from sbm_bp_amd import synth
S.load_library()
N, Q, c, eps, gseed = 200000, 4, 8.0, 0.1, 11
pairs, _, c_eff = synth.dc_sbm_powerlaw(N, Q, c, eps, gseed)
cin, cout = synth.cin_cout(Q, c_eff, eps)
cab = synth.cab_matrix(Q, cin, cout)
sizes = synth.group_sizes(N, Q); tc = synth.true_conf(N, Q)
g = S.Graph.from_edges(pairs, N)
ref_bin = os.path.join(ROOT, "oracle", "_ref", "bp_ref")
refpsi = {}
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "g.bin")
    np.ascontiguousarray(pairs, dtype=np.uint32).tofile(path)
    cabu = [cin if r == s else cout for r in range(Q) for s in range(r, Q)]
    procs = []
    for seed in (0, 1, 2, 3):
        argv = [ref_bin, "infer", "l=" + path, "n=" + ",".join(map(str, sizes)), "pa=" + ",".join(repr(1.0 / Q) for _ in range(Q)),
                "cab=" + ",".join(repr(float(x)) for x in cabu), "d=%d" % seed, "e=1e-9", "t=1000", "skip_fe=1", "dump=psi", "quiet=1"]
        procs.append((seed, subprocess.Popen(argv, stdout=subprocess.PIPE, text=True)))
    for seed, p in procs:
        r = json.loads(p.communicate()[0])
        refpsi[seed] = np.array(r["psi"]).reshape(N, Q)
        print("reference seed", seed, "niter", r["niter"], "overlap", r["overlap"], flush=True)
for a in (0, 1, 2, 3):
    for b in range(a + 1, 4):
        print("reference seeds", a, b, "marginals differ by", float(np.abs(refpsi[a] - refpsi[b]).max()))
def engine(seed, auto=True, mix=None, damp=1.0):
    bp = S.bp_conditional()
    bp.init_messages(S.blockmodel_t(g, Q, 0), 0, None, tc, seed)
    bp.expand_bp_params(S.bp_blockmodel_state(cab, np.array(sizes, dtype=np.uint32)))
    if not auto: bp.set_auto_relax(False)
    if mix is not None: bp.set_schedule(field_mix=mix) if hasattr(bp, "set_schedule") else None
    niter, last = bp.converge(1e-9, 2000, damp)
    psi = bp.get_state()[0]
    f = bp.compute_free_energy()
    d = [float(np.abs(psi - refpsi[s]).max()) for s in refpsi]
    print("engine seed", seed, "auto", auto, "mix", mix, "damp", damp, "niter", niter, "relax", bp.relaxation(), "overlap", round(bp.compute_overlap(), 4), "f", f, "diff to ref seeds", ["%.1e" % x for x in d], flush=True)
for seed in (0, 1, 2, 3):
    engine(seed)
engine(0, auto=False, mix=0.1)
engine(0, auto=False, mix=0.25)
engine(0, auto=False, damp=0.5)
engine(0, auto=True, damp=0.5)

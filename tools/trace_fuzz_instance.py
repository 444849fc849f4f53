"""Debug aid: run one instance of tests/test_gpu_fuzz.py sweep by sweep, engine against the synchronous oracle, and show
where they part. usage: python tools/trace_fuzz_instance.py SEED [max_sweeps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import sbm_bp_amd as S
import oracle as orc
import test_gpu_fuzz as tf

seed = int(sys.argv[1])
nmax = int(sys.argv[2]) if len(sys.argv) > 2 else 40
t = tf._instance(seed)
Q, N, dc = t["Q"], t["N"], t["dc"]
print({k: t[k] for k in ("Q", "N", "dc", "beta", "damp", "flag")}, "cab", np.array2string(t["cab"], precision=3))
g = S.Graph.from_edges(t["pairs"], N)
og = orc.Graph.from_edges(t["pairs"], N)
bp = S.bp_conditional()
bp.init_messages(S.blockmodel_t(g, Q, dc), t["flag"], t["conf"], t["tc"], t["seed"])
bp.set_beta(t["beta"])
bp.expand_bp_params(S.bp_blockmodel_state(t["cab"], t["na"]))
ob = orc.OracleBP(og, Q, dc)
ob.init_messages(t["flag"], t["conf"], t["tc"], orc.Rng(t["seed"]))
ob.set_params(t["cab"], t["na"], t["beta"])
row_ptr, nbr, rev = g.csr()
for k in range(nmax):
    damp = t["damp"] if k < 2 else 1.0
    d1, d2 = bp.sweep(1, damp), ob.sweep_sync(damp)
    psi, msg = bp.get_state()
    opsi, omsg = ob.get_state()
    bad_e, bad_o = int(np.isnan(msg).any(1).sum()), int(np.isnan(omsg).any(1).sum())
    dm = np.nanmax(np.abs(msg - omsg)) if msg.size else 0.0
    print("sweep %2d: diff engine %.3e oracle %.3e | max |msg - omsg| %.3e | NaN messages engine %d oracle %d | psi-form sweeps %d" % (
        k, d1, d2, dm, bad_e, bad_o, bp.stats().psi_form_sweeps))
    if bad_e or dm > 1e-9:
        kbad = int(np.nanargmax(np.abs(msg - omsg).max(1))) if not bad_e else int(np.flatnonzero(np.isnan(msg).any(1))[0])
        i = int(np.searchsorted(row_ptr, kbad, side="right") - 1)
        print("   first bad message: edge %d = row %d (degree %d) -> %d: engine %s oracle %s" % (kbad, i, row_ptr[i + 1] - row_ptr[i], nbr[kbad], msg[kbad], omsg[kbad]))
        print("   incoming messages of that row (oracle, previous sweep not kept): psi engine %s oracle %s" % (psi[i], opsi[i]))
        break

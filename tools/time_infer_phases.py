"""Measurement aid: where `-m infer` spends its time on the device at a bench workload (converge, free energy, entropy, overlap)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import sbm_bp_amd as S
from sbm_bp_amd import synth
from bench import WORKLOADS

wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
if len(sys.argv) > 2:  # "separate": the round-2 kernels (k_fe_frame, k_nonedge_adj, k_em_edges) instead of the fused pass
    os.environ["SBMBP_FUSED_REDUCTIONS"] = "0" if sys.argv[2] == "separate" else "1"
N, Q, c, eps, dc, gseed = WORKLOADS[wl]
S.load_library()
if wl == "C4":
    pairs, cab, _ = synth.dc_sbm_powerlaw(N, Q, c, eps, gseed)
else:
    pairs, cin, cout = synth.planted_partition(N, Q, c, eps, gseed)
    cab = synth.cab_matrix(Q, cin, cout)
g = S.Graph.from_edges(pairs, N)
del pairs
bm = S.blockmodel_t(g, Q, dc)
bp = S.bp_conditional()
bp.init_messages_device(bm, synth.true_conf(N, Q), 1234)
bp.expand_bp_params(S.bp_blockmodel_state(cab, np.array(synth.group_sizes(N, Q), dtype=np.uint32)))


def timed(f, reps=1):  # (one repetition: the fused reduction pass keeps its results until the state changes)
    out = None
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        out = f()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    return out, best * 1e3


t = time.perf_counter()
niter, last = bp.converge(5e-6, 1000, 1.0)
torch.cuda.synchronize()
conv_ms = (time.perf_counter() - t) * 1e3
f, f_ms = timed(bp.compute_free_energy)
e, e_ms = timed(bp.compute_entropy)
o, o_ms = timed(bp.compute_overlap)
em, em_ms = timed(bp.em_expectations)
print("%s: converge %d sweeps %.1f ms | free energy %.2f ms | entropy %.2f ms | overlap %.2f ms | EM expectations %.2f ms" % (
    wl, niter + 1, conv_ms, f_ms, e_ms, o_ms, em_ms))
print("   f = %.9f  e = %.9f  overlap = %.6f" % (f, e, o))

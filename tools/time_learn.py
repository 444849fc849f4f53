"""Measurement aid: wall-clock breakdown of -m learn at config C5 (N=1e6, Q=4, c=5)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sbm_bp_amd as S
from sbm_bp_amd import synth
N, Q, c, eps = 1_000_000, 4, 5.0, 0.1
pairs, cin, cout = synth.planted_partition(N, Q, c, eps, 4)
g = S.Graph.from_edges(pairs, N)
bm = S.blockmodel_t(g, Q, 0)
na = np.array(synth.group_sizes(N, Q), dtype=np.uint32)
bp = S.bp_basic()
bp.init_messages_device(bm, synth.true_conf(N, Q), 7)
st = S.bp_blockmodel_state(synth.cab_matrix(Q, 0.9 * cin, 1.8 * cout), na)
bp.set_schedule(1.0, 8)
def t(f, n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, r
bp.expand_bp_params(st)
ms, _ = t(lambda: bp.sweep(10, 1.0, want_diff=False), 3); print("10 sweeps: %.3f ms (%.3f ms/sweep)" % (ms, ms / 10))
ms, _ = t(lambda: bp.converge(1e-6, 1000, 1.0), 1); print("converge to 1e-6: %.2f ms" % ms, bp.stats().sweeps, "sweeps total")
ms, f = t(bp.compute_free_energy); print("free energy: %.3f ms" % ms, f)
ms, _ = t(bp.em_expectations); print("em expectations: %.3f ms" % ms)
ms, _ = t(bp.compute_overlap); print("overlap: %.3f ms" % ms)
ms, _ = t(bp.compute_entropy); print("entropy: %.3f ms" % ms)
bp.init_messages_device(bm, synth.true_conf(N, Q), 7)
bp.reset_stats()
torch.cuda.synchronize(); t0 = time.perf_counter()
res = bp.learning(bm, st, 1e-6, 100, 0.2, 1.0)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
cab, na1 = bp.get_params()
print("learning: %.1f ms, %d EM steps, %d sweeps, status %d, overlap %.4f" % (dt * 1e3, res.em_steps, res.total_sweeps, res.status, res.overlap))
print("learned cin %.4f (true %.4f) cout %.4f (true %.4f)" % (np.diag(cab).mean(), cin, (cab.sum() - np.trace(cab)) / (Q * Q - Q), cout))
print("edge-msg/s over the whole learn run: %.3e" % (res.total_sweeps * g.E2 / dt))

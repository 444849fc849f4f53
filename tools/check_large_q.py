"""Measurement/verification aid: engine vs synchronous oracle for Q in 9..16 on a small planted graph (3 sweeps at
1e-12, then free energy / entropy / EM expectations at 1e-9), and sweep throughput at N = 1e6 for Q = 12, 16."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import sbm_bp_amd as S
from sbm_bp_amd import synth
import oracle as orc

bad = 0
for Q in (9, 11, 12, 13, 16):
    N = 60 * Q
    pairs, cin, cout = synth.planted_partition(N, Q, 14.0, 0.03, 7)
    tc = synth.true_conf(N, Q)
    cab, na = synth.cab_matrix(Q, cin, cout), np.array(synth.group_sizes(N, Q), dtype=np.uint32)
    for dc in (0, 1):
        g = S.Graph.from_edges(pairs, N)
        bm = S.blockmodel_t(g, Q, dc)
        bp = S.bp_basic()
        bp.init_messages(bm, 0, None, tc, 3)
        scale = 1.0 if dc == 0 else 1.0 / 14.0 ** 2
        bp.expand_bp_params(S.bp_blockmodel_state(cab * scale, na))
        og = orc.Graph.from_edges(pairs, N)
        ob = orc.OracleBP(og, Q, dc)
        ob.init_messages(0, None, tc, orc.Rng(3))
        ob.set_params(cab * scale, na, 1.0)
        worst = 0.0
        for _ in range(3):
            d1, d2 = bp.sweep(1, 1.0), ob.sweep_sync(1.0)
            worst = max(worst, abs(d1 - d2), np.abs(bp.get_state()[0] - ob.get_state()[0]).max(), np.abs(bp.get_state()[1] - ob.get_state()[1]).max())
        it, _ = bp.converge(1e-13, 3000, 1.0)
        it2, _ = ob.converge_sync(1e-13, 3000, 1.0)
        f, fo = bp.compute_free_energy(), ob.free_energy(0)[0]
        na1, nna1, cab1 = bp.em_expectations()
        na2, nna2, cab2 = ob.em_expect()
        ov, ovo = bp.compute_overlap(), ob.overlap()
        em = max(np.abs(na1 - na2).max(), np.abs(cab1 - cab2).max() / max(1.0, np.abs(cab2).max()))
        ok = worst < 1e-12 and abs(f - fo) < 1e-9 * max(1, abs(fo)) and em < 1e-7 and abs(ov - ovo) < 1e-9 and abs(it - it2) <= 1
        bad += not ok
        print("Q=%2d dc=%d: sweeps |diff| %.1e, niter %d/%d, f %.12f/%.12f, EM %.1e, overlap %.6f/%.6f %s" % (
            Q, dc, worst, it, it2, f, fo, em, ov, ovo, "ok" if ok else "MISMATCH"), flush=True)
for Q in (12, 16):
    N = 1_000_000
    pairs, cin, cout = synth.planted_partition(N, Q, 10.0, 0.05, 9)
    g = S.Graph.from_edges(pairs, N)
    bm = S.blockmodel_t(g, Q, 0)
    bp = S.bp_conditional()
    bp.init_messages_device(bm, synth.true_conf(N, Q), 5)
    bp.expand_bp_params(S.bp_blockmodel_state(synth.cab_matrix(Q, cin, cout), np.array(synth.group_sizes(N, Q), dtype=np.uint32)))
    bp.sweep(3, 1.0, want_diff=False)
    bp.reset_stats(); bp.set_timing(True)
    t = time.perf_counter(); bp.sweep(20, 1.0, want_diff=False); dt = time.perf_counter() - t
    st = bp.stats()
    km = st.sweep_kernel_ms / max(1, st.sweep_launches)
    print("Q=%d N=1e6 c=10: %.3e edge-msg/s, %.3f ms per sweep, kernel %.3f ms = %.0f GB/s algorithmic" % (
        Q, 20 * g.E2 / dt, dt * 50, km, st.bytes_per_sweep / km / 1e6), flush=True)
sys.exit(1 if bad else 0)

import sys, numpy as np
sys.path.insert(0, "/root/repo")
import sbm_bp_amd as S
from sbm_bp_amd import synth
S.load_library()
for (N, Q) in ((50_000, 32), (50_000, 64), (200_000, 17)):
    pairs, cin, cout = synth.planted_partition(N, Q, 8.0, 0.05, 5)
    g = S.Graph.from_edges(pairs, N)
    bp = S.bp_conditional()
    bp.init_messages_device(S.blockmodel_t(g, Q, 0), synth.true_conf(N, Q), 1234)
    bp.expand_bp_params(S.bp_blockmodel_state(synth.cab_matrix(Q, cin, cout), np.array(synth.group_sizes(N, Q), dtype=np.uint32)))
    d = bp.sweep(5, 1.0)
    psi = bp.real_psi()
    print(N, Q, "E2", g.E2, "diff after 5 sweeps", d, "row sums ok", bool(np.abs(psi.sum(1) - 1).max() < 1e-12), flush=True)

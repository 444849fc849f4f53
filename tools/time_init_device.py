import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import sbm_bp_amd as S
from sbm_bp_amd import synth
from bench import WORKLOADS
N, Q, c, eps, dc, gseed = WORKLOADS["C3"]
S.load_library()
pairs, cin, cout = synth.planted_partition(N, Q, c, eps, gseed)
g = S.Graph.from_edges(pairs, N); del pairs
bm = S.blockmodel_t(g, Q, dc); bp = S.bp_conditional()
tc = synth.true_conf(N, Q)
bp.init_messages_device(bm, tc, 1234)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): bp.reinit_messages_device(tc, 1234)
torch.cuda.synchronize(); print("init_messages_device at C3: %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
bp.expand_bp_params(S.bp_blockmodel_state(synth.cab_matrix(Q, cin, cout), np.array(synth.group_sizes(N, Q), dtype=np.uint32)))
print("converge", bp.converge(5e-6, 100, 1.0))

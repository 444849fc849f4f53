"""Measurement aid: host-side cost of queueing one sharded sweep (ctypes + collective calls) against the GPU time it
queues, per shard, at 1/8 of C3 per shard (eight HIP shards in one process on one GPU, LocalComm)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import sbm_bp_amd as S
from sbm_bp_amd import synth
from sbm_bp_amd.distributed import LocalComm, ShardedBP
from bench import WORKLOADS

N, Q, c, eps, dc, gseed = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
torch.cuda.set_device(0)
S.load_library()
pairs, cin, cout = synth.planted_partition(N, Q, c, eps, gseed)
g = S.Graph.from_edges(pairs, N)
del pairs
row_ptr, nbr, _ = g.csr()
del g
sb = ShardedBP.from_csr(row_ptr, nbr, Q, dc, LocalComm(world), n_chunks=4)
sb.init_messages_device(1234, synth.true_conf(N, Q))
sb.expand_bp_params(synth.cab_matrix(Q, cin, cout), np.array(synth.group_sizes(N, Q), dtype=np.uint32), 1.0)
sb.sweep(3, 1.0, want_diff=False)
n = 10
sb._begin(-1.0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for j in range(n):
    sb._queue_sweep(j)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
for sh in sb.shards:
    sh.poll()
    sh.commit(n)
print("world %d: host queues one sweep of ALL %d shards in %.3f ms (%.3f ms per shard); GPU finishes it in %.3f ms (%.3f ms per shard)" % (
    world, world, (t1 - t0) * 1e3 / n, (t1 - t0) * 1e3 / n / world, (t2 - t0) * 1e3 / n, (t2 - t0) * 1e3 / n / world))
# the same with the collectives that a real rank issues: count the calls one shard makes per sweep
calls = 4 + 4 + 4 + 1 + 1 + 1 + 1
print("one rank issues %d calls per sweep (4 sweep chunks, 4 packs, 4 all-to-all, unpack, fold, all-gather, finalize)" % calls)

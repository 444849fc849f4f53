"""Rehearsal of the 8-rank plan at FULL C3 size on one GPU: eight ranks of the C++ driver as threads of this process
(in-process transport) against one rank, same counter-based initial state. Checks index widths, halo plans and chunking
at the size the 8-GPU bench runs, and prints what one rank ships (bytes per peer).
usage: python tools/check_shards_full_size.py [workload=C3] [world=8] [sweeps=6]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import sbm_bp_amd as S
from sbm_bp_amd import synth
from sbm_bp_amd.distributed import LocalShards
from bench import WORKLOADS

wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sweeps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
N, Q, c, eps, dc, gseed = WORKLOADS[wl]
S.load_library()
t0 = time.perf_counter()
if wl == "C4":
    pairs, cab_c4, _ = synth.dc_sbm_powerlaw(N, Q, c, eps, gseed)
else:
    pairs, cin, cout = synth.planted_partition(N, Q, c, eps, gseed)
g = S.Graph.from_edges(pairs, N)
del pairs
print("graph: N=%d E2=%d in %.1f s" % (N, g.E2, time.perf_counter() - t0), flush=True)
tc = synth.true_conf(N, Q)
cab = cab_c4 if wl == "C4" else synth.cab_matrix(Q, cin, cout)
na = np.array(synth.group_sizes(N, Q), dtype=np.uint32)


def run(w):
    t = time.perf_counter()
    sb = LocalShards(g, Q, dc, w, n_chunks=int(os.environ.get("SBMBP_SHARD_CHUNKS", "4")) if w > 1 else 1)
    sb.init_messages_device(1234, tc)
    sb.expand_bp_params(cab, na, 1.0)
    print("world %d: plans + shards in %.1f s" % (w, time.perf_counter() - t), flush=True)
    if w > 1:
        for sh in sb.ranks:
            sc = sh.peer_rows()[0].astype(float) * sh.info.halo_components * 8 / 1e6
            print("  rank %d: rows %d edges %d halo %d cut-edge records %d | MB per sweep to peers: %s" % (
                sh.comm.rank, sh.n_own, sh.n_edges, sh.n_halo, sh.info.n_halo_msgs, " ".join("%.1f" % x for x in sc)), flush=True)
    d = sb.sweep(2, 1.0, want_diff=True)
    t = time.perf_counter()
    d = sb.sweep(sweeps, 1.0, want_diff=True)
    dt = time.perf_counter() - t
    print("world %d: %d sweeps, %.2f ms per sweep (all ranks on one GPU), max diff %.3e" % (w, sweeps, dt * 1e3 / sweeps, d), flush=True)
    samples = sb.confusion()
    ov = sb.compute_overlap()
    fe = sb.compute_free_energy()
    sb.close()
    return d, samples, ov, fe


d1, s1, ov1, fe1 = run(1)
dk, sk, ovk, fek = run(world)
print("max diff: %.17g vs %.17g" % (d1, dk))
print("overlap: %.15f vs %.15f   free energy: %.15f vs %.15f" % (ov1, ovk, fe1, fek))
worst = float(np.abs(s1 - sk).max() / N)
print('confusion matrix / N: worst |diff| = %.3e' % worst)
ok = worst < 1e-12 and abs(d1 - dk) < 1e-12 and abs(ov1 - ovk) < 1e-12 and abs(fe1 - fek) < 1e-10 * max(1, abs(fe1))
print("PARTITION INVARIANT" if ok else "MISMATCH")
sys.exit(0 if ok else 1)

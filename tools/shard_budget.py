"""Measurement aid: what ONE rank of a W-rank plan executes per sweep at full workload size, alone on the GPU.
Rank `rank` of `world` runs the C++ driver over the null transport (sbmbp_comm_init_null: exchanges deliver nothing, every
peer "reports" this rank's reduction values), so the kernel work is exactly the rank's share of the multi-GPU run and
nothing else runs on the device. Reported: HIP-event time per sweep on the rank's compute stream, split into the chunk
kernels and fold + all-gather stand-in + finalize (sbmbp_dist_phase_times).

  python3 tools/shard_budget.py [workload=C3] [world=8] [rank=0] [sweeps=20] [out.json]
  knobs: SBMBP_SHARD_CHUNKS, SBMBP_SHARD_STREAMS (1 = chunks on one stream), SBMBP_SHARD_XCD, SBMBP_HALO_COMPRESS
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(wl="C3", world=8, rank=0, sweeps=20, out=None):
    import numpy as np
    import sbm_bp_amd as S
    from bench import WORKLOADS, source_sha
    from sbm_bp_amd import synth
    from sbm_bp_amd.capi import check
    from sbm_bp_amd.distributed import Comm, ShardedBP
    N, Q, c, eps, dc, gseed = WORKLOADS[wl]
    lib = S.load_library()
    if wl == "C4":
        pairs, cab, _ = synth.dc_sbm_powerlaw(N, Q, c, eps, gseed)
    else:
        pairs, cin, cout = synth.planted_partition(N, Q, c, eps, gseed)
        cab = synth.cab_matrix(Q, cin, cout)
    g = S.Graph.from_edges(pairs, N)
    del pairs
    h = C.c_void_p()
    check(lib.sbmbp_comm_init_null(C.byref(h), world, rank))
    sb = ShardedBP(g, Q, dc, Comm(h.value), device=0)
    sb.init_messages_device(1234, synth.true_conf(N, Q))
    sb.expand_bp_params(cab, np.array(synth.group_sizes(N, Q), dtype=np.uint32), 1.0)
    info = sb.info
    sb.sweep(5, 1.0, want_diff=False)
    import time
    import torch
    sb.set_timing(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sb.sweep(sweeps, 1.0, want_diff=False)   # wall clock of the same loop without the phase events: does the host keep up?
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) / sweeps * 1e3
    sb.set_timing(True)
    sb.reset_stats()
    sb.sweep(sweeps, 1.0, want_diff=False)
    ph = sb.phase_times()
    res = {"what": "HIP-event time per sweep of ONE rank of the plan, alone on the GPU (tools/shard_budget.py, null transport)",
           "workload": wl, "world": world, "rank": rank, "rows": int(info.n_own), "edges": int(info.n_edges), "halo_rows": int(info.n_halo),
           "chunks": int(info.n_chunks), "streams": os.environ.get("SBMBP_SHARD_STREAMS", "2"), "sweeps": sweeps,
           "chunk_kernels_ms": round(ph["chunks_ms"], 4), "fold_gather_finalize_ms": round(ph["reduce_ms"], 4),
           "per_sweep_ms": round(ph["chunks_ms"] + ph["reduce_ms"], 4), "wall_ms_per_sweep_untimed": round(wall_ms, 4),
           "source_sha": source_sha()}
    print(json.dumps(res), flush=True)
    if out:
        json.dump(res, open(out, "w"), indent=1)
    sb.close()


if __name__ == "__main__":
    a = sys.argv[1:]
    main(a[0] if a else "C3", int(a[1]) if len(a) > 1 else 8, int(a[2]) if len(a) > 2 else 0, int(a[3]) if len(a) > 3 else 20,
         a[4] if len(a) > 4 else None)

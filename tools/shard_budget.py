"""Measurement aid: what ONE rank of a W-rank plan executes per sweep at full workload size, alone on the GPU.
Rank `rank` of `world` runs the C++ driver with a stand-in transport (callbacks that deliver constant halo rows), so the
kernel work is exactly the rank's share of the multi-GPU run while the kernel durations are not disturbed by other ranks.

  python3 tools/shard_budget.py run  [workload=C3] [world=8] [rank=0] [sweeps=10]       # under rocprofv3 --kernel-trace --stats
  python3 tools/shard_budget.py sum  <rocprof dir> <sweeps> <out.json> [label]         # per-sweep kernel budget by kind
"""
import csv
import ctypes as C
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(wl="C3", world=8, rank=0, sweeps=10):
    import numpy as np
    import sbm_bp_amd as S
    from bench import WORKLOADS
    from sbm_bp_amd import synth
    from sbm_bp_amd.capi import CommCallbacks, check
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

    def exchange(user, send, send_rows, recv, recv_rows, width):  # every halo row arrives as the uniform marginal
        n = sum(int(recv_rows[p]) for p in range(world)) * width
        if n:
            np.ctypeslib.as_array(recv, shape=(n,))[:] = 1.0 / Q
        return 0

    def allgather(user, inp, n, out):  # every rank reports what this one does
        a = np.ctypeslib.as_array(inp, shape=(int(n),))
        np.ctypeslib.as_array(out, shape=(int(n) * world,))[:] = np.tile(a, world)
        return 0

    def allreduce(user, buf, n, op):
        return 0

    cb = CommCallbacks(None, CommCallbacks.EXCHANGE(exchange), CommCallbacks.ALLGATHER(allgather), CommCallbacks.ALLREDUCE(allreduce))
    h = C.c_void_p()
    check(lib.sbmbp_comm_init_callbacks(C.byref(h), world, rank, C.byref(cb)))
    comm = Comm(h.value, keep=cb)
    sb = ShardedBP(g, Q, dc, comm, device=0)
    sb.init_messages_device(1234, synth.true_conf(N, Q))
    sb.expand_bp_params(cab, np.array(synth.group_sizes(N, Q), dtype=np.uint32), 1.0)
    info = sb.info
    print("rank %d of %d: rows %d edges %d halo %d chunks %d" % (rank, world, info.n_own, info.n_edges, info.n_halo, info.n_chunks), flush=True)
    sb.sweep(3, 1.0, want_diff=False)
    sb.set_timing(True)
    sb.reset_stats()
    sb.sweep(sweeps, 1.0, want_diff=False)
    st = sb.stats()
    print("BUDGET_MARK sweeps=%d sweep_kernel_ms_per_sweep=%.4f" % (sweeps, st.sweep_kernel_ms / max(1, st.sweep_launches)), flush=True)
    sb.close()


def summarise(d, sweeps_total, out, label=""):
    from bench import source_sha
    f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)[0]
    per = {}
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        key = next((k for k in ("k_sweep_psi", "k_sweep<", "k_fold_stage", "k_finalize", "k_pack_rows", "k_unpack_rows", "k_psi_sum") if k in n), None)
        if key:
            per[key] = per.get(key, 0.0) + float(r["TotalDurationNs"]) / 1e6
    res = {"what": "kernel time per sweep of ONE rank, alone on the GPU (tools/shard_budget.py; the run has 3 warm-up sweeps: totals are divided by sweeps + 3)",
           "label": label, "source_sha": source_sha(),
           "ms_per_sweep": {k: round(v / sweeps_total, 4) for k, v in per.items()}}
    res["sweep_path_ms"] = round(sum(res["ms_per_sweep"].get(k, 0.0) for k in ("k_sweep_psi", "k_fold_stage", "k_finalize")), 4)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        a = sys.argv[2:]
        run(a[0] if a else "C3", int(a[1]) if len(a) > 1 else 8, int(a[2]) if len(a) > 2 else 0, int(a[3]) if len(a) > 3 else 10)
    else:
        summarise(sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else "")

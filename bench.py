#!/usr/bin/env python3
"""bench.py — BP edge-message updates/sec of the synchronous sweep (BASELINE.json metric).

A "step" is one synchronous sweep over the whole synthetic planted-partition graph, state
resident in HBM. Default workload: the configuration the north-star target is quoted on,
N=1e7, Q=4, c=10 (BASELINE configs[2]; it fits one MI355X). Other workloads: --workload C2|C4|C5|small.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task description): value = whole-job edge-message
updates per second; roofline = algorithmic bytes per sweep / HIP-event time of the sweep kernel;
cpu_baseline = the reference's converge() (oracle/_ref/bp_ref, or the oracle port) on a bounded
sample of the same graph family, timed on this box's host cores.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (N, Q, c, eps, dc, graph seed)   — SURVEY 8(d)
    "C3": (10_000_000, 4, 10.0, 0.1, 0, 2),
    "C2": (1_000_000, 2, 3.0, 0.1, 0, 1),
    "C5": (1_000_000, 4, 5.0, 0.1, 0, 4),
    "small": (200_000, 4, 10.0, 0.1, 0, 9),
    # ten times C3 on one GPU (E2 = 1e9 < 2^32; ~63 GB of the 288 GB): not a SURVEY configuration, a capacity check
    "C3x10": (100_000_000, 4, 10.0, 0.1, 0, 2),
    # degree-corrected SBM, power-law propensities, --deg_corr_flag 1 (hub rows take the workgroup-per-row kernel)
    "C4": (1_000_000, 8, 8.0, 0.1, 1, 3),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def cpu_baseline(Q, c, eps, sample_n=200_000, sweeps=8):
    """reference converge() on one host core, on a bounded sample of the same graph family"""
    from sbm_bp_amd import synth
    pairs, cin, cout = synth.planted_partition(sample_n, Q, c, eps, 12345)
    e2 = 2 * len(pairs)
    sample = "planted partition N=%d Q=%d c=%g eps=%g (E2=%d), %d asynchronous sweeps, converge() only" % (
        sample_n, Q, c, eps, e2, sweeps)
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "bp_ref")
    cabu = []
    for r in range(Q):
        for s in range(r, Q):
            cabu.append(cin if r == s else cout)
    if os.path.exists(ref_bin) and os.access(ref_bin, os.X_OK):
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "sample.bin")
            np.ascontiguousarray(pairs, dtype=np.uint32).tofile(path)
            sizes = synth.group_sizes(sample_n, Q)
            argv = [ref_bin, "converge", "l=" + path, "n=" + ",".join(map(str, sizes)),
                    "pa=" + ",".join(repr(1.0 / Q) for _ in range(Q)), "cab=" + ",".join(repr(float(x)) for x in cabu),
                    "d=0", "e=0", "t=%d" % sweeps, "quiet=1"]
            out = subprocess.run(argv, capture_output=True, text=True, check=True).stdout
            r = json.loads(out)
            return {"value": r["edge_msg_per_s"], "unit": "edge-msg/s", "cores": 1, "kind": "reference", "sample": sample}
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    g = orc.Graph.from_edges(pairs, sample_n)
    bp = orc.OracleBP(g, Q, 0)
    rng = orc.Rng(0)
    bp.init_messages(0, None, synth.true_conf(sample_n, Q), rng)
    bp.set_params(synth.cab_matrix(Q, cin, cout), np.array(synth.group_sizes(sample_n, Q), dtype=np.uint32), 1.0)
    t0 = time.perf_counter()
    bp.converge_async(0.0, sweeps, 1.0, rng)
    dt = time.perf_counter() - t0
    return {"value": sweeps * g.E2 / dt, "unit": "edge-msg/s", "cores": 1, "kind": "port", "sample": sample}


def pmc_traffic(workload, kernel, n_gpus, E2, N, Q):
    """HBM-side bytes per launch from the committed rocprofv3 PMC summary of the same command
    (profiles/*_pmc_*.json; bench.py cannot collect PMC counters itself). Corrected as
    MI355X_MICROARCH.md §HBM prescribes: FETCH_SIZE/WRITE_SIZE are KiB; on gfx950 wide coalesced
    streaming reads are tallied at half (128-B requests counted as 64 B), random 32-B gathers cost one
    64-B request each and are counted in full; writes read exactly. See DESIGN.md §4."""
    import glob
    if n_gpus != 1:
        return None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_*.json")), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("kernel") == kernel and d.get("workload", "").startswith(workload + " "):
            c = d["counters"]
            stream_reads = E2 * (8.0 * (Q - 1) + 4.0) + N * 4.0  # own old message record (Q-1 components) + index per edge, row offsets
            return c["FETCH_SIZE"]["per_launch_mean"] * 1024.0 + 0.5 * stream_reads + c["WRITE_SIZE"]["per_launch_mean"] * 1024.0
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--converge", action="store_true", default=True,
                    help="also report sweeps-to-converge at 5e-6 (untimed, after the timed region; default on)")
    ap.add_argument("--no-converge", dest="converge", action="store_false")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the sharded driver (torch.distributed collectives) even with one rank")
    ap.add_argument("--gather", default="auto", choices=["auto", "messages"],
                    help="sweep form: auto = marginal-gather when exact, messages = always gather messages")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    # SBMBP_REHEARSAL=1 (development aid, never the measured configuration): all ranks share cuda:0 and the
    # collectives go over gloo through host memory, so the multi-process driver can run on a 1-GPU box
    rehearsal = os.environ.get("SBMBP_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    sharded = world > 1 or args.force_sharded
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import sbm_bp_amd as S
    from sbm_bp_amd import synth
    S.load_library()

    N, Q, c, eps, dc, gseed = WORKLOADS[args.workload]
    t0 = time.perf_counter()
    if not sharded:
        if args.workload == "C4":
            pairs, cab_mat, c_eff = synth.dc_sbm_powerlaw(N, Q, c, eps, gseed)
        else:
            pairs, cin, cout = synth.planted_partition(N, Q, c, eps, gseed)
            cab_mat = synth.cab_matrix(Q, cin, cout)
        g = S.Graph.from_edges(pairs, N)
        del pairs
        bm = S.blockmodel_t(g, Q, dc)
        bp = S.bp_conditional(device=local_rank)
        bp.init_messages_device(bm, synth.true_conf(N, Q), 1234)
        bp.expand_bp_params(S.bp_blockmodel_state(cab_mat, np.array(synth.group_sizes(N, Q), dtype=np.uint32)))
        if args.gather == "messages":
            bp.set_gather_mode(1)
        E2_total = g.E2
        runner = bp
    else:
        from sbm_bp_amd.distributed import ShardedBP, HostStagedComm
        runner = ShardedBP.synthetic(N, Q, c, eps, gseed, dc=dc, seed=1234, comm=HostStagedComm() if rehearsal else None)
        E2_total = runner.E2_global
    setup_s = time.perf_counter() - t0

    def barrier():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    runner.set_timing(False)
    runner.sweep(args.warmup, 1.0, want_diff=False)
    runner.reset_stats()
    runner.set_timing(True)
    barrier()
    t1 = time.perf_counter()
    runner.sweep(args.steps, 1.0, want_diff=False)
    barrier()
    dt = time.perf_counter() - t1
    if sharded:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    st = runner.stats()
    phases = runner.phase_times() if sharded else None
    kernel_ms = st.sweep_kernel_ms / max(1, st.sweep_launches)
    bytes_per_launch = st.bytes_per_sweep  # this rank's rows/edges: what ONE launch of k_sweep processes
    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0

    sweeps_to_converge = None
    if args.converge:
        runner.set_timing(False)
        niter, _ = runner.converge(5e-6, 1000, 1.0)
        sweeps_to_converge = (args.warmup + args.steps + niter + 1) if niter >= 0 else -1

    if rank == 0:
        kname = ("k_sweep_psi<%d>" if st.psi_form_sweeps else "k_sweep<%d>") % Q
        out = {
            "metric": "BP edge-message updates/sec", "value": args.steps * E2_total / dt, "unit": "edge-msg/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: planted SBM N=%d Q=%d c=%g eps=%g deg_corr=%d, synchronous BP sweep (-m infer inner loop)" % (
                args.workload, N, Q, c, eps, dc), "N": N, "Q": Q, "E2": int(E2_total), "hub_rows": int(st.n_hub_rows), "parallelism": "vertex-range shards x%d%s" % (world, " (sharded driver)" if sharded else ""),
                "setup_s": round(setup_s, 2)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(args.workload, kname, world, E2_total, N, Q),
                         "kernel": kname, "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": bytes_per_launch},
        }
        if sharded:  # what rank 0 ships per sweep: the numbers needed to read a multi-GPU result
            p0, sh0 = runner.plans[0], runner.shards[0]
            per_peer = p0.send_counts.astype(float) * sh0.ncomp * 8 / 1e6
            out["config"]["exchange"] = {"chunks": int(p0.n_chunks), "payload_components": int(sh0.ncomp),
                                         "halo_rows": int(p0.n_halo), "sent_MB_per_sweep": round(float(per_peer.sum()), 2),
                                         "busiest_peer_MB_per_sweep": round(float(per_peer.max()) if len(per_peer) else 0.0, 2)}
            if phases:  # rank 0's stream: chunk kernels / fold + all-gather + finalize / wait for exchanges still in flight
                out["config"]["exchange"]["rank0_ms_per_sweep"] = {k: round(v, 4) for k, v in phases.items() if k != "sweeps"}
        if rehearsal:
            out["rehearsal"] = "ranks share cuda:0, gloo collectives staged through the host: not a measurement"
        if sweeps_to_converge is not None:
            out["sweeps_to_converge"] = sweeps_to_converge
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(Q, c, eps)  # plain planted partition of the same Q, c (also for C4)
        print(json.dumps(out), flush=True)
    if sharded:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py — BP edge-message updates/sec of the synchronous sweep (BASELINE.json metric).

A "step" is one synchronous sweep over the whole synthetic planted-partition graph, state
resident in HBM. Default workload: the configuration the north-star target is quoted on,
N=1e7, Q=4, c=10 (BASELINE configs[2]; it fits one MI355X). Other workloads: --workload C2|C4|C5|small.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N ...            # bare shell: starts its own N ranks (one process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task description): value = whole-job edge-message
updates per second over the timed steady-state sweeps; `converge` = one converge(5e-6) from the initial
state timed between barriers (the metric as SURVEY 8(d) words it: sweeps x E2 / wall time of the converge
phase); roofline = algorithmic bytes per sweep / HIP-event time of the sweep kernel; cpu_baseline = the
reference's converge() (oracle/_ref/bp_ref, or the oracle port) on a bounded sample of the same graph
family, timed on this box's host cores.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

# the host driver of this pool only supports dmabuf IPC: without this RCCL (and any sharing of device memory between the rank
# processes) fails with "hipIpcGetMemHandle: invalid argument". Set before anything initialises HIP; an exported value wins.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

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
    "C3x20": (200_000_000, 4, 10.0, 0.1, 0, 2),  # capacity: E2 = 2.0e9 directed edges, ~126 GB of HBM
    # degree-corrected SBM, power-law propensities, --deg_corr_flag 1 (hub rows: fragments of 256 edges, two launches of their own)
    "C4": (1_000_000, 8, 8.0, 0.1, 1, 3),
    # the same Q and mean degree as C4 on a plain planted partition (Poisson degrees): separates what Q = 8 costs from what the
    # power-law degrees cost
    "Q8": (1_000_000, 8, 8.0, 0.1, 0, 3),
    "Q8dc": (1_000_000, 8, 8.0, 0.1, 1, 3),
    "Q12": (1_000_000, 12, 8.0, 0.1, 0, 8),
    "Q16": (1_000_000, 16, 8.0, 0.1, 0, 9),
    # label counts above 16: the matrix-core kernels (csrc/kernels_wide.h; message-gather form, full Q-component records)
    "Q32": (1_000_000, 32, 8.0, 0.1, 0, 5),
    "Q48": (500_000, 48, 8.0, 0.1, 0, 7),
    "Q64": (500_000, 64, 8.0, 0.1, 0, 6),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
CONV_CRIT = 5e-6       # the reference's default -e (main.cpp:113)


def source_sha():
    """identifies the kernel sources a number was measured on (the GPU box has no .git): profiles/*.json carry it"""
    h = hashlib.sha256()
    for f in ("kernels.h", "kernels_wide.h", "engine.hip", "dist.hip", "host_graph.cpp"):
        p = os.path.join(ROOT, "sbm-bp_amd", "csrc", f)
        if os.path.exists(p):
            h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def host_info():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"nproc": os.cpu_count(), "cpu_model": model}


def cpu_baseline(Q, c, eps, sample_n=1_000_000, sweeps=4, gseed=12345, full=None):
    """the reference's converge() on ONE host core (the reference is single-threaded). Default: a bounded sample of the
    same graph family (same Q, c, eps at N = 1e6: the reference needs ~220 B of heap per directed edge and ~25 s per sweep
    at the benchmarked N = 1e7). `full` = a workload name: the benchmarked graph ITSELF (same generator seed, same
    parameters, BP seed 0), a fixed number of sweeps, as SURVEY 8(d) words it (--cpu-baseline full: minutes, ~20 GB)."""
    import numpy as np
    from sbm_bp_amd import synth
    pairs, cin, cout = synth.planted_partition(sample_n, Q, c, eps, gseed)
    e2 = 2 * len(pairs)
    sample = "%splanted partition N=%d Q=%d c=%g eps=%g (E2=%d), %d asynchronous sweeps, converge() only" % (
        ("the %s graph itself: " % full) if full else "", sample_n, Q, c, eps, e2, sweeps)
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "bp_ref")
    cabu = []
    for r in range(Q):
        for s in range(r, Q):
            cabu.append(cin if r == s else cout)
    host = host_info()
    if os.path.exists(ref_bin) and os.access(ref_bin, os.X_OK):
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "sample.bin")
            np.ascontiguousarray(pairs, dtype=np.uint32).tofile(path)
            del pairs
            sizes = synth.group_sizes(sample_n, Q)
            argv = [ref_bin, "converge", "l=" + path, "n=" + ",".join(map(str, sizes)),
                    "pa=" + ",".join(repr(1.0 / Q) for _ in range(Q)), "cab=" + ",".join(repr(float(x)) for x in cabu),
                    "d=0", "e=0", "t=%d" % sweeps, "quiet=1"]
            t0 = time.perf_counter()
            out = subprocess.run(argv, capture_output=True, text=True, check=True).stdout
            wall = time.perf_counter() - t0
            r = json.loads(out)
            res = {"value": r["edge_msg_per_s"], "unit": "edge-msg/s", "cores": 1, "kind": "reference", "sample": sample,
                   "host": host}
            for k in ("load_s", "init_s", "converge_s"):  # the reference's own phases: text/binary load, init_messages, converge()
                if k in r:
                    res[k] = r[k]
            res["wall_s"] = round(wall, 1)
            return res
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
    return {"value": sweeps * g.E2 / dt, "unit": "edge-msg/s", "cores": 1, "kind": "port", "sample": sample, "host": host}


def pmc_traffic(workload, kernel, n_gpus, E2, N, Q):
    """HBM-side bytes per launch from the committed rocprofv3 PMC summary of the same command
    (profiles/*_pmc_*.json; bench.py cannot collect PMC counters itself). Corrected as
    MI355X_MICROARCH.md §HBM prescribes: FETCH_SIZE/WRITE_SIZE are KiB; on gfx950 wide coalesced
    streaming reads are tallied at half (128-B requests counted as 64 B), random 32-B gathers cost one
    64-B request each and are counted in full; writes read exactly. See DESIGN.md §4.
    Returns (traffic or None, note): the summary is only used when it was collected on THESE kernel sources."""
    import glob
    if n_gpus != 1:
        return None, None
    sha = source_sha()
    stale = None
    stream_reads = E2 * (8.0 * (Q - 1) + 4.0) + N * 4.0  # own old message record (Q-1 components) + index per edge, row offsets
    if Q > 16:  # k_wsweep: both records of an edge (own and reverse, Q components each) are contiguous runs of 8Q >= 136 bytes
        stream_reads = E2 * (2 * 8.0 * Q + 4.0) + N * 4.0
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc*.json")), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        c = None
        if "kernels" in d:  # tools/pmc_workload.sh (round 3 on): one file per workload, every sweep-path kernel in it
            if d.get("workload") != workload:
                continue
            for k, e in d["kernels"].items():
                kk, want = k.replace(" ", ""), kernel.replace(" ", "")
                if kk.startswith(want.replace(">", ",")) or kk.startswith(want):
                    c = e["counters"]
        elif d.get("kernel") == kernel and d.get("workload", "").startswith(workload + " "):  # round 1/2 files
            c = d["counters"]
        if c is None or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        if d.get("source_sha") != sha:
            stale = stale or "%s was collected on kernel sources %s, this build is %s" % (
                os.path.relpath(path, ROOT), d.get("source_sha", "(unrecorded)"), sha)
            continue
        return (c["FETCH_SIZE"]["per_launch_mean"] * 1024.0 + 0.5 * stream_reads + c["WRITE_SIZE"]["per_launch_mean"] * 1024.0,
                "from %s (same kernel sources)" % os.path.relpath(path, ROOT))
    return None, stale


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(n):
    """`python bench.py --gpus N` from a bare shell: start N fresh rank processes (one per GPU) with the environment
    torch.distributed.run would give them. This parent never touches the GPU and never re-execs itself; rank 0's JSON
    line reaches stdout through the inherited descriptor. All children are polled together: the first one that exits
    non-zero (an out-of-memory rank, a failed RCCL init, its own watchdog) takes the others down - they would sit in a
    collective waiting for it - and its code is the parent's. SIGTERM / SIGINT to the parent do the same."""
    import signal
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SBMBP_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))

    def stop_all(grace=10.0):
        for q in procs:
            if q.poll() is None:
                q.terminate()
        t_end = time.monotonic() + grace
        for q in procs:
            while q.poll() is None and time.monotonic() < t_end:
                time.sleep(0.05)
            if q.poll() is None:
                q.kill()
        for q in procs:
            q.wait()

    signalled = []

    def on_signal(signum, _frame):
        signalled.append(signum)

    for sig in (signal.SIGTERM, signal.SIGINT):
        signal.signal(sig, on_signal)
    rc = 0
    try:
        while True:
            codes = [q.poll() for q in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if signalled:
                rc = 128 + signalled[0]
                break
            if bad:
                rc = bad[0]
                break
            if all(c == 0 for c in codes):
                break
            time.sleep(0.1)
    finally:
        stop_all()
    return rc


class Watchdog:
    """Per-phase deadline inside every rank process (it works the same under torch.distributed.run, where this file is not
    the parent). A daemon thread that never touches the GPU: when a phase outlives its deadline - a rank stuck in an RCCL
    collective whose peer died, a hung kernel - it prints ONE diagnostic JSON line naming the phase and ends the process with
    code 3; the launcher (self_launch above, or torchrun) then takes the other ranks down. Nothing is re-executed.
    SBMBP_PHASE_DEADLINE_S scales every deadline (default 1.0; 0 switches the watchdog off)."""
    DEADLINES = {"import": 600, "rendezvous": 300, "setup": 1800, "chunk_trials": 1800, "warmup": 300, "timed": 600,
                 "converge": 900, "cpu_baseline": 1800, "report": 300}

    def __init__(self, rank, world):
        import threading
        self.rank, self.world = rank, world
        self.scale = float(os.environ.get("SBMBP_PHASE_DEADLINE_S", "1.0"))
        self.phase, self.t0 = "import", time.monotonic()
        self.lock = threading.Lock()
        if self.scale > 0:
            threading.Thread(target=self._run, daemon=True).start()

    def enter(self, phase):
        with self.lock:
            self.phase, self.t0 = phase, time.monotonic()

    def _run(self):
        while True:
            time.sleep(1.0)
            with self.lock:
                phase, el = self.phase, time.monotonic() - self.t0
            if phase == "done":
                return
            limit = self.DEADLINES.get(phase.split(":")[0], 600) * self.scale
            if el > limit:
                sys.stdout.write(json.dumps({"error": "watchdog", "phase": phase, "rank": self.rank, "n_gpus": self.world,
                                             "elapsed_s": round(el, 1), "deadline_s": limit,
                                             "note": "a phase outlived its deadline; this rank exits with code 3 and the launcher ends the others"}) + "\n")
                sys.stdout.flush()
                os._exit(3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", default="sample", choices=["sample", "full", "none"],
                    help="sample (default): the reference's converge() on N = 1e6 of the same family, ~10 s; full: on the "
                         "benchmarked graph itself for 3 sweeps (SURVEY 8(d); minutes and ~20 GB at C3); none")
    ap.add_argument("--converge", action="store_true", default=True,
                    help="also time one converge(5e-6) from the initial state (after the timed sweeps; default on)")
    ap.add_argument("--no-converge", dest="converge", action="store_false")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the sharded driver (RCCL collectives) even with one rank")
    ap.add_argument("--gather", default="auto", choices=["auto", "messages"],
                    help="sweep form: auto = marginal-gather when exact, messages = always gather messages")
    ap.add_argument("--dry-run", action="store_true",
                    help="ranks rendezvous, build their shard plans and cross-check them; no GPU work, no measurement")
    args = ap.parse_args()

    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dog = Watchdog(rank, world)
    import numpy as np
    import torch
    import torch.distributed as dist

    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # SBMBP_REHEARSAL=1 (development aid, never the measured configuration): all ranks share cuda:0 and the
    # collectives go over gloo through host memory, so the multi-process driver can run on a 1-GPU box
    rehearsal = os.environ.get("SBMBP_REHEARSAL", "0") == "1" or args.dry_run
    if rehearsal:
        local_rank = 0
    sharded = world > 1 or args.force_sharded
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:  # only a single-rank run gets here without one (launchers set it)
        os.environ["MASTER_PORT"] = str(free_port())
    if not args.dry_run:
        torch.cuda.set_device(local_rank)
    dog.enter("rendezvous")
    # fault injection for the tests of the launcher and the watchdog (tests/test_sharded_cpu.py): "exit:<rank>:<code>" ends
    # that rank before the rendezvous, "hang:<rank>" parks it there
    inject = os.environ.get("SBMBP_BENCH_INJECT", "").split(":")
    if len(inject) >= 2 and inject[1] == str(rank):
        if inject[0] == "exit":
            sys.exit(int(inject[2]))
        if inject[0] == "hang":
            time.sleep(3600)
    if sharded:
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import sbm_bp_amd as S
    from sbm_bp_amd import synth
    S.load_library()

    N, Q, c, eps, dc, gseed = WORKLOADS[args.workload]
    if args.dry_run:
        return dry_run(args, rank, world, N, Q, c, eps, dc, gseed)
    dog.enter("setup")
    t0 = time.perf_counter()
    tc = synth.true_conf(N, Q)
    if not sharded:
        if args.workload == "C4":
            pairs, cab_mat, c_eff = synth.dc_sbm_powerlaw(N, Q, c, eps, gseed)
        else:
            pairs, cin, cout = synth.planted_partition(N, Q, c, eps, gseed)
            cab_mat = synth.cab_matrix(Q, cin, cout)
            if dc == 1:  # control workloads: Poisson degrees under the degree-corrected model (cab scaled as dc_sbm_powerlaw does)
                cab_mat = cab_mat / (2.0 * len(pairs) / N) ** 2
        g = S.Graph.from_edges(pairs, N)
        del pairs
        bm = S.blockmodel_t(g, Q, dc)
        bp = S.bp_conditional(device=local_rank)
        bp.init_messages_device(bm, tc, 1234)
        state = S.bp_blockmodel_state(cab_mat, np.array(synth.group_sizes(N, Q), dtype=np.uint32))
        bp.expand_bp_params(state)
        if args.gather == "messages":
            bp.set_gather_mode(1)
        E2_total = g.E2
        runner = bp

        def reinit():
            bp.reinit_messages_device(tc, 1234)
            bp.expand_bp_params(state)
    else:
        # the ranks join the library's own RCCL communicators (torch.distributed only carried the id and, below, the
        # barriers around the timed regions); rehearsal: the same C++ driver with gloo between the processes
        from sbm_bp_amd.distributed import Comm, ShardedBP
        comm_note = None
        if rehearsal:
            comm = Comm.callbacks_from_torch()
        else:
            try:
                comm, err = Comm.rccl_from_torch(local_rank), 0.0
            except Exception as ex:  # noqa: BLE001 - whatever RCCL could not do on this node, the line must say so
                comm, err, comm_note = None, 1.0, "RCCL communicator creation failed: %s" % ex
            flag = torch.tensor([err], device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)  # every rank takes the same branch
            if float(flag.item()) > 0:
                # last resort, so that the run still says something about this node: the same driver over gloo with the
                # buffers staged through the host (the line names the transport; it is not the design's data path)
                comm = Comm.callbacks_from_torch(group=dist.new_group(backend="gloo"))
                comm_note = comm_note or "RCCL communicator creation failed on another rank"
        # The number of row chunks per sweep trades kernel efficiency (fewer, larger launches) against how much of the halo
        # exchange hides behind the next chunk's sweep, and which side wins depends on the links of THIS node: measure it.
        # Every rank builds the same candidates in the same order, times a few sweeps between barriers, and the maximum over
        # the ranks decides for all of them (untimed set-up; SBMBP_SHARD_CHUNKS=k pins the count instead).
        chunk_trials = None
        if world > 1 and not os.environ.get("SBMBP_SHARD_CHUNKS"):
            chunk_trials, runner, graph = {}, None, None
            # "4s": four chunks on one stream (SBMBP_SHARD_STREAMS=1): they complete strictly one after the other, without
            # the second stream that fills their tails. An exported SBMBP_SHARD_STREAMS / SBMBP_SHARD_PRIO pins that switch: the
            # variants that would overwrite it are skipped. (Round 2 also tried "4p", stream priorities: dropped from the
            # automatic set. Alone on a GPU it costs 6 - 10 % of the sweep, and the one record of it beside other ranks -
            # profiles/r02_small_bench_3ranks_rehearsal.json, three PROCESSES sharing one GPU - shows 25 ms per sweep against
            # 1.5 - 2.8 ms for every other variant: a high-priority stream of one process starves the low-priority streams of
            # the others while all of them wait for each other in the per-sweep collective. SBMBP_SHARD_PRIO=1 still exists as
            # an opt-in for a node where every rank has its GPU to itself.)
            # (five candidates; a plan of the C3 graph is built in well under a second per rank at 8 ranks, a few seconds at 2)
            pinned_streams = os.environ.get("SBMBP_SHARD_STREAMS")
            variants = [1, 2, 4, 8] + ([] if pinned_streams else ["4s"])
            # two plans resident at once (the best so far and the candidate): at the capacity workloads the old best is
            # closed BEFORE the next candidate is built, and the winner is rebuilt at the end
            tight = N >= 50_000_000
            best_nc = None
            for nc in variants:
                dog.enter("chunk_trials:%s" % nc)
                if not pinned_streams:
                    os.environ["SBMBP_SHARD_STREAMS"] = "1" if str(nc).endswith("s") else "2"
                if tight and runner is not None:
                    graph = runner.graph
                    runner.close()
                    runner = None
                cand = ShardedBP.synthetic(N, Q, c, eps, gseed, dc=dc, seed=1234, comm=comm, device=local_rank,
                                           n_chunks=int(str(nc).rstrip("ps")), graph=graph)
                graph = cand.graph
                cand.sweep(2, 1.0, want_diff=False)
                best_try = None
                for _ in range(2):  # the better of two blocks of five sweeps (max over the ranks each)
                    dist.barrier()
                    torch.cuda.synchronize()
                    t_try = time.perf_counter()
                    cand.sweep(5, 1.0, want_diff=False)
                    dist.barrier()
                    torch.cuda.synchronize()
                    t_loc = torch.tensor([time.perf_counter() - t_try], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
                    dist.all_reduce(t_loc, op=dist.ReduceOp.MAX)
                    best_try = float(t_loc.item()) if best_try is None else min(best_try, float(t_loc.item()))
                chunk_trials[nc] = best_try / 5 * 1e3
                if best_nc is None or chunk_trials[nc] < 0.98 * chunk_trials[best_nc]:
                    if runner is not None:
                        runner.close()
                    runner, best_nc = cand, nc
                elif tight:
                    cand.close()
                    runner = None
                else:
                    cand.close()
            if not pinned_streams:
                os.environ["SBMBP_SHARD_STREAMS"] = "1" if str(best_nc).endswith("s") else "2"  # (what the chosen plan is created with)
            if runner is None or (tight and int(str(best_nc).rstrip("ps")) != runner.info.n_chunks):
                if runner is not None:
                    runner.close()
                runner = ShardedBP.synthetic(N, Q, c, eps, gseed, dc=dc, seed=1234, comm=comm, device=local_rank,
                                             n_chunks=int(str(best_nc).rstrip("ps")), graph=graph)
            runner.init_messages_device(1234, tc)
            runner.expand_bp_params(runner.cab, runner.na, 1.0)
        else:
            runner = ShardedBP.synthetic(N, Q, c, eps, gseed, dc=dc, seed=1234, comm=comm, device=local_rank)
        E2_total = runner.E2_global

        def reinit():
            runner.init_messages_device(1234, tc)
            runner.expand_bp_params(runner.cab, runner.na, 1.0)
    setup_s = time.perf_counter() - t0

    def barrier():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if not sharded:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    dog.enter("warmup")
    runner.set_timing(False)
    runner.sweep(args.warmup, 1.0, want_diff=False)
    runner.reset_stats()
    runner.set_timing(True)
    barrier()
    dog.enter("timed")
    t1 = time.perf_counter()
    runner.sweep(args.steps, 1.0, want_diff=False)
    barrier()
    dt = max_over_ranks(time.perf_counter() - t1)
    st = runner.stats()
    phases = runner.phase_times() if sharded else None
    kernel_ms = st.sweep_kernel_ms / max(1, st.sweep_launches)
    # what ONE launch of the timed kernel processes: this rank's rows and edges, minus the hub rows (above one segment's
    # edge capacity), which the fragment kernels update in launches of their own (C4: 8 % of the edges)
    hub_edges = int(getattr(st, "hub_edges", 0))
    bytes_per_launch = st.bytes_per_sweep - hub_edges * (24.0 * Q + 4.0) - int(st.n_hub_rows) * (8.0 * Q + 8.0)
    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0

    converge = None
    if args.converge:
        # the metric as SURVEY 8(d) defines it: the whole converge phase from the initial state, host loop included
        dog.enter("converge")
        runner.set_timing(False)
        reinit()
        barrier()
        t2 = time.perf_counter()
        niter, _ = runner.converge(CONV_CRIT, 1000, 1.0)
        barrier()
        dtc = max_over_ranks(time.perf_counter() - t2)
        sweeps = niter + 1 if niter >= 0 else 1000
        # overlap of the converged marginals with the planted labels (untimed): the device initial state and the sweeps are
        # partition invariant, so `sweeps` and `overlap` of an N-GPU line must equal the 1-GPU line's
        converge = {"crit": CONV_CRIT, "sweeps": sweeps, "converged": niter >= 0, "wall_ms": dtc * 1e3,
                    "edge_msg_per_s": sweeps * E2_total / dtc, "ms_per_sweep": dtc * 1e3 / sweeps,
                    "overlap": runner.compute_overlap(),
                    # the timed converge starts from the DEVICE initial state (random marginals, every message = its sender's
                    # marginal), not from the reference's flag-0 state of independent random messages per edge (bin/bp
                    # reproduces that one bit for bit; it needs a few sweeps more: README)
                    "init": "device: random marginals, message = sender's marginal (not the reference's -i 0 state)",
                    # adaptive relaxation: (field level, generic level) the run ended on; [0, -1] = plain synchronous sweeps
                    "relaxation": list(runner.relaxation()[:2])}

    if rank == 0:
        kname = ("k_wsweep<%d>" % ((Q + 15) // 16)) if Q > 16 else ("k_sweep_psi<%d>" if st.psi_form_sweeps else "k_sweep<%d>") % Q
        traffic, traffic_note = pmc_traffic(args.workload, kname, world, E2_total, N, Q)
        # the marginal-gather kernel's OWN minimum bytes (Q-1-component records, neighbour index instead of a reverse
        # index): stated next to SURVEY's work-unit figure, which `achieved` / `frac` use
        own_bytes = E2_total / world * (2 * 8.0 * (Q - 1) + 8.0 * Q + 4.0) + N / world * (8.0 * Q + 4.0) if st.psi_form_sweeps else None
        out = {
            "metric": "BP edge-message updates/sec", "value": args.steps * E2_total / dt, "unit": "edge-msg/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: planted SBM N=%d Q=%d c=%g eps=%g deg_corr=%d, synchronous BP sweep (-m infer inner loop)" % (
                args.workload, N, Q, c, eps, dc), "N": N, "Q": Q, "E2": int(E2_total), "hub_rows": int(st.n_hub_rows), "hub_edges": hub_edges, "parallelism": "vertex-range shards x%d%s" % (world, " (sharded driver)" if sharded else ""),
                "setup_s": round(setup_s, 2)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": kname, "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": bytes_per_launch,
                         "frac_uses": "SURVEY 8(d) bytes per sweep: E2(24Q+4) + N(8Q+8)",
                         "kernel_min_bytes_per_launch": own_bytes},
            "build": {"source_sha": source_sha()},
        }
        if own_bytes and kernel_ms > 0:
            out["roofline"]["frac_of_kernel_min_bytes"] = own_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        if traffic_note:
            out["roofline"]["traffic_source" if traffic is not None else "traffic_stale"] = traffic_note
        if sharded:  # what rank 0 ships per sweep: the numbers needed to read a multi-GPU result
            out["n_ranks_seen"] = runner.comm.world  # ranks of the communicator the sweeps ran on
            info = runner.info
            per_peer = runner.peer_rows()[0].astype(float) * info.halo_components * 8 / 1e6
            if comm_note:
                out["config"]["comm_fallback"] = comm_note
            if chunk_trials:
                out["config"]["chunk_trials_ms_per_sweep"] = {str(k): round(v, 4) for k, v in chunk_trials.items()}
                out["config"]["chunk_choice"] = str(best_nc)
            out["config"]["exchange"] = {"transport": runner.comm.transport, "chunks": int(info.n_chunks),
                                         "payload_components": int(info.halo_components), "halo_rows": int(info.n_halo),
                                         "sent_MB_per_sweep": round(float(per_peer.sum()), 2),
                                         "busiest_peer_MB_per_sweep": round(float(per_peer.max()) if len(per_peer) else 0.0, 2)}
            if phases:  # rank 0's stream: chunk kernels / fold + all-gather + finalize / wait for exchanges still in flight
                out["config"]["exchange"]["rank0_ms_per_sweep"] = {k: round(v, 4) for k, v in phases.items() if k != "sweeps"}
        if rehearsal:
            out["rehearsal"] = "ranks share cuda:0, gloo collectives staged through the host: not a measurement"
        if converge is not None:
            out["converge"] = converge
            out["sweeps_to_converge"] = converge["sweeps"] if converge["converged"] else -1
        if world == 1 and not args.no_cpu_baseline and args.cpu_baseline != "none":
            dog.enter("cpu_baseline")
            if args.cpu_baseline == "full" and args.workload != "C4":
                out["cpu_baseline"] = cpu_baseline(Q, c, eps, sample_n=N, sweeps=3, gseed=gseed, full=args.workload)
            else:
                out["cpu_baseline"] = cpu_baseline(Q, c, eps)  # plain planted partition of the same Q, c (also for C4)
        dog.enter("report")
        print(json.dumps(out), flush=True)
    dog.enter("report")
    if sharded:
        runner.close()
        dist.barrier()
        dist.destroy_process_group()
    dog.enter("done")


def dry_run(args, rank, world, N, Q, c, eps, dc, gseed):
    """no GPU: every rank builds its shard plan (csrc/dist.hip, host code) of a small graph of the workload's family and the
    ranks cross-check what they will send each other (rank r's send count to p must be rank p's receive count from r)"""
    import ctypes as C
    import numpy as np
    import torch
    import torch.distributed as dist
    import sbm_bp_amd as S
    from sbm_bp_amd import synth
    from sbm_bp_amd.capi import DistInfo, c_u64p, check
    n = min(N, 200_000)
    pairs, _, _ = synth.planted_partition(n, Q, c, eps, gseed)
    g = S.Graph.from_edges(pairs, n)
    info = DistInfo()
    sc, rc, mc = (np.zeros(world, dtype=np.uint64) for _ in range(3))
    check(S.load_library().sbmbp_plan_summary(g._h, world, rank, 2, C.byref(info), sc.ctypes.data_as(c_u64p), rc.ctypes.data_as(c_u64p),
                                              mc.ctypes.data_as(c_u64p)))
    ok = True
    if world > 1:
        mine = torch.tensor(np.concatenate([sc, rc, mc]).astype(np.int64))
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        for p in range(world):  # what p expects from me (marginal rows and cut-edge records)
            ok = ok and int(allv[p][world + rank]) == int(sc[p]) and int(allv[p][2 * world + rank]) == int(mc[p])
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "n_ranks_seen": dist.get_world_size() if world > 1 else 1,
                          "plans_consistent": bool(ok), "sample_N": n, "halo_rows_rank0": int(info.n_halo),
                          "cut_edge_records_rank0": int(info.n_halo_msgs)}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        sys.exit(1)


if __name__ == "__main__":
    main()

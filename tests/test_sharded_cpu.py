"""CPU tier for the N>1 path: the shard plan csrc/dist.hip builds (sbmbp_plan_*: host code, no GPU) against its numpy
restatement (tests/plan_model.py) array for array, its invariants (what p sends is what q expects, slice for slice; cut
edges line up record for record), and the multi-rank protocol run on that C++ plan with a numpy stand-in for the shard
kernels (tests/shard_protocol_model.py): ranks in lock-step in one process, and two real processes over gloo."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, args_of, best_perm_diff, golden


def _problem(orc, name="c1_matched_tight_seed0"):
    gd = golden(name)
    a, r = args_of(gd), gd["result"]
    g = orc.Graph.from_edgelist(a["path"], a["N"])
    bp = orc.OracleBP(g, a["Q"], a["dc"])
    bp.init_messages(0, None, a["true_conf"], orc.Rng(a["seed"]))
    cab, na = orc.param_from_direct(a["N"], a["Q"], a["pa"], a["cab_upper"])
    psi0, msg0 = bp.get_state()
    return a, r, g, cab, na, psi0, msg0


def _graph(g):
    import sbm_bp_amd as S
    return S.Graph.from_csr(g.row_ptr, g.nbr, g.rev)


def _model(g, a, cab, na, psi0, msg0, world, n_chunks=4):
    from shard_numpy_backend import NumpyShardBackend
    from shard_protocol_model import CppPlan, LocalComm, ProtocolModel
    gg = _graph(g)
    plans = [CppPlan(gg, world, r, n_chunks if world > 1 else 1) for r in range(world)]
    sb = ProtocolModel(plans, a["Q"], a["dc"], LocalComm(world), backend_factory=lambda p: NumpyShardBackend(p, a["Q"], a["dc"]))
    for sh in sb.shards:
        sh.init_from_global(psi0, msg0, a["true_conf"])
    sb.expand_bp_params(cab, na, a["beta"])
    return sb


@pytest.mark.parametrize("name", ["c1_matched_tight_seed0", "q4_tight_seed0", "hub_dc1_tight_seed0"])
def test_cpp_plan_equals_the_numpy_restatement(orc, name):
    from plan_model import ShardPlan, partition_rows
    from shard_protocol_model import CppPlan
    a, r, g, *_ = _problem(orc, name)
    gg = _graph(g)
    for world in (1, 2, 3, 8):
        bounds = partition_rows(g.row_ptr, world)
        for rank in range(world):
            ref = ShardPlan(g.row_ptr, g.nbr, bounds, rank, n_chunks=3)
            p = CppPlan(gg, world, rank, 3)
            assert (p.row0, p.n_own, p.n_halo, p.n_edges, p.edge0) == (ref.row0, ref.n_own, ref.n_halo, ref.n_edges, ref.edge0)
            for f in ("nbr_local", "halo_global", "chunk_row", "send_counts", "recv_counts", "send_counts_cp", "recv_counts_cp",
                      "send_idx_chunked", "snd_ptr", "snd_slot", "send_off_c", "stage_off_c", "send_off_cp"):
                assert (np.asarray(getattr(p, f)).astype(np.int64) == np.asarray(getattr(ref, f)).astype(np.int64)).all(), (world, rank, f)


def test_plan_invariants_and_cut_edges(orc):
    from shard_protocol_model import CppPlan
    a, r, g, *_ = _problem(orc)
    gg = _graph(g)
    for world in (1, 2, 3, 8):
        plans = [CppPlan(gg, world, k, 3) for k in range(world)]
        assert sum(p.n_own for p in plans) == g.N and sum(p.n_edges for p in plans) == g.E2
        assert plans[0].row0 == 0 and all(plans[k].row0 + plans[k].n_own == plans[k + 1].row0 for k in range(world - 1))
        w = [p.n_edges + 2 * p.n_own for p in plans]
        assert max(w) <= 1.2 * (sum(w) / world) + 64  # balanced on sum(deg + 2)
        for p in plans:  # chunked exchange: per (chunk, peer) what p sends is what the peer expects, slice for slice
            assert p.chunk_row[0] == 0 and p.chunk_row[-1] == p.n_own and (np.diff(p.chunk_row.astype(np.int64)) >= 0).all()
            table = np.concatenate([np.arange(p.row0, p.row0 + p.n_own), p.halo_global]).astype(np.int64)
            assert (table[p.nbr_local] == g.nbr[p.edge0:p.edge0 + p.n_edges]).all()  # every edge reads the right vertex
            for q in plans:
                for c in range(3):
                    s0, n = int(p.send_off_cp[c, q.rank]), int(p.send_counts_cp[c, q.rank])
                    sent = p.row0 + p.send_idx_chunked[s0:s0 + n].astype(np.int64)
                    assert ((p.row0 + int(p.chunk_row[c]) <= sent) & (sent < p.row0 + int(p.chunk_row[c + 1]))).all()
                    h0 = int(q.stage_off_c[c] + q.recv_counts_cp[c, :p.rank].sum())
                    m = int(q.recv_counts_cp[c, p.rank])
                    assert m == n and (sent == q.halo_global[h0:h0 + m]).all()
            assert p.snd_ptr[-1] == len(p.send_idx_chunked)
            for i in (0, p.n_own // 2, p.n_own - 1):  # row i's marginal goes to exactly the slots whose send index is i
                slots = p.snd_slot[p.snd_ptr[i]:p.snd_ptr[i + 1]]
                assert (p.send_idx_chunked[slots] == i).all() and len(slots) == int((p.send_idx_chunked == i).sum())
        # message-gather form: rev_local of an own-own edge is the local reverse edge; the records p sends q arrive, in order,
        # exactly where q's cut edges to p point (the x-th record from p is the reverse message of q's x-th cut edge to p)
        for p in plans:
            src = np.repeat(np.arange(p.n_own), p.deg) + p.row0
            dst = g.nbr[p.edge0:p.edge0 + p.n_edges].astype(np.int64)
            own = (dst >= p.row0) & (dst < p.row0 + p.n_own)
            assert (p.rev_local[own].astype(np.int64) == g.rev[p.edge0:p.edge0 + p.n_edges][own].astype(np.int64) - p.edge0).all()
            assert p.n_halo_msgs == int((~own).sum()) and sorted(p.rev_local[~own].tolist()) == list(range(p.n_edges, p.n_edges + p.n_halo_msgs))
            off = 0
            for q in plans:
                n = int(p.msg_counts[q.rank])
                assert n == int(q.msg_counts[p.rank])
                sent = p.msg_send_edge[off:off + n].astype(np.int64)  # p's local edges (i -> l), l owned by q
                off += n
                qsrc = np.repeat(np.arange(q.n_own), q.deg) + q.row0
                qdst = g.nbr[q.edge0:q.edge0 + q.n_edges].astype(np.int64)
                r0 = q.n_edges + int(q.msg_counts[:p.rank].sum())
                for x in (0, n // 2, n - 1) if n else ():
                    k = int(np.flatnonzero(q.rev_local == r0 + x)[0])  # q's edge that gathers record x from p
                    assert (qsrc[k], qdst[k]) == (dst[sent[x]], src[sent[x]])


@pytest.mark.parametrize("world", [2, 3, 5])
def test_protocol_iterates_equal_one_rank(orc, world):
    a, r, g, cab, na, psi0, msg0 = _problem(orc)
    ref = _model(g, a, cab, na, psi0, msg0, 1)
    sb = _model(g, a, cab, na, psi0, msg0, world)
    for _ in range(3):
        d1, dk = ref.sweep(2), sb.sweep(2)
        assert abs(d1 - dk) < 1e-13
        psi_ref, msg_ref = ref.shards[0].get_state()
        psi_k = np.concatenate([sh.get_state()[0] for sh in sb.shards])
        msg_k = np.concatenate([sh.get_state()[1] for sh in sb.shards])
        assert np.abs(psi_k - psi_ref).max() < 1e-13 and np.abs(msg_k - msg_ref).max() < 1e-13


@pytest.mark.parametrize("name", ["c1_matched_tight_seed0", "c1_dc1_tight_seed0", "q4_tight_seed0"])
def test_protocol_converges_to_reference_fixed_point(orc, name):
    a, r, g, cab, na, psi0, msg0 = _problem(orc, name)
    sb = _model(g, a, cab, na, psi0, msg0, 3)
    niter, exact = sb.converge(1e-12, 3000, 1.0, check_every=5)
    assert niter >= 0 and exact < 1e-12
    psi = np.concatenate([sh.get_state()[0] for sh in sb.shards])
    d, _ = best_perm_diff(psi, np.array(r["psi"]).reshape(psi.shape))
    assert d < 1e-9
    assert abs(sb.compute_overlap() - r["overlap"]) < 1e-9
    one = _model(g, a, cab, na, psi0, msg0, 1)
    assert one.converge(1e-12, 3000, 1.0, check_every=1)[0] == niter  # batching and sharding do not change niter


def test_two_processes_over_gloo(orc, tmp_path):
    """world_size 2, backend gloo: each process builds ITS rank's plan in C++ and the two exchange what the plans say"""
    script = os.path.join(ROOT, "tests", "sharded_gloo_worker.py")
    out = tmp_path / "result.npz"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    from bench import free_port
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), script, str(out)]
    subprocess.run(cmd, check=True, env=env, timeout=300, capture_output=True)
    res = np.load(out)
    a, r, g, cab, na, psi0, msg0 = _problem(orc)
    ref = _model(g, a, cab, na, psi0, msg0, 1)
    niter, exact = ref.converge(1e-12, 3000, 1.0, check_every=4)
    assert int(res["niter"]) == niter
    assert np.abs(res["psi"] - ref.shards[0].get_state()[0]).max() < 1e-12
    assert abs(float(res["overlap"]) - ref.compute_overlap()) < 1e-12


def test_bench_self_launches_its_ranks_dry_run():
    """`python bench.py --gpus 2` from a bare shell starts its own two ranks (no torchrun): rendezvous, per-rank C++ plans,
    cross-check of what the ranks will send each other — the part of the N > 1 bench that needs no GPU"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--workload", "small"],
                        env=env, timeout=300, capture_output=True, text=True)
    assert pr.returncode == 0, pr.stderr[-2000:]
    line = [l for l in pr.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["dry_run"] and out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["plans_consistent"]


def _bench_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(extra)
    return env


def test_bench_launcher_ends_the_other_ranks_when_one_dies():
    """a rank that dies outside a collective (out of memory, a failed RCCL init) leaves its peers waiting for it in the
    rendezvous or a collective: the launcher polls all its children, takes the others down and exits with the failed rank's
    code instead of hanging (round 2 waited for rank 0 first, forever)"""
    import time
    t0 = time.monotonic()
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--dry-run", "--workload", "small"],
                        env=_bench_env(SBMBP_BENCH_INJECT="exit:2:7"), timeout=240, capture_output=True, text=True)
    assert pr.returncode == 7, (pr.returncode, pr.stderr[-1500:])
    assert time.monotonic() - t0 < 200  # well inside the rendezvous timeout of the surviving ranks


def test_bench_watchdog_names_the_phase_of_a_hung_rank():
    """a rank parked before the rendezvous: every rank's in-process watchdog (a thread that never touches the GPU) ends its
    process with ONE diagnostic JSON line naming the phase and code 3; the launcher returns that code. Nothing re-executes."""
    import json
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--workload", "small"],
                        env=_bench_env(SBMBP_BENCH_INJECT="hang:1", SBMBP_PHASE_DEADLINE_S="0.02"), timeout=240,
                        capture_output=True, text=True)
    assert pr.returncode == 3, (pr.returncode, pr.stderr[-1500:])
    lines = [json.loads(l) for l in pr.stdout.splitlines() if l.startswith("{")]
    assert lines and all(l.get("error") == "watchdog" for l in lines)
    assert any(l["phase"] == "rendezvous" for l in lines)


def test_bench_launcher_passes_a_sigterm_on():
    """SIGTERM to the launcher must not orphan the ranks"""
    import signal
    import time
    pr = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--workload", "small"],
                          env=_bench_env(SBMBP_BENCH_INJECT="hang:0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    time.sleep(8.0)  # the ranks are up (importing torch) or parked
    pr.send_signal(signal.SIGTERM)
    rc = pr.wait(timeout=60)
    assert rc == 128 + signal.SIGTERM
    # no rank process of that launcher is left behind
    out = subprocess.run(["ps", "-eo", "ppid,pid,args"], capture_output=True, text=True).stdout
    assert not [l for l in out.splitlines() if l.split()[0] == str(pr.pid)]


def test_shard_plan_under_asan_with_rank_threads(tmp_path):
    """the plan code of csrc/dist.hip (host code; the ranks of `bin/bp --gpus N` and of LocalShards build their plans side by
    side as threads over ONE shared graph) compiled with the host side instrumented by AddressSanitizer - device code
    untouched: GPU AddressSanitizer is not available on this pool - and driven by concurrent rank threads"""
    import shutil
    import subprocess
    import sys
    from conftest import ROOT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    clang = "/opt/rocm/lib/llvm/bin/clang"
    if not (os.path.exists(hipcc) and os.path.exists(clang)):
        pytest.skip("no hipcc")
    rt = subprocess.run([clang, "--print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(rt) or not os.path.exists(rt):
        pytest.skip("no shared AddressSanitizer runtime in this toolchain")
    lib = tmp_path / "libsbmbp_asan.so"
    csrc = os.path.join(ROOT, "sbm-bp_amd", "csrc")
    subprocess.run([hipcc, "-std=c++14", "-O1", "-g", "-fsanitize=address", "-fno-gpu-sanitize", "-shared-libsan", "--offload-arch=gfx950",
                    "-fPIC", "-shared", "-pthread", "-DSBMBP_ONLY_Q=4", "-o", str(lib)] +
                   [os.path.join(csrc, f) for f in ("engine.hip", "dist.hip", "host_graph.cpp")] + ["-lrccl"], check=True, timeout=900)
    prog = r"""
import sys, threading
sys.path[:0] = [%r, %r, %r]
import numpy as np
import sbm_bp_amd as S
from shard_protocol_model import CppPlan
rng = np.random.default_rng(3)
N = 3000
pairs = rng.integers(0, N, size=(12000, 2)).astype(np.uint32)
pairs = np.concatenate([pairs, np.stack([np.zeros(700, dtype=np.uint32), np.arange(1, 701, dtype=np.uint32)], 1)])  # one long row
for it in range(6):
    g = S.Graph.from_edges(pairs, N)
    for world, nc in ((3, 4), (5, 1), (2, 3), (8, 2)):
        errs = []
        def work(r):
            try:
                p = CppPlan(g, world, r, nc)
                assert (np.diff(p.snd_ptr.astype(np.int64)) >= 0).all()
            except Exception as ex:
                errs.append((r, repr(ex)))
        ts = [threading.Thread(target=work, args=(r,)) for r in range(world)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert not errs, errs
print("plans ok")
""" % (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"))
    pr = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=900,
                        env=dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", SBMBP_LIB=str(lib)))
    assert pr.returncode == 0 and "plans ok" in pr.stdout, (pr.stdout[-500:], pr.stderr[-3000:])
    assert "AddressSanitizer" not in pr.stderr

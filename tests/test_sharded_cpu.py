"""CPU tier for the N>1 path: the sharding plan, the halo exchange and the convergence driver of
sbm-bp_amd/distributed.py, with a numpy stand-in for the shard kernel (tests/shard_numpy_backend.py).
Covers in-process lock-step shards (LocalComm) and two real processes over gloo (TorchDistComm)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, args_of, best_perm_diff, golden, gpath


def _problem(orc, name="c1_matched_tight_seed0"):
    gd = golden(name)
    a, r = args_of(gd), gd["result"]
    g = orc.Graph.from_edgelist(a["path"], a["N"])
    bp = orc.OracleBP(g, a["Q"], a["dc"])
    bp.init_messages(0, None, a["true_conf"], orc.Rng(a["seed"]))
    cab, na = orc.param_from_direct(a["N"], a["Q"], a["pa"], a["cab_upper"])
    psi0, msg0 = bp.get_state()
    return a, r, g, cab, na, psi0, msg0


def _sharded(g, a, cab, na, psi0, msg0, world):
    from sbm_bp_amd.distributed import LocalComm, ShardedBP
    from shard_numpy_backend import NumpyShardBackend
    comm = LocalComm(world)
    sb = ShardedBP.from_csr(g.row_ptr, g.nbr, a["Q"], a["dc"], comm, backend_factory=lambda p: NumpyShardBackend(p, a["Q"], a["dc"]))
    for sh in sb.shards:
        sh.init_from_global(psi0, msg0, a["true_conf"])
    sb.expand_bp_params(cab, na, a["beta"])
    return sb


def test_partition_and_plan_invariants(orc):
    from sbm_bp_amd.plan import ShardPlan, partition_rows
    a, r, g, *_ = _problem(orc)
    for world in (1, 2, 3, 8):
        bounds = partition_rows(g.row_ptr, world)
        assert bounds[0] == 0 and bounds[-1] == g.N and (np.diff(bounds) > 0).all()
        plans = [ShardPlan(g.row_ptr, g.nbr, bounds, k, n_chunks=3) for k in range(world)]
        assert sum(p.n_own for p in plans) == g.N and sum(p.n_edges for p in plans) == g.E2
        for p in plans:  # chunked exchange: per (chunk, peer) what p sends is what the peer expects, slice for slice
            assert p.chunk_row[0] == 0 and p.chunk_row[-1] == p.n_own and (np.diff(p.chunk_row.astype(np.int64)) >= 0).all()
            assert sorted(p.send_idx_chunked.tolist()) == sorted(p.send_idx.tolist())
            for q in plans:
                for c in range(3):
                    s0, n = int(p.send_off_cp[c, q.rank]), int(p.send_counts_cp[c, q.rank])
                    sent = p.row0 + p.send_idx_chunked[s0:s0 + n]
                    assert ((p.row0 + p.chunk_row[c] <= sent) & (sent < p.row0 + p.chunk_row[c + 1])).all()
                    h0, m = int(q.recv_off_cp[c, p.rank]), int(q.recv_counts_cp[c, p.rank])
                    assert m == n and (sent == q.halo_global[h0:h0 + m]).all()
        w = [p.n_edges + 2 * p.n_own for p in plans]
        assert max(w) <= 1.2 * (sum(w) / world) + 64
        for p in plans:
            assert (p.nbr_local < p.n_own + p.n_halo).all()
            # the local table entry of every edge is the right global vertex
            table = np.concatenate([np.arange(p.row0, p.row0 + p.n_own), p.halo_global])
            assert (table[p.nbr_local] == g.nbr[p.edge0:p.edge0 + p.n_edges]).all()
            for q in plans:  # what p sends to q is exactly the part of q's halo that p owns (the halo is kept in receive
                s0 = int(p.send_counts[:q.rank].sum())  # order, (chunk, peer, id): checked slice for slice above)
                sent = p.row0 + p.send_idx[s0:s0 + int(p.send_counts[q.rank])]
                owned = q.halo_global[(q.halo_global >= p.row0) & (q.halo_global < p.row0 + p.n_own)]
                assert sorted(sent.tolist()) == sorted(owned.tolist())
            # send slots: row i's marginal goes to exactly the slots whose send index is i
            assert p.snd_ptr[-1] == len(p.send_idx_chunked)
            for i in (0, p.n_own // 2, p.n_own - 1):
                slots = p.snd_slot[p.snd_ptr[i]:p.snd_ptr[i + 1]]
                assert (p.send_idx_chunked[slots] == i).all() and len(slots) == int((p.send_idx_chunked == i).sum())


@pytest.mark.parametrize("world", [2, 3, 5])
def test_sharded_iterates_equal_unsharded(orc, world):
    a, r, g, cab, na, psi0, msg0 = _problem(orc)
    ref = _sharded(g, a, cab, na, psi0, msg0, 1)
    sb = _sharded(g, a, cab, na, psi0, msg0, world)
    for _ in range(3):
        d1, dk = ref.sweep(2), sb.sweep(2)
        assert abs(d1 - dk) < 1e-13
        psi_ref, msg_ref = ref.local_state()[0]
        psi_k = np.concatenate([s[0] for s in sb.local_state()])
        msg_k = np.concatenate([s[1] for s in sb.local_state()])
        assert np.abs(psi_k - psi_ref).max() < 1e-13 and np.abs(msg_k - msg_ref).max() < 1e-13


@pytest.mark.parametrize("name", ["c1_matched_tight_seed0", "c1_dc1_tight_seed0", "q4_tight_seed0"])
def test_sharded_converges_to_reference_fixed_point(orc, name):
    a, r, g, cab, na, psi0, msg0 = _problem(orc, name)
    sb = _sharded(g, a, cab, na, psi0, msg0, 3)
    niter, exact = sb.converge(1e-12, 3000, 1.0, check_every=5)
    assert niter >= 0 and exact < 1e-12
    psi = np.concatenate([s[0] for s in sb.local_state()])
    d, _ = best_perm_diff(psi, np.array(r["psi"]).reshape(psi.shape))
    assert d < 1e-9
    assert abs(sb.compute_overlap() - r["overlap"]) < 1e-9
    one = _sharded(g, a, cab, na, psi0, msg0, 1)
    assert one.converge(1e-12, 3000, 1.0, check_every=1)[0] == niter  # batching and sharding do not change niter


def test_two_processes_over_gloo(orc, tmp_path):
    """world_size 2, backend gloo: the TorchDistComm path (all_to_all_single + all_reduce)"""
    script = os.path.join(ROOT, "tests", "sharded_gloo_worker.py")
    out = tmp_path / "result.npz"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", script, str(out)]
    subprocess.run(cmd, check=True, env=env, timeout=300, capture_output=True)
    res = np.load(out)
    a, r, g, cab, na, psi0, msg0 = _problem(orc)
    ref = _sharded(g, a, cab, na, psi0, msg0, 1)
    niter, exact = ref.converge(1e-12, 3000, 1.0, check_every=4)
    assert int(res["niter"]) == niter
    assert np.abs(res["psi"] - ref.local_state()[0][0]).max() < 1e-12
    assert abs(float(res["overlap"]) - ref.compute_overlap()) < 1e-12


@pytest.mark.parametrize("world,k", [(2, 2), (3, 2), (4, 4)])
def test_block_cyclic_layout_is_the_same_problem(orc, world, k):
    """dealing k*world row blocks round robin (plan.block_cyclic_layout) renames the vertices and nothing else: same
    graph, every shard one contiguous range, and the sharded run lands on the same marginals in the caller's order"""
    from sbm_bp_amd.distributed import LocalComm, ShardedBP
    from sbm_bp_amd.plan import block_cyclic_layout, busiest_link_rows, edge_order, permute_csr
    from shard_numpy_backend import NumpyShardBackend
    a, r, g, cab, na, psi0, msg0 = _problem(orc)
    order, bounds, bb = block_cyclic_layout(g.row_ptr, world, k)
    assert sorted(order.tolist()) == list(range(g.N)) and bounds[0] == 0 and bounds[-1] == g.N
    rp2, nb2, inv = permute_csr(g.row_ptr, g.nbr, order, bb)
    eo = edge_order(g.row_ptr, order)
    assert (np.diff(rp2.astype(np.int64)) == np.diff(g.row_ptr.astype(np.int64))[order]).all()
    assert (order[nb2] == g.nbr[eo]).all()  # edge for edge the same neighbours under the renaming
    assert busiest_link_rows(g.row_ptr, g.nbr, world, 1, 0) > 0
    plain = _sharded(g, a, cab, na, psi0, msg0, world)
    plain.converge(1e-12, 3000, 1.0)
    sb = ShardedBP.from_csr(g.row_ptr, g.nbr, a["Q"], a["dc"], LocalComm(world), interleave=k,
                            backend_factory=lambda p: NumpyShardBackend(p, a["Q"], a["dc"]))
    assert sb.interleave == k and (sb.order == order).all()
    for sh in sb.shards:
        sh.init_from_global(psi0[order], msg0[eo], np.asarray(a["true_conf"])[order])
    sb.expand_bp_params(cab, na, a["beta"])
    niter, _ = sb.converge(1e-12, 3000, 1.0)
    assert niter >= 0
    psi = sb.to_caller_order(np.concatenate([s[0] for s in sb.local_state()]))
    assert np.abs(psi - np.concatenate([s[0] for s in plain.local_state()])).max() < 1e-10
    assert abs(sb.compute_overlap() - plain.compute_overlap()) < 1e-10
    # the link-load estimate used by interleave="auto" equals what the plan really sends
    for rk, p in enumerate(sb.plans):
        assert busiest_link_rows(g.row_ptr, g.nbr, world, k, rk) == int(p.send_counts.max())

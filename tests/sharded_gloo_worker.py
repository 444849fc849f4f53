"""worker of tests/test_sharded_cpu.py::test_two_processes_over_gloo (launched by torch.distributed.run)"""
import os
import sys

import numpy as np
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

import oracle as orc  # noqa: E402
import sbm_bp_amd as S  # noqa: E402
from conftest import args_of, golden  # noqa: E402
from shard_numpy_backend import NumpyShardBackend  # noqa: E402
from shard_protocol_model import CppPlan, ProtocolModel, TorchDistComm  # noqa: E402


def main():
    dist.init_process_group("gloo")
    a = args_of(golden("c1_matched_tight_seed0"))
    g = orc.Graph.from_edgelist(a["path"], a["N"])
    bp = orc.OracleBP(g, a["Q"], a["dc"])
    bp.init_messages(0, None, a["true_conf"], orc.Rng(a["seed"]))
    cab, na = orc.param_from_direct(a["N"], a["Q"], a["pa"], a["cab_upper"])
    psi0, msg0 = bp.get_state()
    comm = TorchDistComm()
    plan = CppPlan(S.Graph.from_csr(g.row_ptr, g.nbr, g.rev), comm.world, comm.rank, 4)  # this rank's plan, built in C++
    sb = ProtocolModel([plan], a["Q"], a["dc"], comm, backend_factory=lambda p: NumpyShardBackend(p, a["Q"], a["dc"]))
    sb.shards[0].init_from_global(psi0, msg0, a["true_conf"])
    sb.expand_bp_params(cab, na, a["beta"])
    niter, exact = sb.converge(1e-12, 3000, 1.0, check_every=4)
    ov = sb.compute_overlap()
    psi_local = sb.shards[0].get_state()[0]
    gathered = [None] * comm.world
    dist.all_gather_object(gathered, psi_local)
    if comm.rank == 0:
        np.savez(sys.argv[1], niter=niter, exact=exact, overlap=ov, psi=np.concatenate(gathered))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

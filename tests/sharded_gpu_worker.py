"""worker of tests/test_gpu_sharded.py::test_three_processes_share_one_gpu (launched by torch.distributed.run):
one rank of the C++ driver per process, all on cuda:0, the callback transport over gloo (buffers staged through the host);
with a third argument "rccl" (test_two_processes_two_gpus_over_rccl): one GPU per rank and the library's RCCL communicators."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

import oracle as orc  # noqa: E402
import sbm_bp_amd as S  # noqa: E402
from conftest import args_of, golden  # noqa: E402
from sbm_bp_amd.distributed import Comm, ShardedBP  # noqa: E402


def main():
    out, name = sys.argv[1], sys.argv[2]
    rccl = len(sys.argv) > 3 and sys.argv[3] == "rccl"
    device = int(os.environ.get("LOCAL_RANK", "0")) if rccl else 0
    torch.cuda.set_device(device)
    dist.init_process_group("nccl" if rccl else "gloo")
    a = args_of(golden(name))
    g = orc.Graph.from_edgelist(a["path"], a["N"])
    bp = orc.OracleBP(g, a["Q"], a["dc"])
    bp.init_messages(0, None, a["true_conf"], orc.Rng(a["seed"]))
    cab, na = orc.param_from_direct(a["N"], a["Q"], a["pa"], a["cab_upper"])
    psi0, msg0 = bp.get_state()
    comm = Comm.rccl_from_torch(device) if rccl else Comm.callbacks_from_torch()
    sb = ShardedBP(S.Graph.from_csr(g.row_ptr, g.nbr), a["Q"], a["dc"], comm, device=device)
    assert sb.info.n_chunks == 4 and comm.transport == ("rccl" if rccl else "callbacks")  # the chunked exchange of the multi-rank path
    e0 = int(g.row_ptr[sb.row0])
    sb.init_messages_device(7, a["true_conf"])
    sb.set_state(psi0[sb.row0:sb.row0 + sb.n_own], msg0[e0:e0 + sb.n_edges])
    sb.expand_bp_params(cab, na, a["beta"])
    d3 = sb.sweep(3)
    niter, exact = sb.converge(1e-12, 3000, 1.0, check_every=6)
    ov = sb.compute_overlap()
    fe = sb.compute_free_energy()
    ent = sb.compute_entropy()
    psi_local = sb.get_state()[0]
    gathered = [None] * comm.world
    dist.all_gather_object(gathered, psi_local)  # (pickled through the process group: works on either backend)
    if comm.rank == 0:
        np.savez(out, d3=d3, niter=niter, exact=exact, overlap=ov, fe=fe, entropy=ent, psi=np.concatenate(gathered))
    sb.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

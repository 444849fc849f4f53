"""Measurement aid: builds and tears down the in-process multi-rank driver over and over (rank threads creating their shards
side by side), on a fixture graph. Looks for anything intermittent in the set-up path; prints the first error in full.
  python3 tools/stress_shard_create.py [rounds=60]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main(rounds=60):
    import numpy as np
    import oracle as orc
    from conftest import args_of
    from sbm_bp_amd.distributed import LocalShards
    a = args_of(json.load(open(os.path.join(ROOT, "tests", "golden", "q4_tight_seed0.json"))))
    g = orc.Graph.from_edgelist(a["path"], a["N"])
    cab, na = orc.param_from_direct(a["N"], a["Q"], a["pa"], a["cab_upper"])
    n = 0
    for it in range(rounds):
        for world in (3, 5, 2, 4):
            sb = LocalShards.from_csr(g.row_ptr, g.nbr, a["Q"], a["dc"], world, None if it % 2 else 3)
            sb.init_messages_device(7, a["true_conf"])
            sb.expand_bp_params(cab, na, a["beta"])
            if it % 4 == 0:
                sb.sweep(2, 1.0)
            if it % 3:
                sb.close()  # (otherwise left to the garbage collector, as a test that fails half way would)
            n += 1
    print("created and dropped %d multi-rank drivers without an error" % n)


if __name__ == "__main__":
    main(*(int(x) for x in sys.argv[1:]))

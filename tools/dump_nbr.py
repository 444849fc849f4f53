"""Measurement aid: dump the neighbour array of the C3 bench graph for tools/probe_floor."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sbm_bp_amd as S
from sbm_bp_amd import synth
pairs, cin, cout = synth.planted_partition(10_000_000, 4, 10.0, 0.1, 2)
g = S.Graph.from_edges(pairs, 10_000_000)
g.csr()[1].tofile(sys.argv[1])

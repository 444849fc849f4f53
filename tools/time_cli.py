"""Measurement aid: end-to-end wall time of bin/bp on a C2-size edge list (N=1e6, Q=2, c=3)."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sbm_bp_amd import synth
p, cin, cout = synth.planted_partition(1_000_000, 2, 3.0, 0.1, 1)
np.savetxt('/tmp/c2.edgelist', p, fmt='%d')
base = [os.path.join(ROOT, 'bin', 'bp'), '-l', '/tmp/c2.edgelist', '-n', '500000', '500000', '--epsilon_c', '0.1', '3.0', '-d', '0',
        '-t', '1000', '--metrics_json', '/tmp/m.json']
for extra in (['-m', 'infer'], ['-m', 'infer'], ['-m', 'learn', '-t', '200']):
    t0 = time.perf_counter()
    r = subprocess.run(base + extra, capture_output=True, text=True)
    dt = time.perf_counter() - t0
    m = json.load(open('/tmp/m.json'))
    print(' '.join(extra), '| wall %.2f s | engine phase %.3f s | sweeps %d |' % (dt, m['run_seconds'], m['sweeps']), r.stdout.strip().replace('\n', ' / '))

"""Measurement aid: end-to-end wall time of bin/bp on a synthetic edge list, with the host phase times.
usage: python tools/time_cli.py [N Q c]   (default 1000000 2 3 = C2)"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sbm_bp_amd import synth
N, Q, c = (int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (1_000_000, 2, 3.0)
import tempfile
tmp = tempfile.mkdtemp(prefix='sbmbp_cli_')  # scratch outside gpurun_out/ (which is copied back, 64 MiB cap)
path, mj = os.path.join(tmp, 'cli.edgelist'), os.path.join(tmp, 'm.json')
t0 = time.perf_counter()
p, cin, cout = synth.planted_partition(N, Q, c, 0.1, 1)
with open(path, 'w') as f:  # np.savetxt is far too slow at 5e7 lines
    step = 2_000_000
    for i in range(0, len(p), step):
        blk = p[i:i + step]
        f.write('\n'.join(map(' '.join, blk.astype(str))) + '\n')
print('edge list: %d lines, %.0f MB, written in %.1f s' % (len(p), os.path.getsize(path) / 1e6, time.perf_counter() - t0), flush=True)
del p
base = [os.path.join(ROOT, 'bin', 'bp'), '-l', path, '-n'] + [str(N // Q)] * Q + ['--epsilon_c', '0.1', str(c), '-d', '0',
        '-t', '1000', '--metrics_json', mj]
env = dict(os.environ, SBMBP_HOST_TIMING='1')
for extra in (['-m', 'infer'], ['-m', 'infer']):
    t0 = time.perf_counter()
    r = subprocess.run(base + extra, capture_output=True, text=True, env=env)
    dt = time.perf_counter() - t0
    m = json.load(open(mj))
    print(' '.join(extra), '| wall %.2f s | engine phase %.3f s | sweeps %d |' % (dt, m['run_seconds'], m['sweeps']), r.stdout.strip().replace('\n', ' / '))
    print('\n'.join(l for l in r.stderr.split('\n') if l.startswith('[sbmbp')), flush=True)
import shutil
shutil.rmtree(tmp, ignore_errors=True)

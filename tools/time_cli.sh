#!/bin/bash
# Measurement aid: end-to-end wall time of bin/bp on a C2-size edge list (N=1e6, Q=2, c=3)
set -e
cd "$(dirname "$0")/.."
python - <<'PY'
import sys, time
sys.path.insert(0, '.')
from sbm_bp_amd import synth
import numpy as np
p, cin, cout = synth.planted_partition(1_000_000, 2, 3.0, 0.1, 1)
t = time.time()
np.savetxt('/tmp/c2.edgelist', p, fmt='%d')
print('wrote', len(p), 'edges in %.1f s' % (time.time() - t))
PY
for args in "-m infer" "-m infer --precision 12 -e 1e-10 -t 2000" "-m learn -t 200"; do
  echo "== bin/bp $args"
  t0=$(date +%s.%N)
  ./bin/bp -l /tmp/c2.edgelist -n 500000 500000 --epsilon_c 0.1 3.0 -d 0 -t 1000 $args --metrics_json /tmp/m.json 2>&1 | grep -v "^Randomly\|^Warning"
  t1=$(date +%s.%N); echo "wall $(echo "$t1 - $t0" | bc) s"
  cat /tmp/m.json
done

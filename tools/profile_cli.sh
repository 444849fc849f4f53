#!/bin/bash
# rocprofv3 kernel summary of one `bin/bp -m infer` run on a synthetic C3-size edge list (N=1e7, Q=4, c=10).
# usage (repo root, GPU box): tools/profile_cli.sh r01
set -u
R=${1:-r02}
OUT=gpurun_out/profile_cli_$R
mkdir -p $OUT
export TMPDIR=/tmp
EL=/tmp/sbmbp_c3.edgelist
python3 - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from sbm_bp_amd import synth
p, cin, cout = synth.planted_partition(10_000_000, 4, 10.0, 0.1, 1)
with open("/tmp/sbmbp_c3.edgelist", "w") as f:
    step = 2_000_000
    for i in range(0, len(p), step):
        f.write("\n".join(map(" ".join, p[i:i + step].astype(str))) + "\n")
print("edge list written:", len(p), "lines", flush=True)
PY
SBMBP_HOST_TIMING=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- bin/bp -l $EL -n 2500000 2500000 2500000 2500000 \
  --epsilon_c 0.1 10 -d 0 -t 1000 -m infer > $OUT/stdout.txt 2> $OUT/stderr.txt
cat $OUT/stdout.txt
grep "sbmbp cli" $OUT/stderr.txt
find $OUT -name "*kernel_trace.csv" -delete
rm -f $EL

#!/bin/bash
# per-rank kernel budget of the 8-rank C3 plan for 1 / 2 / 4 chunks (one rank alone on the GPU, stand-in transport)
# usage (repo root, GPU box): tools/budget_round.sh r02
set -u
R=${1:-r02}
export TMPDIR=/tmp
for CH in 1 2 4; do
  OUT=gpurun_out/budget_${R}_c$CH
  mkdir -p $OUT
  SBMBP_SHARD_CHUNKS=$CH rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 tools/shard_budget.py run C3 8 0 10 > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
  grep -h "rank 0\|BUDGET_MARK" $OUT/run.log
  python3 tools/shard_budget.py sum $OUT 13 $OUT/budget.json "C3, rank 0 of 8, $CH chunk(s)"
  find $OUT -name "*kernel_trace.csv" -delete
done

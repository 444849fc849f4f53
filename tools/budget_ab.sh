#!/bin/bash
# per-rank budget of the 8-rank C3 plan over the tuning knobs (one rank alone on the GPU): tools/budget_ab.sh
mkdir -p gpurun_out/budget
for CH in 1 2 4 8; do
  for ST in 1 2; do
    [ $CH = 1 ] && [ $ST = 2 ] && continue
    SBMBP_SHARD_CHUNKS=$CH SBMBP_SHARD_STREAMS=$ST python3 tools/shard_budget.py C3 8 0 20 gpurun_out/budget/c3_w8_c${CH}_s${ST}.json 2>&1 | grep -v amdgpu.ids
  done
done

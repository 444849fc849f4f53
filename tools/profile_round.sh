#!/bin/bash
# Regenerates the evidence under profiles/ for one round on the GPU box:
#   tools/profile_round.sh r01        (run from the repo root; writes gpurun_out/profile_<round>/ then summarises)
# rocprofv3 is always given the program itself after `--`; PMC counters are collected in their own passes.
set -u
R=${1:-r02}
OUT=gpurun_out/profile_$R
mkdir -p $OUT
export TMPDIR=/tmp
python3 -c "from bench import source_sha; print(source_sha())" > $OUT/source_sha.txt  # the sources everything below is measured on
# counters first: the benches below then find a PMC summary collected on THESE sources and report roofline.traffic from it
for CTR in FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum; do
  echo "== pmc $CTR" >&2
  rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d $OUT/pmc_$CTR -o run -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > $OUT/pmc_$CTR.log 2>&1 || echo "pmc $CTR failed" >&2
done
python3 tools/summarise_profiles.py $R pmc
for WL in C3 C2 C4 C5; do
  echo "== bench $WL" >&2
  if [ $WL = C3 ]; then python3 bench.py --workload $WL --converge > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err || exit 1
  else python3 bench.py --workload $WL --no-cpu-baseline --converge > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err || exit 1; fi
  tail -c 600 $OUT/bench_$WL.json >&2
  echo "== rocprofv3 stats $WL" >&2
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$WL -o run -- python3 bench.py --workload $WL --no-cpu-baseline --no-converge > $OUT/stats_$WL.log 2>&1 || exit 1
done
echo "== bench C3 --gather messages" >&2
python3 bench.py --workload C3 --no-cpu-baseline --gather messages > $OUT/bench_C3_messages.json 2> $OUT/bench_C3_messages.err || exit 1
echo "== per-rank budget of the 8-rank C3 plan" >&2
for CH in 1 2 4; do
  SBMBP_SHARD_CHUNKS=$CH python3 tools/shard_budget.py C3 8 0 20 $OUT/budget_c3_w8_c$CH.json 2>/dev/null | tail -1 >&2
done
echo "== the multi-process path on this one GPU (3 ranks, callback transport over gloo): functional rehearsal" >&2
SBMBP_REHEARSAL=1 python3 bench.py --gpus 3 --workload small --no-cpu-baseline > $OUT/bench_rehearsal3.json 2> $OUT/bench_rehearsal3.err || echo "rehearsal failed" >&2
echo "== RCCL communicators with one rank" >&2
python3 bench.py --force-sharded --no-cpu-baseline > $OUT/bench_C3_rccl1.json 2> $OUT/bench_C3_rccl1.err || echo "rccl1 failed" >&2
find $OUT -name "*kernel_trace.csv" -delete
python3 tools/summarise_profiles.py $R

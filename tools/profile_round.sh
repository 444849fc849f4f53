#!/bin/bash
# Regenerates the evidence under profiles/ for one round on the GPU box:
#   tools/profile_round.sh r03        (run from the repo root; everything lands under gpurun_out/profile_<round>/ and
#   gpurun_out/pmc_<round>_<workload>/; afterwards, in the build container: python3 tools/summarise_profiles.py r03)
# rocprofv3 is always given the program itself after `--`; PMC counters are collected in their own passes.
set -u
R=${1:-r03}
OUT=gpurun_out/profile_$R
mkdir -p $OUT
export TMPDIR=/tmp
python3 -c "from bench import source_sha; print(source_sha())" > $OUT/source_sha.txt  # the sources everything below is measured on
# counters first: the benches below then find a PMC summary collected on THESE sources and report roofline.traffic from it
# (a gpurun call is limited to 20 minutes: PROFILE_PART=pmc runs the counter passes only, PROFILE_PART=rest everything else)
PART=${PROFILE_PART:-all}
if [ $PART != rest ]; then
  for WL in C3 C4 C2 C5 Q32 Q64; do
    tools/pmc_workload.sh $R $WL > $OUT/pmc_$WL.log 2>&1 || echo "pmc $WL failed" >&2
  done
fi
[ $PART = pmc ] && { echo "profile round $R: counters done" >&2; exit 0; }
for WL in C3 C2 C4 C5; do
  echo "== bench $WL" >&2
  if [ $WL = C3 ]; then python3 bench.py --workload $WL --converge > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err || exit 1
  else python3 bench.py --workload $WL --no-cpu-baseline --converge > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err || exit 1; fi
  tail -c 400 $OUT/bench_$WL.json >&2
  echo "== rocprofv3 stats $WL" >&2
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$WL -o run -- python3 bench.py --workload $WL --no-cpu-baseline --no-converge > $OUT/stats_$WL.log 2>&1 || exit 1
done
echo "== label counts above 16 (matrix-core kernels): bench lines, kernel stats and counters at Q = 32" >&2
for WL in Q32 Q48 Q64; do
  python3 bench.py --workload $WL --no-cpu-baseline --converge > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err || echo "bench $WL failed" >&2
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_Q32 -o run -- python3 bench.py --workload Q32 --no-cpu-baseline --no-converge > $OUT/stats_Q32.log 2>&1 || echo "stats Q32 failed" >&2
echo "== bench C3, 200 timed steps" >&2
python3 bench.py --workload C3 --steps 200 --warmup 5 --no-cpu-baseline --no-converge > $OUT/bench_C3_200steps.json 2> $OUT/bench_C3_200steps.err || exit 1
echo "== bench C3 --gather messages" >&2
python3 bench.py --workload C3 --no-cpu-baseline --gather messages > $OUT/bench_C3_messages.json 2> $OUT/bench_C3_messages.err || exit 1
echo "== -m infer phases at C3 (fused reduction pass), kernel stats" >&2
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_infer_C3 -o run -- python3 tools/time_infer_phases.py C3 > $OUT/infer_phases_C3.log 2>&1 || echo "infer phases failed" >&2
python3 tools/time_learn.py > $OUT/learn_C5.log 2>&1 || echo "time_learn failed" >&2
echo "== per-rank budget of the 8-rank C3 plan" >&2
for CH in 1 2 4; do
  SBMBP_SHARD_CHUNKS=$CH python3 tools/shard_budget.py C3 8 0 20 $OUT/budget_c3_w8_c$CH.json 2>/dev/null | tail -1 >&2
done
echo "== the multi-process path on this one GPU (3 ranks, callback transport over gloo): functional rehearsal" >&2
SBMBP_REHEARSAL=1 python3 bench.py --gpus 3 --workload small --no-cpu-baseline > $OUT/bench_rehearsal3.json 2> $OUT/bench_rehearsal3.err || echo "rehearsal failed" >&2
echo "== RCCL communicators with one rank" >&2
python3 bench.py --force-sharded --no-cpu-baseline > $OUT/bench_C3_rccl1.json 2> $OUT/bench_C3_rccl1.err || echo "rccl1 failed" >&2
find $OUT -name "*kernel_trace.csv" -delete
echo "profile round $R done" >&2

#!/bin/bash
# per-kernel PMC counters of the marginal-gather sweep, one rocprofv3 pass per counter: tools/pmc_kernel.sh C4 "SQ_WAVE_CYCLES SQ_WAIT_ANY ..."
set -u
WL=$1
export TMPDIR=/tmp
OUT=gpurun_out/pmc_$WL
mkdir -p $OUT
for CTR in $2; do
  rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d $OUT/$CTR -o run -- python3 bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline --no-converge > $OUT/$CTR.log 2>&1 || { echo "pmc $CTR failed"; tail -2 $OUT/$CTR.log; continue; }
  python3 - <<PY
import csv, glob
f = glob.glob("$OUT/$CTR/**/*counter_collection.csv", recursive=True)
vals = {}
for row in csv.DictReader(open(f[0])):
    k = row["Kernel_Name"].split("(")[0][:40]
    vals.setdefault((k, row["Counter_Name"]), []).append(float(row["Counter_Value"]))
for (k, c), v in sorted(vals.items()):
    if "k_sweep" in k:
        print("%-42s %-26s mean %.4g  (n=%d)" % (k, c, sum(v) / len(v), len(v)))
PY
  find $OUT/$CTR -name "*kernel_trace.csv" -delete
done

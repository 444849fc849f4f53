set -e
mkdir -p gpurun_out/rb
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/rb/tests.log 2>&1 || { tail -40 gpurun_out/rb/tests.log; exit 1; }
tail -3 gpurun_out/rb/tests.log
for w in C4 C3 C5 C2; do timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > gpurun_out/rb/$w.json 2>gpurun_out/rb/$w.err || { tail -5 gpurun_out/rb/$w.err; exit 1; }; python - <<PY
import json; d=json.loads(open("gpurun_out/rb/$w.json").read().strip().splitlines()[-1]); c=d.get("converge") or {}; print("$w", "%.4g"%d["value"], "%.4f"%d["ms_per_step"], "%.4f"%d["roofline"]["kernel_ms"], "%.3f"%d["roofline"]["frac"], c.get("sweeps"), c.get("ms_per_sweep"), c.get("edge_msg_per_s"))
PY
done
for w in C2 C4; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rb/prof_$w -o run -- python3 bench.py --workload $w --no-cpu-baseline --no-converge > gpurun_out/rb/prof_$w.log 2>&1
find gpurun_out/rb/prof_$w -name "*kernel_stats.csv" | head -1 | xargs -I{} python3 -c "
import csv,sys
for r in list(csv.reader(open('{}')))[:6]: print('$w', r[0][:40], r[1:4], r[5:7])"
done

set -e
mkdir -p gpurun_out/rb
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "extreme_hub or powerlaw or c4_c5" > gpurun_out/rb/tests.log 2>&1 || { tail -40 gpurun_out/rb/tests.log; exit 1; }
tail -3 gpurun_out/rb/tests.log
for w in C4; do timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --no-converge > gpurun_out/rb/$w.json 2>gpurun_out/rb/$w.err || { tail -5 gpurun_out/rb/$w.err; exit 1; }; python - <<PY
import json; d=json.loads(open("gpurun_out/rb/$w.json").read().strip().splitlines()[-1]); print("$w", "%.4g"%d["value"], "%.4f"%d["ms_per_step"], "%.4f"%d["roofline"]["kernel_ms"], "%.3f"%d["roofline"]["frac"], d["config"].get("hub_edges"))
PY
done

set -e
mkdir -p gpurun_out/rb
timeout -k 10 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/rb/tests.log 2>&1 || { tail -40 gpurun_out/rb/tests.log; exit 1; }
tail -2 gpurun_out/rb/tests.log
for CH in 1 4; do SBMBP_SHARD_CHUNKS=$CH timeout -k 10 400 python3 tools/shard_budget.py C3 8 0 20 gpurun_out/rb/budget_c$CH.json 2>/dev/null | tail -1 | cut -c 200-; done
timeout -k 10 400 python3 bench.py --force-sharded --no-cpu-baseline --no-converge 2>/dev/null | tail -1 | cut -c1-200

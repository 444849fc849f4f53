set -e
mkdir -p gpurun_out/rb
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/rb/tests.log 2>&1 || { tail -40 gpurun_out/rb/tests.log; exit 1; }
tail -3 gpurun_out/rb/tests.log
SBMBP_SHARD_CHUNKS=1 timeout -k 10 400 python3 tools/shard_budget.py C3 8 0 20 gpurun_out/rb/budget_c1.json 2>/dev/null | tail -1

for V in none nosend nostage noboth; do
  if [ $V = none ]; then unset SBMBP_LIB; else export SBMBP_LIB=$PWD/sbm-bp_amd/csrc/variants/libsbmbp_$V.so; fi
  for CH in 1 4; do
    SBMBP_SHARD_CHUNKS=$CH python3 tools/shard_budget.py C3 8 0 20 2>&1 | grep -v amdgpu.ids | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$V', d['chunks'], d['chunk_kernels_ms'], d['fold_gather_finalize_ms'])"
  done
done

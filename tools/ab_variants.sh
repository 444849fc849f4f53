#!/bin/bash
# A/B of tuning variants built by sbm_bp_amd.build.build_variant: tools/ab_variants.sh "v1 v2 ..." "C4 C3 ..."
mkdir -p gpurun_out/ab
for V in $1; do
  export SBMBP_LIB=$PWD/sbm-bp_amd/csrc/variants/libsbmbp_$V.so
  for WL in $2; do
    timeout -k 10 200 python bench.py --workload $WL --no-cpu-baseline --no-converge --steps 30 > gpurun_out/ab/${V}_$WL.json 2> gpurun_out/ab/${V}_$WL.err || { echo "$V $WL failed"; tail -3 gpurun_out/ab/${V}_$WL.err; }
    python3 -c "
import json,sys
d=json.loads([l for l in open('gpurun_out/ab/${V}_$WL.json') if l.startswith('{')][-1])
print('$V $WL ms/step %.4f kernel_ms %.4f frac %.3f'%(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
  done
done

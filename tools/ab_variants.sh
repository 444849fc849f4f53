#!/bin/bash
# A/B of tuning variants built by sbm_bp_amd.build.build_variant: tools/ab_variants.sh "v1 v2 ..." "C4 C3 ..." [reps]
# (reps > 1: each (variant, workload) is run that many times, interleaved, and the minimum kernel time is printed too: the
# box-to-box and run-to-run spread is 1 - 2 %, more than many of the effects being measured)
mkdir -p gpurun_out/ab
REPS=${3:-1}
for I in $(seq 1 $REPS); do
for V in $1; do
  export SBMBP_LIB=$PWD/sbm-bp_amd/csrc/variants/libsbmbp_$V.so
  for WL in $2; do
    timeout -k 10 200 python bench.py --workload $WL --no-cpu-baseline --no-converge --steps 30 > gpurun_out/ab/${V}_${WL}_$I.json 2> gpurun_out/ab/${V}_${WL}_$I.err || { echo "$V $WL failed"; tail -3 gpurun_out/ab/${V}_${WL}_$I.err; }
  done
done
done
python3 - "$1" "$2" $REPS <<'PY'
import json, sys
vs, wls, reps = sys.argv[1].split(), sys.argv[2].split(), int(sys.argv[3])
for wl in wls:
    for v in vs:
        ks, ms = [], []
        for i in range(1, reps + 1):
            try:
                d = json.loads([l for l in open('gpurun_out/ab/%s_%s_%d.json' % (v, wl, i)) if l.startswith('{')][-1])
                ks.append(d['roofline']['kernel_ms']); ms.append(d['ms_per_step'])
            except Exception:
                pass
        if ks:
            print('%-14s %-5s kernel_ms min %.4f (all: %s)  ms/step min %.4f' % (v, wl, min(ks), ' '.join('%.4f' % k for k in ks), min(ms)))
PY

#!/bin/bash
# PMC counters of every sweep-path kernel of one bench workload, one rocprofv3 pass per counter GROUP (SQ has 8 slots per
# pass, FETCH_SIZE and WRITE_SIZE need a pass of their own: MI355X_MICROARCH.md "rocprofv3 PMC slots"), never combined with
# a runtime trace. Writes profiles/<round>_<workload>_pmc.json (per kernel, per launch) with the hash of the kernel sources.
#   tools/pmc_workload.sh r03 C4 [extra bench flags]
set -u
R=$1; WL=$2; shift 2
export TMPDIR=/tmp
OUT=gpurun_out/pmc_${R}_$WL
mkdir -p $OUT
python3 -c "from bench import source_sha; print(source_sha())" > $OUT/source_sha.txt
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
G2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
G3="FETCH_SIZE"
G4="WRITE_SIZE"
G5="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum"
G6="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum GRBM_GUI_ACTIVE"
G7="SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES"   # matrix-core issue (the kernels of label counts above 16)
i=0
for G in "$G1" "$G2" "$G3" "$G4" "$G5" "$G6" "$G7"; do
  i=$((i+1))
  echo "== pmc group $i: $G" >&2
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT/g$i -o run -- python3 bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline --no-converge "$@" > $OUT/g$i.log 2>&1 || { echo "pmc group $i failed" >&2; tail -3 $OUT/g$i.log >&2; }
  find $OUT/g$i -name "*kernel_trace.csv" -delete
done
python3 tools/summarise_pmc.py $R $WL

#!/bin/bash
# SQ / TCC counters of the conv kernels at 16 sequences per launch (own passes, no tracing)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp; mkdir -p gpurun_out
Q="--no_cpu_baseline --no_roofline --no_sub_records --percentile_frames 0"
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|TCC_[A-Z_0-9a-z]*\|GRBM_[A-Z_]*\|TCP_[A-Z_0-9a-z]*" | sort -u > gpurun_out/counters_list.txt; wc -l gpurun_out/counters_list.txt
pass() {  # name, tp_kernel, counters...
  local name=$1 tpk=$2; shift 2
  (cd /tmp && DYB_TP_KERNEL=$tpk timeout 400 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$name -o pmc -- python $R/bench.py --seqs 16 --steps 2 --warmup 1 $Q) > gpurun_out/pmc_$name.log 2>&1
  f=$(find gpurun_out/pmc_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python tools/pmc_multi.py $f gpurun_out/pmc_$name.json igemm > gpurun_out/pmc_$name.txt 2>&1; else echo "no csv for $name"; tail -5 gpurun_out/pmc_$name.log; fi
  rm -rf gpurun_out/pmc_$name
}
pass A_tp 1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
pass A_old 0 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
pass B_tp 1 TCC_HIT_sum TCC_MISS_sum SQ_BUSY_CYCLES SQ_WAVES
pass C_tp 1 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
head -60 gpurun_out/pmc_A_tp.txt

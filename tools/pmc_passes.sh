#!/bin/bash
# rocprofv3 PMC passes over a short bench run (separate passes; never combined with trace domains other than kernel-trace).
#   tools/pmc_passes.sh <precision> <tag>     -> gpurun_out/pmc_<tag>.json (+ .md)
P=${1:-bf16x3}; TAG=${2:-$P}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
export LSDM_STREAMS=1    # counters per kernel: no overlapping kernels (see round_profile.sh)
CMD="python $R/bench.py --precision $P --steps 2 --warmup 1 --no-cpu-baseline --no-others --no-parity --no-hoisted --no-configs"
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET -f csv -d $OUT/p$i -o p$i -- $CMD > $OUT/p$i.log 2>&1 || tail -3 $OUT/p$i.log
done
python $R/tools/summarize_rocprof.py pmc $OUT $R/gpurun_out/pmc_$TAG.json > /dev/null
python $R/tools/pmc_table.py $R/gpurun_out/pmc_$TAG.json > $R/gpurun_out/pmc_$TAG.md
rm -rf $OUT
cat $R/gpurun_out/pmc_$TAG.md

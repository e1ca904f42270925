#!/bin/bash
# Memory-path PMC passes (TA / TCP / TCC) over a short kbench run: tools/pmc_mem.sh <tag> [env ...]  -> gpurun_out/pmcmem_<tag>.json
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
OUT=$R/gpurun_out/pmcmem_$TAG; rm -rf $OUT; mkdir -p $OUT
export LSDM_STREAMS=1
CMD="python $R/tools/kbench.py --steps 4 --modes invariant"
i=0
for SET in "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" \
           "TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_ACCESSES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  env "$@" timeout 300 rocprofv3 --kernel-trace --pmc $SET -f csv -d $OUT/p$i -o p$i -- $CMD > $OUT/p$i.log 2>&1 || tail -3 $OUT/p$i.log
done
python $R/tools/summarize_rocprof.py pmc $OUT $R/gpurun_out/pmcmem_$TAG.json > /dev/null
rm -rf $OUT
python - <<PY
import json
d=json.load(open("$R/gpurun_out/pmcmem_$TAG.json"))
for k,v in d.items():
    if any(s in k for s in ("fp_d_kernel","fp_c_kernel","sa_w_kernel","sa_p_kernel","sa_ur","fp_pre_kernel<128","fp_pre_kernel<256")):
        print(k[:50])
        for c in sorted(v): print("    %-44s %14.0f  (%d launches)"%(c, v[c]["mean_per_launch"], v[c]["launches"]))
PY

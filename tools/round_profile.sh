#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel-trace stats of the bench command + PMC passes (separate runs).
#   tools/round_profile.sh <tag> [precision]       -> gpurun_out/<tag>_kernel_stats.{csv,md}, gpurun_out/pmc_<tag>.{json,md}
TAG=${1:-r01_b}; P=${2:-bf16x3}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd /tmp && export TMPDIR=/tmp
# per-kernel durations and counters are attributable only when kernels do not overlap: the dense chains of a group on ONE stream
# (the library's default runs alternate steps on a side stream; bench.py's own per-stage event pass is serialised the same way)
export LSDM_STREAMS=1
OUT=$R/gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT -o bench -- python $R/bench.py --precision $P --steps 10 --warmup 2 \
    --no-cpu-baseline --no-others --no-parity --no-hoisted --no-configs > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $OUT/bench.err || tail -5 $OUT/bench.err
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp "$F" $R/gpurun_out/${TAG}_kernel_stats.csv
python $R/tools/summarize_rocprof.py stats $OUT $R/gpurun_out/${TAG}_kernel_stats.md > /dev/null
rm -rf $OUT
cd $R && tools/pmc_passes.sh $P $TAG > /dev/null
head -30 $R/gpurun_out/${TAG}_kernel_stats.md

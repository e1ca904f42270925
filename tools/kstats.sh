#!/bin/bash
# Per-kernel durations of the bench command under rocprofv3 (kernel trace only, one stream so that kernels do not overlap), on the GPU box:
#   [ENV=..] tools/kstats.sh <tag> [grep pattern]     -> gpurun_out/<tag>_kstats.csv, matching lines on stdout
TAG=${1:-k}; PAT=${2:-.}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd /tmp && export TMPDIR=/tmp
export LSDM_STREAMS=${LSDM_STREAMS:-1}
OUT=/tmp/kstats_$TAG; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT -o bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-others --no-parity \
    --no-hoisted --no-configs > /dev/null 2> $OUT/bench.err || tail -5 $OUT/bench.err
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
mkdir -p $R/gpurun_out; cp "$F" $R/gpurun_out/${TAG}_kstats.csv
grep -E "$PAT" "$F" | cut -c1-200
rm -rf $OUT

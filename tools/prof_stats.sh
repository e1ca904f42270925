#!/bin/bash
# rocprofv3 kernel-trace stats of tools/kbench.py on the GPU box:  tools/prof_stats.sh <tag> [kbench args...]  (env LSDM_* applies)
TAG=$1; shift
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_$TAG -o kb -- python $R/tools/kbench.py "$@" > /tmp/prof_$TAG.log 2>&1 || tail -5 /tmp/prof_$TAG.log
python $R/tools/summarize_rocprof.py stats /tmp/prof_$TAG $R/gpurun_out/${TAG}_kernel_stats.md > /dev/null
cp $(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_kernel_stats.csv 2>/dev/null

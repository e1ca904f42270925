#!/bin/bash
# Kernel TIMELINE of the bench command (default two-lane schedule) under rocprofv3, on the GPU box:  tools/ktrace.sh <tag> [steps]
#   -> gpurun_out/<tag>_ktrace.csv (kernel name, queue, start, end); analyse with tools/ktrace_gaps.py
TAG=${1:-t}; K=${2:-20}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd /tmp && export TMPDIR=/tmp
OUT=/tmp/ktrace_$TAG; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace -f csv -d $OUT -o bench -- python $R/bench.py --steps $K --warmup 5 --no-cpu-baseline --no-others --no-parity \
    --no-hoisted --no-configs > /dev/null 2> $OUT/bench.err || tail -5 $OUT/bench.err
F=$(find $OUT -name "*kernel_trace.csv" | head -1)
mkdir -p $R/gpurun_out
python - "$F" $R/gpurun_out/${TAG}_ktrace.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["kernel", "queue", "start_ns", "end_ns"])
for r in rows:
    w.writerow([r["Kernel_Name"][:90], r.get("Queue_Id", ""), r["Start_Timestamp"], r["End_Timestamp"]])
print(len(rows), "dispatches")
PY
rm -rf $OUT

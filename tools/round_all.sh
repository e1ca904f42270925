#!/bin/bash
# Everything a round's profiles/ entry consists of, on the GPU box, at the current build:
#   tools/round_all.sh <tag>   -> gpurun_out/<tag>_{kernel_stats.csv,kernel_stats.md,bench.json,bench_k20.json,loop_wall.txt,traffic.json}, pmc_<tag>.{json,md}, pmcmem_<tag>.json
TAG=$1
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
tools/round_profile.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1
tools/pmc_mem.sh $TAG > gpurun_out/${TAG}_pmc_mem.txt 2>&1
python tools/make_traffic.py bf16x3=gpurun_out/pmc_$TAG.json > /dev/null 2>&1 && cp profiles/traffic.json gpurun_out/${TAG}_traffic.json
python bench.py --steps 50 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_k20.json 2>> gpurun_out/${TAG}_bench.err
python tools/loop_wall.py > gpurun_out/${TAG}_loop_wall.txt 2>&1
tail -2 gpurun_out/${TAG}_loop_wall.txt
python - <<PY
import json
for f in ("bench", "bench_k20"):
    d = json.loads(open("gpurun_out/${TAG}_%s.json" % f).read().strip().splitlines()[-1])
    print(f, d["value"], d["unit"], "ms/step", d["ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"].get("traffic"))
PY

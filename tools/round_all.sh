#!/bin/bash
# Everything a round's profiles/ entry consists of, on the GPU box, at the current build:
#   tools/round_all.sh <tag>   -> gpurun_out/<tag>_{bench_k20.json,bench.json,kernel_stats.csv,kernel_stats.md,loop_wall.txt,traffic.json,pmc_mem.txt}, pmc_<tag>.{json,md}, pmcmem_<tag>.json
# The two bench records come first, on the fresh box, the driver's command (--steps 20 --warmup 5) before the default one; the rocprofv3
# passes (kernel trace, PMC, memory-path PMC: ~100 s of GPU work) after them.
TAG=$1
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_k20.json 2> gpurun_out/${TAG}_bench.err
python bench.py --steps 50 --warmup 7 > gpurun_out/${TAG}_bench.json 2>> gpurun_out/${TAG}_bench.err
python tools/loop_wall.py > gpurun_out/${TAG}_loop_wall.txt 2>&1
tools/round_profile.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1
tools/pmc_mem.sh $TAG > gpurun_out/${TAG}_pmc_mem.txt 2>&1
# profiles/traffic.json is what bench.py's roofline.traffic reads: regenerated from THIS tag's PMC pass (and copied back with the run's outputs --
# the GPU box's working copy is scratch); the round fails loudly if the file does not name this tag as its source
python tools/make_traffic.py bf16x3=gpurun_out/pmc_$TAG.json > /dev/null 2>&1 && cp profiles/traffic.json gpurun_out/${TAG}_traffic.json
python - <<PY || { echo "round_all: profiles/traffic.json was NOT regenerated from pmc_$TAG.json"; exit 1; }
import json, sys
sys.exit(0 if json.load(open("profiles/traffic.json")).get("bf16x3", {}).get("_source") == "pmc_$TAG.json" else 1)
PY
tail -2 gpurun_out/${TAG}_loop_wall.txt
python - <<PY
import json
for f in ("bench_k20", "bench"):
    d = json.loads(open("gpurun_out/${TAG}_%s.json" % f).read().strip().splitlines()[-1])
    print(f, d["value"], "ms/step", d["ms_per_step"], "strict", d["strict_recompute"]["value"], "roofline", d["roofline"]["kernel"], d["roofline"]["frac"])
PY

#!/bin/bash
# Quick GPU check used while tuning: stage report (parity vs oracle) + a short bench in the given precisions.
#   tools/quick_gpu.sh "bf16x3 bf16"
mkdir -p gpurun_out
for P in ${1:-bf16x3}; do
  LSDM_PRECISION=$P timeout 300 python tools/gpu_stage_report.py > gpurun_out/stage_$P.log 2>&1
  grep -E "precision|sa4.new_points|fp2.out|pcd|x \(|out  |guiding" gpurun_out/stage_$P.log
  timeout 200 python bench.py --precision $P --steps 10 --warmup 2 --no-cpu-baseline $BENCH_FLAGS > gpurun_out/bench_$P.json 2> gpurun_out/bench_$P.err || tail -5 gpurun_out/bench_$P.err
  python - <<PY
import json
r=json.loads(open('gpurun_out/bench_$P.json').read().strip().splitlines()[-1])
print('$P', 'steps/s %.1f  ms/step %.3f'%(r['value'], r['ms_per_step']))
for k in r['roofline']['kernels'][:14]: print('   %-32s %8.4f ms  %7.1f TF'%(k['name'],k['avg_ms'],k['tflops']))
PY
done

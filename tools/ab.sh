#!/bin/bash
# Same-box A/B of library switches: tools/ab.sh "<stage-grep>" "ENV1=.. ENV2=.." "ENV=.." ...   (each argument one environment; "" = defaults)
# Boxes differ by +-3 % -- only figures from one gpurun call are comparable.
pat=$1; shift
mkdir -p gpurun_out
i=0
for e in "$@"; do
  env $e python tools/kbench.py --steps 21 --modes ${AB_MODES:-invariant} > gpurun_out/ab_$i.log 2>&1
  echo "[$e]"; grep -E "^==|$pat" gpurun_out/ab_$i.log | cut -c1-140
  i=$((i+1))
done

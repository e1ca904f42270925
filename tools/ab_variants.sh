#!/bin/bash
# Same-box A/B of library VARIANTS (tools/build_variant.sh): tools/ab_variants.sh "<stage-grep>" name1 name2 ...   ("base" = the in-tree library)
# Each variant is copied over lsdm_amd/liblsdm_hip.so on the GPU box's scratch copy of the repo (the original is restored at the end).
pat=$1; shift
mkdir -p gpurun_out
cp lsdm_amd/liblsdm_hip.so /tmp/liblsdm_base.so
for v in "$@"; do
  if [ "$v" = base ]; then cp /tmp/liblsdm_base.so lsdm_amd/liblsdm_hip.so; else cp lsdm_amd/variants/$v.so lsdm_amd/liblsdm_hip.so; fi
  python tools/kbench.py --steps ${AB_STEPS:-21} --modes ${AB_MODES:-invariant} ${AB_ARGS} > gpurun_out/abv_$v.log 2>&1
  echo "[$v]"; grep -E "^==|$pat|bit for bit|parity" gpurun_out/abv_$v.log | cut -c1-150
done
cp /tmp/liblsdm_base.so lsdm_amd/liblsdm_hip.so

#!/bin/bash
# Build a variant of liblsdm_hip.so for same-box A/B runs:  tools/build_variant.sh <name> "<extra hipcc flags, e.g. -DLSDM_X_FOO=1>"
#   -> lsdm_amd/variants/<name>.so (git-ignored, travels to the GPU box); tools/ab_variants.sh runs kbench with each of them.
set -e
NAME=$1; FLAGS=$2
R=$(cd "$(dirname "$0")/.." && pwd)
O=/tmp/lsdm_var_$NAME; mkdir -p $O $R/lsdm_amd/variants
cd $R/lsdm_amd/csrc
pids=()
for f in *.hip; do
  extra="-Xclang -target-feature -Xclang -packed-fp32-ops"; [ "$f" = geometry.hip ] && extra="-ffp-contract=off -fno-slp-vectorize"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -Wno-inline-asm $extra $FLAGS -c $f -o $O/${f%.hip}.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map -o $R/lsdm_amd/variants/$NAME.so $O/*.o
ls -la $R/lsdm_amd/variants/$NAME.so

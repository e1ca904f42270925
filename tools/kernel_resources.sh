#!/bin/bash
# Register / LDS / occupancy table of every kernel in lsdm_amd/csrc (hipcc -Rpass-analysis=kernel-resource-usage; no GPU needed).
#   tools/kernel_resources.sh [file.hip ...]  ->  one line per kernel: VGPRs AGPRs scratch occupancy LDS name
cd "$(dirname "$0")/../lsdm_amd/csrc" || exit 1
files=("$@"); [ ${#files[@]} -eq 0 ] && files=(*.hip)
for f in "${files[@]}"; do
  extra="-Xclang -target-feature -Xclang -packed-fp32-ops"; [ "$f" = geometry.hip ] && extra="-ffp-contract=off"; [ "$f" = api.hip -o "$f" = metrics.hip ] && extra=""
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  awk -v F="$f" '/Function Name:/ {n=$0; sub(/.*Function Name: /,"",n); sub(/ \[.*/,"",n)}
       / VGPRs:/ {v=$(NF-1)} /AGPRs:/ {a=$(NF-1)} /ScratchSize/ {s=$(NF-1)} /Occupancy/ {o=$(NF-1)}
       /LDS Size/ {l=$(NF-1); cmd="c++filt " n; cmd | getline d; close(cmd); printf "%-16s v=%3d a=%3d scratch=%4d occ=%d lds=%6d  %s\n", F, v, a, s, o, l, substr(d,1,110)}'
done

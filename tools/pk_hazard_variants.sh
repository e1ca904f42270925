#!/bin/bash
# Packed-fp32 hazard experiments (round 5): builds of the library WITHOUT -packed-fp32-ops and with fp_ws_kernel's interpolation
# left to hipcc's packing, in a few forms; each is run through the fp2 determinism stress (LSDM_FP2_WS=1: the round-4 kernel whose gather waves failed).
#   tools/pk_hazard_variants.sh build      (here: needs hipcc only)      -> lsdm_amd/variants/pk_*.so
#   tools/pk_hazard_variants.sh run        (on the GPU box)              -> gpurun_out/r05_pk_hazard.txt
R=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = build ]; then
  build() {  # name, flags: like tools/build_variant.sh but with NO -packed-fp32-ops anywhere
    O=/tmp/lsdm_var_$1; mkdir -p $O $R/lsdm_amd/variants; cd $R/lsdm_amd/csrc
    for f in *.hip; do
      extra=""; [ "$f" = geometry.hip ] && extra="-ffp-contract=off"
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-function -Wno-inline-asm $extra $2 -c $f -o $O/${f%.hip}.o &
    done; wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map -o $R/lsdm_amd/variants/$1.so $O/*.o
  }
  build pk_scalar ""                                   # control: interp3 as scalar instructions (the shipped source), packing allowed elsewhere
  build pk_packed "-DLSDM_INTERP_PACKED=1"             # round 4's failing form
  build pk_vgprw  "-DLSDM_INTERP_PACKED=2"             # packed, weights in VGPRs
  build pk_nops   "-DLSDM_INTERP_PACKED=1 -DLSDM_PK_NOPS=1"   # packed, 32 idle cycles after the wake-up
  ls -la $R/lsdm_amd/variants/pk_*.so
  exit 0
fi
cd $R; mkdir -p gpurun_out; cp lsdm_amd/liblsdm_hip.so /tmp/liblsdm_base.so
{
for v in pk_scalar pk_packed pk_vgprw pk_nops; do
  cp lsdm_amd/variants/$v.so lsdm_amd/liblsdm_hip.so
  echo "== $v"
  LSDM_FP2_WS=1 N=${PK_RUNS:-150} LOOP_N=0 timeout 600 python tools/determinism_stress.py 2>&1 | grep -E "determinism stress"
done
} > gpurun_out/r05_pk_hazard.txt 2>&1
cp /tmp/liblsdm_base.so lsdm_amd/liblsdm_hip.so
cat gpurun_out/r05_pk_hazard.txt

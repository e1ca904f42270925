#!/bin/bash
# quick parity subset with a variant library:  tools/ab_variants_test.sh <variant> [pytest -k expression]
v=$1; k=${2:-"split_bf16 or odd_batch or invariant_motion or deterministic"}
cp lsdm_amd/liblsdm_hip.so /tmp/liblsdm_base.so
[ "$v" != base ] && cp lsdm_amd/variants/$v.so lsdm_amd/liblsdm_hip.so
python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "$k" 2>&1 | tail -5
cp /tmp/liblsdm_base.so lsdm_amd/liblsdm_hip.so

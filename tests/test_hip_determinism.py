"""Run-to-run determinism of the whole forward under load (B = 16, 144 clouds: every persistent kernel runs several tiles per
work-group).  The reference is deterministic (models/sdm.py, pointnet2_utils.py have no atomics); so is every kernel here by
construction -- this test is the net for intra-kernel races and hardware hazards, which only show with the chip full
(tools/determinism_stress.py is the long form; round 4 found one in fp_ws_kernel with it: DESIGN.md "Round 4")."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("precision", ["bf16x3", "f32"])
def test_repeated_forward_is_bit_identical_at_b16(precision):
    from determinism_stress import stress
    counts = stress(B=16, N=60 if precision == "bf16x3" else 12, precision=precision)
    assert not any(counts.values()), counts


@pytest.mark.parametrize("entry", ["invariant", "strict", "hoisted"])
def test_repeated_sampler_calls_are_bit_identical_at_b64(entry):
    """The benched configuration (B = 64: 576 clouds, 36 tiles per persistent work-group), six steps per call through each bulk entry point."""
    from determinism_stress import stress_loop
    assert stress_loop(B=64, K=6, N=12, precision="bf16x3", entry=entry) == 0

"""GPU: a time-boxed random-shape sweep of the kernels against their references (tests/tools/fuzz_kernels.py): GEMMs and
weight gradients exact on integer data at random (M, N, K), attention forward / backward / fused bias gradient at random
(B, T, H) against fp32 torch and the rounding-matched statement, LayerNorm, FPS and kNN against the oracle at random sizes.
The parametrised tests list the shapes of the model; this walks between them (1,200 cases in 200 s ran clean when the
tool was written; here ~20 s with a fixed seed)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'tools'))


def test_random_shape_sweep_of_the_kernels():
    import fuzz_kernels
    n, failures = fuzz_kernels.run(20.0, seed=2024, verbose=False)
    assert n >= 40, n
    assert not failures, failures[:5]

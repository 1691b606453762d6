"""The two randomised differential checkers as COLLECTED `-m gpu` tests (round 6: `tests/fuzz_gpu.py` was only reached through a
subprocess in tests/test_gpu_fused.py with one seed, `tests/fuzz_rowlazy.py` by nobody): seeded, time-boxed, at least 50 random
cells each, in process -- a failure names the cell.
  * fuzz_gpu: rasterization() (one autograd node) against the operator-by-operator composition of the same kernels and, every fifth
    cell, against the CPU oracle, over random sizes / cameras / channel counts / render modes / degenerate inputs; the neighbour
    kernels against their PyTorch formulations;
  * fuzz_rowlazy: the exact row-lazy optimizer against the streaming fused Adam over random visibility patterns, traversals,
    refinement events."""
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BUDGET_S = 240.0      # per test: the cells stop here (at least MIN_CELLS are always run)
MIN_CELLS = 50


@pytest.mark.parametrize("seed", [7, 2026])
def test_fuzz_rasterization_cells(hip_lib, oracle, seed):
    from tests import fuzz_gpu
    rng = np.random.default_rng(seed)
    t0, done = time.monotonic(), 0
    for i in range(120):
        cfg = fuzz_gpu.rand_case(rng)
        try:
            fuzz_gpu.check_raster(cfg, with_oracle=(i % 5 == 0))
            if i % 3 == 0:
                fuzz_gpu.check_neighbours(rng)
            if i % 3 == 1:
                fuzz_gpu.check_later_neighbours(rng)
        except Exception as e:      # noqa: BLE001
            raise AssertionError(f"fuzz_gpu seed {seed} cell {i}: {cfg}: {type(e).__name__}: {e}") from e
        done += 1
        if done >= MIN_CELLS and time.monotonic() - t0 > BUDGET_S:
            break
    torch.cuda.synchronize()
    assert done >= MIN_CELLS


@pytest.mark.parametrize("seed", [3, 11])
def test_fuzz_rowlazy_cells(hip_lib, seed):
    from tests import fuzz_rowlazy
    dev = torch.device("cuda")
    t0, done = time.monotonic(), 0
    for i in range(120):
        try:
            fuzz_rowlazy.case(seed * 100003 + i, dev)
        except Exception as e:      # noqa: BLE001
            raise AssertionError(f"fuzz_rowlazy seed {seed} cell {i} (case {seed * 100003 + i}): {type(e).__name__}: {e}") from e
        done += 1
        if done >= MIN_CELLS and time.monotonic() - t0 > BUDGET_S:
            break
    torch.cuda.synchronize()
    assert done >= MIN_CELLS

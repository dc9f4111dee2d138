"""Seeded, bounded slices of the randomised parity sweeps (tools/fuzz_parity.py, fuzz_ba.py, fuzz_pose.py) inside `-m gpu`, so that
the driver's GPU record carries them: every case compares the HIP path, through the C ABI, with the CPU oracle.  The front-end
cases draw a mapping reserve of 0 / 4 / 8 / 16: three quarters of them run the QUEUED extraction kernels (the ones bench.py times)
directly against the oracle.  The cases live in tests/fuzz_cases.py; a case is a function of (seed, index) alone."""
import pytest

import fuzz_cases

pytestmark = pytest.mark.gpu

SEED = 5


@pytest.mark.parametrize("case", range(48))
def test_frontend_fuzz(hiplib, oracle, case):
    ok, tag = fuzz_cases.frontend_case(oracle, SEED, case)
    assert ok, tag


@pytest.fixture(scope="module")
def small_ctx(hiplib):
    ctx = hiplib.Context(640, 480, 500, 1.2, 4, max_images=2)
    yield ctx
    ctx.close()


@pytest.mark.parametrize("case", range(48))
def test_ba_fuzz(hiplib, oracle, small_ctx, case):
    ok, tag = fuzz_cases.ba_case(oracle, small_ctx, SEED, case, max_points=700)
    assert ok, tag


@pytest.mark.parametrize("case", range(120))
def test_pose_fuzz(hiplib, oracle, small_ctx, case):
    ok, tag, dr, dt = fuzz_cases.pose_case(oracle, small_ctx, SEED, case)
    assert ok, tag

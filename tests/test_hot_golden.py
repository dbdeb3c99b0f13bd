"""GPU: the HIP path against the golden outputs of the real reference (tests/golden/*.npz)."""
import pytest

import golden_util as gu
from conftest import small_cases, wide_scan_cases

pytestmark = pytest.mark.gpu
NAMES = [c[0] for c in small_cases()] + [c[0] for c in wide_scan_cases()]


@pytest.fixture(scope="module")
def hot():
    from rsicnv_amd import api
    h = api.RsiHot(0)
    yield h
    h.close()


@pytest.mark.parametrize("name", NAMES)
def test_hip_matches_reference_golden(hot, hotlib, name):
    gu.check_hip_against_golden(hot, hotlib, name)

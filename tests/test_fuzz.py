"""GPU: random chromosomes (length, depth model and mean, events, gaps, pile-ups, uncovered stretches, assembly gaps) under random
flags (-m 11 ... 439, -NB / -MED / -ALL, caps, -NOGC, -nomerge) through the library and through the oracle in a child process:
every array, scalar, status vector and call list must agree, and where the oracle refuses (the reference exits or aborts) the
library must refuse too.  tools/fuzz_probe.py is the same loop for longer sessions (40 cases of seed 1 ran clean in round 4)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.timeout(900)
@pytest.mark.parametrize("seed", [11, 12])
def test_random_chromosomes_and_flags_agree_with_the_oracle(seed):
    import fuzz_probe
    assert fuzz_probe.run(8, seed, max_bins=60_000) == 0

"""-m gpu: a few seeds of the two lane-family soaks (tests/perf/soak_lane_forms.py, soak_lane_shards.py: random ragged problems at padded
rank 32 forced onto the LDS tiles -- power-law and uniform patterns, empty rows, one loss or a loss per column).  Whichever form of the
trial rounds runs (full grid / CSR / gathered waves from chunk or packed lists), whichever form of the stream (padded / compact), dealt or
undealt slot permutations, one handle or 2 .. 6 shards with chunked row sweeps: the same bits and the same line-search counts
(src/algorithms/proxgrad.jl:118-156,162-201 are the loops whose rounds these are).  The long runs: profiles/r06_soak_lane_*.txt."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,first,last", [("soak_lane_forms.py", 3000, 3008), ("soak_lane_shards.py", 3000, 3006)])
def test_a_few_seeds_of_the_lane_soaks(script, first, last):
    env = {k: v for k, v in os.environ.items() if not k.startswith("GLRM_HIP_LANE")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "perf", script), str(first), str(last)], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0 and "0 differences" in p.stdout, (p.stdout[-2000:], p.stderr[-2000:])

"""Bounded runs of the two fuzzers (tools/probes/fuzz_*.py) inside `-m gpu`: every accelerated exact path against
the independent brute-force / streaming kernel of the same operator on random sizes, scales, degenerate and
NaN/Inf/huge inputs.  A few seconds each, different seed per parametrisation."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,seed", [("fuzz_point_in_tet.py", 11), ("fuzz_point_in_tet.py", 12), ("fuzz_surface_ops.py", 21),
                                         ("fuzz_surface_ops.py", 22)])
def test_fuzzer_bounded(cuda, script, seed):
    env = dict(os.environ, FUZZ_SEED=str(seed))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probes", script), "4"], env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    m = re.search(r"fuzz ok: (\d+) random", out.stdout)
    assert m and int(m.group(1)) >= 5, out.stdout[-500:]

"""A fixed-seed slice of the fuzzers (their totals were builder-run only): 40 random detection configurations
(image sizes, sigma / threshold / scale counts, FP16 mode, batches up to 13 images, their pairwise matchings) and 12 random matcher
problems (size mixes, duplicates, ties, full-range bytes, filtered matching) through tools/fuzz_parity.py / tools/fuzz_match.py —
every case bit-exact against the oracle, or the tool exits non-zero and prints the case."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tool, *args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), *map(str, args)], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-1500:])
    return r.stdout


def test_detection_fuzz_slice():
    out = _run("fuzz_parity.py", 20260929, 40, 700, 64)
    assert "cases 40 bad 0" in out


def test_matcher_fuzz_slice():
    out = _run("fuzz_match.py", 20260929, 12, 6000)
    assert "cases 12 bad 0" in out

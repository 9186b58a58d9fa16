"""The two differential fuzzers of scripts/ under pytest, with fewer cases: random soups, scales, materials, modes, depth
caps and chain counts; every render path against one launch per stage AND against the oracle's image, bit for bit
(scripts/fuzz_paths.py), and Scene.trace / Scene.hitsNothing against the exhaustive oracle (scripts/fuzz_parity.py)."""
import os
import runpy
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('script,cases', [('fuzz_paths.py', 16), ('fuzz_parity.py', 10)])
def test_fuzzers_find_no_difference(require_gpu, script, cases, capsys):
    argv = sys.argv
    sys.argv = [script, str(cases)]
    try:
        runpy.run_path(os.path.join(ROOT, 'scripts', script), run_name='__main__')       # exits non-zero on the first mismatch
    except SystemExit as e:
        assert not e.code, capsys.readouterr().out[-2000:]
    finally:
        sys.argv = argv
    assert 'all cases equal' in capsys.readouterr().out

"""The matrix-core quad primitives of csrc/lgssm_m4.h (P = four v_mfma_f32_4x4x1, its bit-exact transpose twin, the rank-one form,
column loads, the natural-order 4x4 solve and its pivot watch) against plain loops: tools/m4_selftest.hip, built by
__graft_entry__.build() (or here, if the binary is missing) and run on the device."""
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


@pytest.mark.gpu
def test_m4_primitives():
    import __graft_entry__ as G
    exe = G.build_selftest()
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "m4_selftest OK" in r.stdout, r.stdout + r.stderr

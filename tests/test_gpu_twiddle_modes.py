"""Every fixed-point twiddle rule the mount evidences, on the GPU, bit for bit.

The rule is a process-wide switch that must be chosen before the first q15 / q31 init, so each mode runs in a fresh
process (tests/twiddle_mode_worker.py): 0 floor, 1 round to nearest (the MVE / Neon copies, arm_mve_tables.c:3769,
arm_neon_tables.c:8951), 2 the fork's own generator for its 8192 / 16384 tables (CMSIS/CMSIS.cpp:445-458 + :309-327),
3 the default (floor up to 4096 points, the fork's generator beyond)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_fixed_point_bit_exact_under_twiddle_rule(mode):
    r = subprocess.run([sys.executable, os.path.join(HERE, "twiddle_mode_worker.py"), str(mode)], capture_output=True, text=True,
                       timeout=240, cwd=os.path.dirname(HERE))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["mode"] == mode and line["cfft_cases"] == 80 and line["rfft_cases"] == 20 and line["golden_cases"] == 36 and line["tables"] == 10

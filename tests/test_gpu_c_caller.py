"""A compiled C host program (tests/c_caller/fft_c32_caller.c: gcc, include/arm_fft_mi355.h,
-larmfft_mi355, no Python in the process) runs arm_cfft_init_f32 / arm_cfft_f32 and the fork's
fft_c32 sequence (CMSIS/CMSIS.cpp:218-234) on the reference's golden vectors."""
import os
import subprocess

import numpy as np
import pytest

import helpers as H

CALLER_DIR = os.path.join(H.ROOT, "tests", "c_caller")
CALLER = os.path.join(CALLER_DIR, "fft_c32_caller")


def _built():
    """The binary is built by __graft_entry__.build() (CPU container); rebuild only if gcc and the library are here."""
    if not os.path.exists(CALLER) and os.path.exists(os.path.join(H.ROOT, "fft_amd", "lib", "libarmfft_mi355.so")):
        subprocess.run(["make", "-C", CALLER_DIR], check=True, capture_output=True)
    return os.path.exists(CALLER)


def test_c_caller_builds_and_links_against_the_header():
    """CPU: the program compiles against the public header alone and links with -larmfft_mi355."""
    assert _built()
    out = subprocess.run(["readelf", "-d", CALLER], capture_output=True, text=True).stdout
    assert "libarmfft_mi355.so" in out


@pytest.mark.gpu
@pytest.mark.parametrize("n,sig", [(1024, "noisy"), (4096, "step"), (64, "noisy")])
def test_c_caller_runs_golden_vectors(n, sig, tmp_path):
    assert _built()
    x, want, _ = H.golden_case("f32", "complex", sig, n)
    fi, fw = str(tmp_path / "in.f32"), str(tmp_path / "want.f32")
    np.asarray(x, np.float32).tofile(fi)
    np.asarray(want, np.float32).tofile(fw)
    r = subprocess.run([CALLER, str(n), fi, fw], capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all C-caller checks passed" in r.stdout

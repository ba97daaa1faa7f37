"""The integer arithmetic of the matrix-pipe MME kernel (profiles/ab/me_mme_fx.hpp) on the host: digit features
-> column sums (what v_mfma_i32_16x16x64_i8 accumulates) -> moments about the query, against exact __int128 arithmetic and the
fp64 sums of the vector kernel.  The same header is compiled into the device code; no GPU needed here."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fixed_point_moments_are_exact_on_the_host():
    src = os.path.join(ROOT, "tests", "native", "test_mme_fx.cpp")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "test_mme_fx")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, src])
        out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout[-2000:] + out.stderr[-2000:]

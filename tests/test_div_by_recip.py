"""The kernels divide by a per-cell / constant denominator with one correctly rounded reciprocal, one multiply and two
FMAs (nus_device.hpp: div_by_recip).  That this equals the IEEE quotient bit for bit -- also for a denominator whose
mantissa is all ones, the exception of Markstein's theorem -- is checked here exhaustively on the CPU."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_div_by_recip_equals_ieee_division_for_every_mantissa(tmp_path):
    src = os.path.join(HERE, "helpers", "div_by_recip_check.c")
    exe = str(tmp_path / "div_by_recip_check")
    flags = ["-O2", "-ffp-contract=off"]
    if "fma" in open("/proc/cpuinfo").read().split():
        flags.append("-mfma")  # hardware fmaf; without it libm's (slower, same results)
    subprocess.run(["gcc", *flags, "-o", exe, src, "-lm"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.startswith("ok "), out.stdout + out.stderr
    assert int(out.stdout.split()[1]) > 150_000_000

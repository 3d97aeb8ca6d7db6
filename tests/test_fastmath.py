"""The in-kernel fp64 routines of csrc/glrm_fastmath.hpp, compiled for the host (same straight-line code; v_rcp_f64 stands in as a
24-bit float reciprocal) and checked by tools/check_fastmath.cpp against long double over 4e6 points per function: exp, log on [1, Inf)
and on (0, 1] incl. denormals, the LogisticLoss pair in the reference's rounding structure (src/losses.jl:298-311: exactly 0 where the
reference's formula is exactly 0, Inf where it overflows), and the special values."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="no host compiler")
def test_host_check_of_the_in_kernel_exp_log_reciprocal(tmp_path):
    exe = str(tmp_path / "cfm")
    subprocess.run(["g++", "-O2", "-ffp-contract=off", os.path.join(ROOT, "tools", "check_fastmath.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    num = lambda pat: float(re.search(pat, out).group(1))
    assert num(r"exp ([0-9.e+-]+)") < 2.5e-16 and num(r"log \(w >= 1\) ([0-9.e+-]+)") < 5e-16 and num(r"logistic derivative ([0-9.e+-]+)") < 6e-16
    assert num(r"fm_log on \(0, 1\] incl\. denormals: max rel err ([0-9.e+-]+)") < 5e-16
    assert "log(0) -inf log(-1) nan log(Inf) inf log(NaN) nan log(1) 0" in out.replace("-nan", "nan")
    assert "nonzero where the reference is exactly 0: 0" in out
    assert "L inf (reference Inf) dL -1 (reference -1)" in out and "z = +800: L 0 (reference 0)" in out
    assert "logistic(1, true) 0.31326168751822286 (reference KAT 0.31326168751822286)" in out  # test/loss_test.jl's value
    assert "logistic(1, false) 1.3132616875182228 (reference KAT 1.3132616875182228)" in out

"""Nested-dissection Cholesky, host side (no GPU): the symbolic phase csrc/nd_symbolic.hpp -- tree, fronts, child ->
parent maps, assembly lists -- drives the plain multifrontal restatement tools/nd_ref.hpp, which must solve random SPD
stencil systems (7-point TV stencil, 13-point sum-of-regularisers stencil; entries up to 1e6 apart) to rounding on
odd shapes and several leaf sizes.  The GPU kernels are checked against the same restatement (tools/nd_unit.hip,
tests/test_gpu_evaluate.py::test_nd_solver_unit_checks).  Replaces the sparse LU behind Julia's `\\` at
/root/reference/src/TVLearningFunctionVec.jl:131,248."""
import os
import re
import subprocess
from conftest import ROOT

EXE = os.path.join(ROOT, "tools", "_bin", "nd_host_check")


def test_multifrontal_restatement_solves_stencil_systems():
    assert os.path.exists(EXE), "built by __graft_entry__.build()"
    out = subprocess.run([EXE, "v"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "all ok" in out.stdout, out.stdout[-3000:]
    assert out.stdout.count("ok  ") >= 60 and "FAIL" not in out.stdout


def test_structure_of_a_1024_grid_is_cubic_not_quartic():
    """George's bounds for nested dissection of an n x n grid: 9.9 n^3 multiplications, 7.75 n^2 log2 n fill.  The
    dense-front tree stays within a small factor of both (and three orders of magnitude below the band's n^4)."""
    out = subprocess.run([EXE, "v", "1024", "32"], capture_output=True, text=True, timeout=300).stdout
    m = re.search(r"1024x1024 tv leaf 32: nodes (\d+) levels (\d+) max front (\d+) max p (\d+) factor ([\d.]+) MB .* flop ([\d.e+]+)", out)
    assert m, out[-2000:]
    fill_mb, flop = float(m.group(5)), float(m.group(6))
    assert int(m.group(3)) <= 1537 and int(m.group(4)) == 1024
    assert fill_mb < 2.0 * 7.75 * 1024 ** 2 * 10 * 8e-6          # < 2 x George's fill
    assert flop < 2.5 * 9.9 * 1024 ** 3                           # multiply-adds, < 2.5 x George
    assert flop < 0.05 * 1024.0 ** 4                              # and far below the band's n * bw^2 = 1.1e12

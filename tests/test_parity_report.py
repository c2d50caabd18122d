"""CPU twin of the per-component pure-relative parity report (tests/parity_report.py): the oracle in the
device's storage modes against every golden E / V / D / W / R episode of the reference.  The -m gpu twin is
tests/test_gpu_golden.py::test_pure_relative_parity_per_component."""
import pytest

import parity_report as pr


@pytest.mark.parametrize("mode", ["float32", "float64"])
def test_pure_relative_parity_per_component_cpu_twin(mode):
    rep = pr.collect(pr.OracleBackend(mode))          # asserts: every value above the bar is a zero crossing
    print("\n" + pr.format_report("VecOracle store_mode=%s vs golden float64 traces" % mode, rep))
    assert rep["samples"] > 250000
    if mode == "float64":
        assert rep["worst"].max() == 0.0              # bit-exact: no exception of any kind
        return
    # north_star's literal bar, wherever a relative error is well-posed: all twelve components
    assert (rep["worst_steady"] <= pr.BAR).all(), rep["worst_steady"]
    # in units of the trajectory's own scale the stored words (29 significant bits) are 10x inside the bar
    assert (rep["worst_range"] <= 1.5e-6).all(), rep["worst_range"]
    # the documented exception, stated: z within centimetres of the ground (and a spin's velocity at its
    # turning point) exceed 1e-5 relative -- and nothing else does
    over = {pr.NAMES[k]: int(v) for k, v in enumerate(rep["over_bar"]) if v}
    assert set(over) <= {"z", "dx", "dy", "dz", "x", "y"} and "z" in over, over
    assert rep["worst"][4] < 1e-3                      # |err| ~ 8e-7 m on |z| >= 1 mm
    assert (rep["worst"][6:] <= pr.BAR).all()          # the attitude half never crosses the bar

"""Runs LAST (file name): every kernel instantiation of the code object must have been launched by some GPU test of this session.

r03 shipped three wrong-image code-generation failures in instantiations no test had ever run; the launch log of that suite (r04)
showed 91 of 176 instantiations never started.  tests/conftest.py points GRT_LOG_LAUNCHES (grt_kernels.hip:note_launch) at a session
file for this process and every child it starts; the instantiations come from the compiler's resource remarks of the shipped object
(grtrans_amd/csrc/kernel_resource_usage.txt, which travels with the built library).  An instantiation that nothing launches gets a test
against its single-frequency / plain twin (test_frequency_batches_on_every_fluid_model) or is not instantiated."""
import os
import re
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def instantiations():
    path = os.path.join(ROOT, "grtrans_amd", "csrc", "kernel_resource_usage.txt")
    if not os.path.exists(path):
        pytest.skip("no resource remarks here (the library was not built in this tree)")
    names = set()
    for line in open(path):
        m = re.search(r"Function Name: (\S+)", line)
        if not m:
            continue
        n = m.group(1)
        mm = re.match(r"_Z\d+(grt_\w+?_kernel)(I.*?EE)?v", n) or re.match(r"(grt_\w+_kernel)$", n) or re.match(r"_Z\d+(grt_\w+_kernel)", n)
        if mm:
            names.add(mm.group(1) + (mm.group(2) if mm.lastindex and mm.lastindex > 1 and mm.group(2) else ""))
    return names


# Tests whose assertions compare device results with the compiled reference (tests/golden/*, made by oracle/_ref), the reference's own
# pickles, or the oracle (itself pinned to the compiled reference).  Everything else in the GPU suite is a self-consistency check
# (tiles = whole image, batch = single frequency, record on = off, A/B of two kernels, properties at full size, plumbing).
REFERENCE_TESTS = re.compile(
    r"vs_reference|vs_the_reference|vs_oracle|vs_compiled_reference|matches_reference|per_step_record|random_cameras_vs|camera_sweep|"
    r"open_cases|near_polar_ray|end_to_end|ten_inclinations|movie|slow_light|nload|polarized_thindisk|polar_camera_grid|"
    r"reference_regression_problem|table_bin_flip|quadrature_survives|delo_step_on_the_device|fits_output|hybrid_emissivities|"
    r"geokerr_rays_abi|standard2_with_many_points|powerlaw_on_grmhd|every_pair_and_integrator|smoke")
# ... and the one self-consistency test the chain below leans on: every frequency-batch (FB > 1) and extra=1 instantiation is held
# BIT-EQUAL to its single-frequency / plain twin there
TWIN_TEST = "test_frequency_batches_on_every_fluid_model"


def _read_log():
    log = os.environ.get("GRT_LOG_LAUNCHES")
    if not log or not os.path.exists(log):
        pytest.skip("no launch log (run the whole GPU suite: tests/conftest.py sets GRT_LOG_LAUNCHES for the session)")
    by = {}
    for line in open(log):
        parts = line.rstrip("\n").split("\t")
        if parts[0].strip():
            by.setdefault(parts[0].strip(), set()).add(parts[1] if len(parts) > 1 else "?")
    if len(by) < 40:
        pytest.skip("only %d distinct kernels in the log: a partial run of the suite" % len(by))
    return by


def test_every_kernel_instantiation_was_launched_by_the_suite():
    by = _read_log()
    launched = set(by)
    inst = instantiations()
    assert len(inst) > 100, len(inst)
    never = sorted(inst - launched)
    assert not never, "%d of %d instantiations never launched: %s" % (len(never), len(inst), never[:12])
    assert not (launched - inst), sorted(launched - inst)[:5]       # (a launch the remarks do not know: the parser or the log is off)


def test_every_kernel_instantiation_is_held_to_the_reference_by_some_test():
    # VERDICT r04 next #3c: "launched" is not "checked".  An instantiation counts as reference-checked when (a) a test of REFERENCE_TESTS
    # launched it, or (b) it is a frequency-batch / extra=1 transfer kernel that TWIN_TEST holds bit-equal to its FB = 1, non-extra twin
    # (same integrator, fluid, emissivity) and that twin satisfies (a).
    by = _read_log()
    inst = instantiations()

    def ref_checked(name):
        return any(REFERENCE_TESTS.search(t) for t in by.get(name, ()))

    def twin(name):      # grt_transfer_kernelILi<I>ELi<F>ELi<E>ELi<FB>ELb<X>ELi<TSH>EE -> the FB = 1, non-extra instantiations of (I, F, E)
        m = re.match(r"grt_transfer_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELi(\d)EE$", name)
        if not m or (m.group(4) == "1" and m.group(5) == "0"):
            return []
        return [n for n in inst if re.match(r"grt_transfer_kernelILi%sELi%sELi%sELi1ELb0ELi\dEE$" % m.group(1, 2, 3), n)]

    unchecked = []
    for n in sorted(inst):
        if ref_checked(n):
            continue
        if TWIN_TEST in by.get(n, ()) and any(ref_checked(t) for t in twin(n)):
            continue
        unchecked.append((n, sorted(by.get(n, ()))[:4]))
    assert not unchecked, "%d of %d instantiations are launched only by self-consistency tests: %s" % (len(unchecked), len(inst), [u[0] for u in unchecked])

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


def test_every_kernel_instantiation_was_launched_by_the_suite():
    log = os.environ.get("GRT_LOG_LAUNCHES")
    if not log or not os.path.exists(log):
        pytest.skip("no launch log (run the whole GPU suite: tests/conftest.py sets GRT_LOG_LAUNCHES for the session)")
    launched = set(l.strip() for l in open(log) if l.strip())
    if len(launched) < 40:
        pytest.skip("only %d distinct kernels in the log: a partial run of the suite" % len(launched))
    inst = instantiations()
    assert len(inst) > 100, len(inst)
    never = sorted(inst - launched)
    assert not never, "%d of %d instantiations never launched: %s" % (len(never), len(inst), never[:12])
    assert not (launched - inst), sorted(launched - inst)[:5]       # (a launch the remarks do not know: the parser or the log is off)

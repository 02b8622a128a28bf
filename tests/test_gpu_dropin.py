"""GPU: the REFERENCE'S OWN drivers judging this backend.

oracle/_ref/{compare,time}_results_dropin_{f32,f64} are the reference's unmodified compare_results.c / time_results.c
+ ann.c + algc.c (its CPU path) compiled where they lie and linked against libapproxnn_hip instead of the reference's
OpenCL objects (oracle/Makefile `dropin`; exactly the substitution INTEGRATION.md describes).  compare_results runs
precomp/query on the "GPU" (this library) and on the reference's CPU path from the same seed and prints the number of
differences (compare_results.c:98-143); it must print 0.  The binaries exist only where /root/reference was available
at build time; they travel to the GPU box with the snapshot."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
REF = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")


def _run(name, *args):
    exe = os.path.join(REF, name)
    if not os.path.exists(exe):
        pytest.skip("%s not built (no /root/reference at build time)" % name)
    out = subprocess.run([exe] + list(args), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    return out.stdout


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_reference_compare_results_precomp_mode(prec):
    out = _run("compare_results_dropin_" + prec, "-o", "2")                      # reference defaults n=1000 d=80 k=10 t=10
    m = re.search(r"Average diffs for comp: (\S+)", out)
    assert m and float(m.group(1)) == 0.0, out


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_reference_compare_results_query_mode(prec):
    out = _run("compare_results_dropin_" + prec, "-n", "3000", "-d", "64", "-y", "100", "-o", "3")
    m = re.search(r"Average diffs for query: (\S+)", out)
    assert m and float(m.group(1)) == 0.0, out


def test_reference_time_results_runs_on_this_backend():
    out = _run("time_results_dropin_f32", "-n", "20000", "-d", "32", "-y", "500", "-o", "5")
    assert "Average time for query (on GPU)" in out

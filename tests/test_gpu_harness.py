"""GPU: the C harnesses (tests/harness) -- the reference-shaped drivers run against the HIP library through
the plain C surface of include/ann.h, with the oracle as the CPU side.  compare_results exits non-zero on any
id difference or distance violation."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
H = os.path.join(os.path.dirname(os.path.abspath(__file__)), "harness")


def _run(args):
    out = subprocess.run(args, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    return out.stdout


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_compare_results_precomp_mode(prec):
    out = _run([os.path.join(H, "compare_results_" + prec), "-o", "2", "-S", "41"])     # reference defaults: n=1000 d=80 k=10
    assert "Average diffs for comp: 0" in out and "PASS" in out and "0 not bit-identical" in out


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_compare_results_query_mode(prec):
    out = _run([os.path.join(H, "compare_results_" + prec), "-n", "5000", "-d", "64", "-y", "200", "-o", "3", "-S", "42"])
    assert "Average diffs for query: 0" in out and "PASS" in out and "0 not bit-identical" in out


@pytest.mark.parametrize("pieces", ["1", "3", "8"])
def test_query_gpu_batch_sent_in_pieces(pieces):
    """query_gpu on a batch large enough to be sent in pieces (>= 1 MB each): the copy kernel of piece i crosses PCIe
    while the host threads fill piece i+1 of the pinned buffer; one hash launch behind the last piece; ragged last piece.
    A fresh process per setting: ANN_HIP_IO_PIECES is read once."""
    env = dict(os.environ, ANN_HIP_IO_PIECES=pieces)
    out = subprocess.run([os.path.join(H, "compare_results_f32"), "-n", "20000", "-d", "128", "-y", "16391", "-o", "2", "-S", "43"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "Average diffs for query: 0" in out.stdout and "PASS" in out.stdout and "0 not bit-identical" in out.stdout


def test_time_results_runs_config1_shape():
    out = _run([os.path.join(H, "time_results_f32"), "-n", "20000", "-d", "32", "-k", "10", "-y", "500", "-o", "3", "-S", "7"])
    assert "queries/s" in out and "on CPU, oracle" in out


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_compare_results_many_seeds(prec):
    """The reference drivers' default shape (n=1000 d=80 k=10 tries=10) over many seeds: EVERY save_t field bit-equal.
    (Round 2 found 1-ulp differences in `bases` of the double build for ~7 % of the seeds: gcc fuses the reference's
    cos/sin pair into one glibc sincos() call, which differs from cos()/sin() in the last bit now and then.)"""
    for seed in range(100, 130):
        out = _run([os.path.join(H, "compare_results_" + prec), "-o", "1", "-S", str(seed)])
        assert "Average diffs for comp: 0\n" in out and "PASS" in out and "0 not bit-identical" in out, (seed, out)


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_test_correctness_tool_gpu_and_cpu_columns_agree(prec):
    """tests/harness/test_correctness (counterpart of the reference's recall driver, test_correctness.c:92-141): the three
    quality numbers for precomp's graph and for query batches; the GPU column and the CPU column (-c: the oracle's
    answers, same scorer) must print the same numbers, the answers being bit-identical."""
    import re
    exe = os.path.join(H, "test_correctness_" + prec)
    for args in (["-o", "2", "-S", "5"], ["-n", "4000", "-d", "64", "-y", "300", "-o", "2", "-S", "6"]):
        g = _run([exe] + args)
        c = _run([exe] + args + ["-c"])
        num = lambda txt: [float(v.rstrip(".")) for v in re.findall(r": ([0-9.e+-]+)", txt)]  # noqa: E731
        assert "Average index score for" in g and "Prob correct" in g and "Max index score" in g
        assert "(on GPU)" in g and "(on CPU)" in c and num(g) == num(c), (g, c)
        assert 0.0 < num(g)[1] <= 1.0

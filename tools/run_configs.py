#!/usr/bin/env python3
"""Run BASELINE.json's five configurations on ONE MI355X and emit one JSON line per configuration.

    python tools/run_configs.py [--out gpurun_out/configs] [--only cfg1,cfg3] [--quick]

  cfg1  N=100k d=32  k=10 Q=1k  float   through the reference's host-pointer ABI (tests/harness/time_results_f32)
  cfg2  N=1M   d=64  k=10 Q=10k float   the same
  cfg3  N=10M  d=128 k=10 Q=10k float   the same, AND bench.py (the metric: inputs and index resident in HBM)
  cfg4  N=40M  d=128 k=10 Q=100k float  bench.py on one GPU (the 8-GPU run is the driver's: bench.py --gpus 8)
  cfg5  N=10M  d=256 k=100 Q=10k double bench.py --dtype f64

Every bench.py line carries its own full-size parity sample against the reference's query_cpu / the oracle
(cpu_baseline.parity_on_sample); the harness lines carry queries/s, the stage-1 kernel's algorithmic GB/s and fraction
of the 8 TB/s HBM peak, the pipelined host rate and the CPU column.  Raw logs go to --out (copy the ones to keep into
profiles/).  cfg4/cfg5 use --data randn: drawing 5 G values from libc random() would take minutes.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(cmd, log):
    t0 = time.time()
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT)
    open(log, "w").write("$ %s\n%s\n---- stderr ----\n%s" % (" ".join(cmd), out.stdout, out.stderr[-4000:]))
    return out.returncode, out.stdout, time.time() - t0


def harness(tag, n, d, k, Q, reps, cpu_q, outdir):
    cmd = [os.path.join(ROOT, "tests", "harness", "time_results_f32"), "-n", str(n), "-d", str(d), "-k", str(k), "-y", str(Q),
           "-o", str(reps), "-S", "12345", "-P", "3", "-C", str(cpu_q)]
    rc, txt, wall = run(cmd, os.path.join(outdir, "%s_time_results.log" % tag))
    line = {"config": tag, "via": "query()/precomp() of ann.h with host pointers (tests/harness/time_results_f32)",
            "workload": "N=%d d=%d k=%d Q=%d float" % (n, d, k, Q), "rc": rc, "wall_s": round(wall, 1)}

    def grab(pat, cast=float):
        m = re.search(pat, txt)
        return cast(m.group(1)) if m else None
    line["precomp_s"] = grab(r"precomp \(with save\) on GPU: ([0-9.]+) s")
    line["query_gpu_qps"] = grab(r"on GPU, host buffers\): \S+\s+=> (\d+) queries/s")
    line["query_gpu_first_call_s"] = grab(r"first call ([0-9.e+-]+)s")
    line["stage1_kernel_GBps"] = grab(r"=> ([0-9.]+) GB/s = ")
    line["stage1_frac_of_hbm_peak"] = (lambda v: None if v is None else round(v / 100, 4))(grab(r"GB/s = ([0-9.]+) % of the 8000"))
    line["whole_call_GBps"] = grab(r"PCIe included\): ([0-9.]+) GB/s")
    line["stream_3_lanes_qps"] = grab(r"3-lane pipeline\): \S+\s+=> (\d+) queries/s")
    line["cpu_reference_qps"] = grab(r"reference's query_cpu, [^)]*\): \S+\s+=> ([0-9.]+) queries/s")
    line["cpu_oracle_qps"] = grab(r"on CPU, oracle, [^)]*\): \S+\s+=> (\d+) queries/s")
    return line


def bench(tag, extra, outdir):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + extra
    rc, txt, wall = run(cmd, os.path.join(outdir, "%s_bench.json.log" % tag))
    js = [l for l in txt.splitlines() if l.startswith("{")]
    line = json.loads(js[0]) if js else {}
    return {"config": tag, "via": "bench.py " + " ".join(extra), "rc": rc, "wall_s": round(wall, 1), "bench": line}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "configs"))
    ap.add_argument("--only", default="cfg1,cfg2,cfg3,cfg4,cfg5")
    ap.add_argument("--quick", action="store_true", help="fewer repetitions / steps")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    want = set(a.only.split(","))
    reps = "6" if a.quick else "20"
    steps = ["--steps", "5", "--warmup", "2"] if a.quick else ["--steps", "20", "--warmup", "3"]
    out = []
    if "cfg1" in want:
        out.append(harness("cfg1", 100_000, 32, 10, 1_000, reps, 1000, a.out))
    if "cfg2" in want:
        out.append(harness("cfg2", 1_000_000, 64, 10, 10_000, reps, 2000, a.out))
    if "cfg3" in want:
        out.append(harness("cfg3", 10_000_000, 128, 10, 10_000, "6" if a.quick else "12", 1000, a.out))
        out.append(bench("cfg3", steps, a.out))
    if "cfg4" in want:
        out.append(bench("cfg4_one_gpu", ["--points", "40000000", "--queries", "100000", "--data", "randn", "--steps", "10", "--warmup",
                                          "2", "--cpu-seconds", "6"], a.out))
    if "cfg5" in want:
        out.append(bench("cfg5", ["--dtype", "f64", "--points", "10000000", "--dim", "256", "--knn", "100", "--queries", "10000",
                                  "--data", "randn", "--steps", "5", "--warmup", "1", "--cpu-seconds", "6"], a.out))
    with open(os.path.join(a.out, "configs.jsonl"), "w") as f:
        for line in out:
            s = json.dumps(line)
            print(s, flush=True)
            f.write(s + "\n")


if __name__ == "__main__":
    main()

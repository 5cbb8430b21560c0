#!/usr/bin/env python3
"""A/B of two builds of the engine on ONE box: MI355X devices differ by several percent in wall time under the same
VALU-dense kernels (and a profiled run clocks lower than an unprofiled one), so two bench.py lines are comparable only
when they come from the same device, back to back.  usage: ab_bench.py <other libitsx_hip.so> [bench.py arguments]
Runs bench.py with the in-tree library, then with the other one, alternating twice, and prints the reads/s and the
Forward / Backward kernel times of every run."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUN = ("import sys, runpy, os; sys.path.insert(0, %r); import itsxpress_amd._lib as l; "
       "l.LIB_PATH = sys.argv[1] if sys.argv[1] != '-' else l.LIB_PATH; sys.argv = ['bench.py'] + sys.argv[2:]; "
       "runpy.run_path(os.path.join(%r, 'bench.py'), run_name='__main__')" % (ROOT, ROOT))


def main():
    other = os.path.abspath(sys.argv[1])
    extra = sys.argv[2:] or ["--cpu-sample", "0"]
    for arm in ("in-tree", "other", "in-tree", "other"):
        out = subprocess.run([sys.executable, "-c", RUN, "-" if arm == "in-tree" else other] + extra, capture_output=True, text=True, cwd=ROOT)
        line = [x for x in out.stdout.split("\n") if x.startswith("{")]
        if not line:
            print(arm, "FAILED", out.stderr[-500:])
            continue
        d = json.loads(line[-1])
        print(arm, round(d["value"]), "reads/s  fwd %.1f ms  bwd %.1f ms  decode %.1f ms  env %.1f ms  bias %.1f ms  msv %.1f ms" % tuple(d["stage_ms"][k] for k in ("ms_fwd_kernel", "ms_bwd_kernel", "ms_decode_kernel", "ms_env_kernel", "ms_bias_kernel", "ms_msv_kernel")), flush=True)


if __name__ == "__main__":
    main()

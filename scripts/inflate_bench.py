#!/usr/bin/env python3
"""Reader timing: a gzip FASTQ of N synthetic 300-bp reads through itsx_io_read, serially (libdeflate / zlib) and with the
block-parallel inflater (pinflate.cpp), with and without huge pages.  usage: inflate_bench.py [reads]"""
import gzip
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time
sys.path.insert(0, %r)
from itsxpress_amd.trim import read_text
for i in range(3):
    t0 = time.perf_counter(); t = read_text(sys.argv[1]); print("%%.3f s  %%d bytes" %% (time.perf_counter() - t0, len(t)), flush=True)
''' % ROOT


def main():
    import numpy as np
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    rng = np.random.default_rng(5)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    tmpl = acgt[rng.integers(0, 4, (max(1, n // 50), 300))]
    reads = tmpl[rng.integers(0, len(tmpl), n)].copy()
    err = rng.random(reads.shape) < 0.003
    reads[err] = acgt[rng.integers(0, 4, int(err.sum()))]
    q = (np.clip(38 - (np.arange(300) // 25)[None, :] - rng.integers(0, 6, reads.shape), 2, 40) + 33).astype(np.uint8)
    tmp = tempfile.mkdtemp(prefix="itsx_inflate_")
    path = os.path.join(tmp, "in.fastq.gz")
    with gzip.open(path, "wb", compresslevel=6) as f:
        for i in range(n):
            f.write(b"@read%d 1:N:0:1\n" % i + reads[i].tobytes() + b"\n+\n" + q[i].tobytes() + b"\n")
    print("compressed MB", round(os.path.getsize(path) / 1e6, 1), flush=True)
    for name, env in (("serial", {"ITSX_PARALLEL_INFLATE": "0"}), ("parallel, 4-KB pages", {"ITSX_HUGEPAGES": "0"}), ("parallel, huge pages", {})):
        e = dict(os.environ, ITSX_TEXT_CACHE_GB="0", ITSX_TRACE_ALLOC="1", **env)
        out = subprocess.run([sys.executable, "-c", CHILD, path], env=e, capture_output=True, text=True)
        print("==", name)
        print(out.stdout.strip())
        print("\n".join(l for l in out.stderr.split("\n") if "parallel inflate" in l))
    os.remove(path)


if __name__ == "__main__":
    main()

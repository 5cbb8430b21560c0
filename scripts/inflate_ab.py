#!/usr/bin/env python3
"""A/B of the block-parallel inflater's switches on ONE box (run-to-run differences between boxes are larger than the effects):
writes N synthetic 2x250-like FASTQ records through the engine's own gzip writer, then inflates the file `reps` times per
setting with the text cache off and prints the phase times the library reports (ITSX_TRACE_ALLOC=1).
(Round 5: decoding up to three literals per 64-bit fetch instead of one changed nothing -- decode 290 vs 290-320 ms per 2.1 GB of text,
gpurun_out/inflate_ab.txt -- and was taken out again; the default settings below compare huge pages on / off.)
usage: python scripts/inflate_ab.py [--records 4000000] [--reps 3] [--switch ITSX_HUGEPAGES --values 1,0]"""
import argparse
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time
sys.path.insert(0, %r)
from itsxpress_amd import _lib
import ctypes as C
L = _lib.lib()
for _ in range(%d):
    L.itsx_io_cache_clear()
    t = C.c_char_p(); n = C.c_int64()
    t0 = time.perf_counter()
    rc = L.itsx_io_read(%r.encode(), C.byref(t), C.byref(n))
    dt = time.perf_counter() - t0
    assert rc == 0
    L.itsx_io_free(t)
    print("wall %%.3f s, %%.1f MB" %% (dt, n.value / 1e6), file=sys.stderr)
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=4000000)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--switch", default="ITSX_HUGEPAGES")
    ap.add_argument("--values", default="1,0")
    a = ap.parse_args()
    sys.path.insert(0, ROOT)
    import numpy as np
    from itsxpress_amd.trim import write_trimmed_fastq
    rng = np.random.default_rng(3)
    tmp = tempfile.mkdtemp(prefix="itsx_inflate_ab_")
    plain = os.path.join(tmp, "r.fastq")
    acgt = np.frombuffer(b"ACGT", np.uint8)
    qv = np.frombuffer(b"#-7AFF", np.uint8)
    with open(plain, "wb") as f:
        for i0 in range(0, a.records, 100000):
            m = min(100000, a.records - i0)
            sq = acgt[rng.integers(0, 4, (m, 250))]
            ql = qv[rng.integers(0, 6, (m, 250))]
            for i in range(m):
                f.write(b"@M0:1:000:1:%d:%d 1:N:0:1\n" % (1000 + (i0 + i) // 1000, (i0 + i) % 1000) + sq[i].tobytes() + b"\n+\n" + ql[i].tobytes() + b"\n")
    gz = plain + ".gz"
    write_trimmed_fastq(plain, gz, np.zeros(a.records, np.int32), np.full(a.records, 1 << 30, np.int32), gzipped=True)
    os.remove(plain)
    print("file: %.1f MB gz" % (os.path.getsize(gz) / 1e6))
    vals = a.values.split(",")
    for name, env in [("%s=%s%s" % (a.switch, v, " again" if k >= len(vals) else ""), {a.switch: v}) for k, v in enumerate(vals + vals)]:
        e = dict(os.environ, ITSX_TRACE_ALLOC="1", **env)
        r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, a.reps, gz)], env=e, capture_output=True, text=True)
        print("== " + name)
        for ln in r.stderr.split("\n"):
            if "parallel inflate" in ln or ln.startswith("wall"):
                print("   " + ln)
    os.remove(gz)
    os.rmdir(tmp)


if __name__ == "__main__":
    main()

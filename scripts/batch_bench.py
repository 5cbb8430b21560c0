"""Per-sample batching: S small samples through the path one by one (what the QIIME 2 plugin's loop does,
q2_itsxpress.py:273-333, here already on one shared context) against ONE batched pass (itsx_set_samples).
Checks that both give the same per-read coordinates.  Usage: python scripts/batch_bench.py [n_samples] [reads_per_sample]"""
import gzip
import json
import os
import sys
import time

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
sys.path.insert(0, ROOT + "/tests")
import synth  # noqa: E402
from bench import its2_profiles  # noqa: E402
from itsxpress_amd import Engine  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 96
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
thmm = gzip.open(ROOT + "/tests/golden/T.hmm.gz", "rt").read()
eng = Engine(0)
eng.load_profiles(text=its2_profiles(thmm))
parts = [synth.make_reads(thmm, n, config=2, seed=synth.SEED + 100 + s) for s in range(S)]


def one_by_one():
    out = []
    for blob, offs in parts:
        eng.set_reads_buffer(blob, offs)
        eng.derep()
        eng.search()
        eng.finalize()
        out.append([x.copy() for x in eng.trim_coords("3_", "4_")])
    return out


def batched():
    blob = b"".join(bytes(b) for b, _ in parts)
    offs = np.concatenate([[0]] + [o[1:] + k for (_, o), k in zip(parts, np.cumsum([0] + [int(o[-1]) for _, o in parts[:-1]]))]).astype(np.int64)
    smp = np.repeat(np.arange(S, dtype=np.int32), [len(o) - 1 for _, o in parts])
    eng.set_reads_buffer(blob, offs)
    eng.set_samples(smp, S)
    eng.derep()
    eng.search()
    eng.finalize()
    a = eng.trim_coords("3_", "4_")
    first = np.concatenate([[0], np.cumsum([len(o) - 1 for _, o in parts])])
    return [[x[first[i]:first[i + 1]] for x in a] for i in range(S)]


res = {}
for name, fn in (("one_by_one", one_by_one), ("batched", batched)):
    fn()                                   # warm-up: buffers grow to their working size
    t0 = time.perf_counter()
    out = fn()
    res[name] = time.perf_counter() - t0
    res[name + "_out"] = out
same = all(np.array_equal(a, b) for x, y in zip(res["one_by_one_out"], res["batched_out"]) for a, b in zip(x, y))
print(json.dumps({"samples": S, "reads_per_sample": n, "one_by_one_s": round(res["one_by_one"], 3), "batched_s": round(res["batched"], 3),
                  "reads_per_s_one_by_one": round(S * n / res["one_by_one"]), "reads_per_s_batched": round(S * n / res["batched"]),
                  "identical_coordinates": bool(same)}))

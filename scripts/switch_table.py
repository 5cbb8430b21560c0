#!/usr/bin/env python3
"""INTEGRATION.md section 7 from the library's own registry (csrc/switches.cpp via itsx_switch_registry) + the Python layer's switches:
one row per switch with its class, its meaning and the tests / scripts that set it.  usage: switch_table.py  (prints markdown)"""
import ctypes as C
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# the Python layer (itsxpress_amd/*.py): read with os.environ there, listed here with the same classes
PY_SWITCHES = [
    ("ITSXPRESS_GPU", "mode", "device ordinal of a one-GPU engine (default 0)"),
    ("ITSXPRESS_GPUS", "mode", "N > 1: one sample over N worker processes, one GPU each (multi.py); outputs byte-identical to one GPU's"),
    ("ITSXPRESS_GPU_IDS", "mode", "comma-separated device ordinals for ITSXPRESS_GPUS"),
    ("ITSXPRESS_ARRAYS", "mode", "=1: arrays instead of uc.txt / rep.fa / domtbl.txt between the stages, lazy rows mode (same trimmed reads)"),
    ("ITSXPRESS_DOMTBL", "mode", "=winners: files as ever, the lazy search behind them; domtbl.txt holds per target and side the row ItsPosition.parse ends up with (same dictionary, same trimmed reads)"),
    ("ITSXPRESS_STREAM", "mode", "=1 / 0: streamed file-order chunks on / off (default: by input size, ITSX_STREAM_AUTO_MB)"),
    ("ITSXPRESS_DB_DIR", "mode", "directory with additional ITSx_db HMM files (F.hmm)"),
    ("ITSXPRESS_XDIR", "tuning", "directory of the multi-GPU exchange files (default /dev/shm when it has room, else the temp directory)"),
    ("ITSX_STREAM_AUTO_MB", "tuning", "input size from which file mode streams"),
    ("ITSX_STREAM_CHUNK_MB", "tuning", "text per streamed chunk"),
    ("ITSX_STREAM_FINISHERS", "tuning", "threads finalizing streamed chunks"),
    ("ITSX_STREAM_SEARCHES", "tuning", "chunk searches in flight"),
    ("ITSX_STREAM_PAIR_THREADS", "tuning", "inflating threads per file of a streamed paired sample (default: the I/O pool's size each)"),
    ("ITSX_MULTI_LOAD", "tuning", "=pieces: the multi-GPU driver cuts the pieces after the whole file is inflated (round 5's load) instead of dealing them while it inflates"),
    ("ITSX_FORCE_DIST", "diagnostic", "=1: run the collectives even with one rank (tests/test_gpu_dist.py)"),
]


def registry():
    from itsxpress_amd import _lib
    L = _lib.lib()
    n = L.itsx_switch_registry(None, 0)
    b = C.create_string_buffer(int(n))
    L.itsx_switch_registry(b, n)
    return [tuple(line.split("\t")) for line in b.value.decode().strip().split("\n")]


def users(name):
    out = []
    for f in sorted(glob.glob(os.path.join(ROOT, "tests", "*.py")) + glob.glob(os.path.join(ROOT, "scripts", "*.py")) + [os.path.join(ROOT, "bench.py")]):
        if os.path.basename(f) == "switch_table.py":
            continue
        if re.search(r"\b%s\b" % name, open(f).read()):
            out.append(os.path.relpath(f, ROOT))
    return out


def table():
    rows = [(n, k, w, "library") for n, k, w in registry()] + [(n, k, w, "Python layer") for n, k, w in PY_SWITCHES]
    order = {"hook": 0, "mode": 1, "diagnostic": 2, "tuning": 3}
    rows.sort(key=lambda r: (order[r[1]], r[3], r[0]))
    lines = ["| switch | class | read by | meaning | set by (tests / scripts) |", "|---|---|---|---|---|"]
    for n, k, w, where in rows:
        u = users(n)
        lines.append("| `%s` | %s | %s | %s | %s |" % (n, k, where, w, ", ".join("`%s`" % x for x in u[:4]) + (" ..." if len(u) > 4 else "") if u else "--"))
    return "\n".join(lines)


if __name__ == "__main__":
    print(table())

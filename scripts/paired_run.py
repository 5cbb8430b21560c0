#!/usr/bin/env python3
"""Paired-end, file to file, on one GPU -- the reference's main use (`itsxpress --fastq R1.fq.gz --fastq2 R2.fq.gz
--region ITS2 --taxa ... --outfile o1.fq.gz --outfile2 o2.fq.gz`, main.py:534-606) through the mirror classes:
merge (f2) -> dereplicate (a1) -> search (a4) -> per-read coordinates (a5-a7) -> paired trimmed output (f1).
N synthetic ITS2 amplicons of 300-480 bases are sequenced as 2x250 with Illumina-like qualities and errors.
Prints one JSON line with the stage times.  usage: paired_run.py [--pairs 500000] [--keep-files]"""
import argparse
import gzip
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=500000)
    ap.add_argument("--array-path", action="store_true", help="skip uc.txt / rep.fa / domtbl.txt (the in-memory hand-off, f3)")
    ap.add_argument("--stream", action="store_true", help="round 6: after the staged run, the same files through the STREAMED paired pipeline "
                                                          "(ITSXPRESS_STREAM=1 ITSXPRESS_ARRAYS=1, SeqSample.plan_output_paired) and its wall time")
    ap.add_argument("--check", action="store_true", help="--stream: the two outputs (inflated) must equal the staged run's byte for byte")
    args = ap.parse_args()
    print(json.dumps(run(args.pairs, args.array_path, args.stream, args.check)))


def stream_leg(paths, hmm, tmp):
    """the reference's call sequence (main.py:513-519, 534-554, 556-624) through the mirror with the streaming engine: R1 / R2 inflated side
    by side, merged chunk by chunk on the device, scored, the two outputs deflated while later chunks are scored"""
    import importlib
    S = importlib.import_module("itsxpress_amd.SeqSample")
    from itsxpress_amd import trim
    keep = {k: os.environ.get(k) for k in ("ITSXPRESS_ARRAYS", "ITSXPRESS_STREAM", "ITSXPRESS_GPUS")}
    os.environ.update({"ITSXPRESS_ARRAYS": "1", "ITSXPRESS_STREAM": "1", "ITSXPRESS_GPUS": "1"})
    trim.cache_clear()
    o1, o2 = os.path.join(tmp, "s1.fastq.gz"), os.path.join(tmp, "s2.fastq.gz")
    sobj = None
    try:
        t0 = time.perf_counter()
        sobj = S.SeqSamplePairedNotInterleaved(fastq=paths[0], tempdir=os.path.join(tmp, "swork"), fastq2=paths[1])
        sobj.plan_output_paired(o1, o2, "ITS2", gzipped=True)
        sobj._merge_reads(threads=1, stagger=False)
        sobj.deduplicate(threads=1)
        sobj._search(hmmfile=hmm, threads=1)
        t_pipe = time.perf_counter() - t0
        its_pos = S.ItsPosition(domtable=sobj.dom_file, region="ITS2")
        dd = S.Dedup(uc_file=sobj.uc_file, rep_file=sobj.rep_file, seq_file=sobj.seq_file, fastq=sobj.r1, fastq2=sobj.fastq2)
        dd.create_paired_trimmed_seqs(o1, o2, gzipped=True, zstd_file=False, itspos=its_pos, wri_file=True)
        total = time.perf_counter() - t0
        eng = sobj._engine
        return {"s_total": round(total, 3), "s_merge+derep+search (streamed)": round(t_pipe, 3), "s_finalize+write tail": round(total - t_pipe, 3),
                "chunks": int(eng.world), "pairs": int(getattr(eng, "n_pairs", 0)), "merged": int(eng.n_reads),
                "timeline_s(chunk, text ready, loaded, searched)": [list(x) for x in eng.timeline],
                "output_gz_MB": round((os.path.getsize(o1) + os.path.getsize(o2)) / 1e6, 1)}, (o1, o2)
    finally:
        if sobj is not None and getattr(sobj, "_engine", None) is not None:
            sobj._engine.close()
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def run(pairs, array_path=True, stream=False, check=False):
    """the run as a function (bench.py --paired-pairs calls it for its `paired_file_to_file` key): the stage times as a dict"""
    args = argparse.Namespace(pairs=int(pairs), array_path=bool(array_path))
    import synth
    from bench import its2_profiles
    from itsxpress_amd import Engine, SeqSamplePairedNotInterleaved
    from itsxpress_amd.trim import read_text, write_trimmed_fastq, write_trimmed_paired
    with gzip.open(os.path.join(ROOT, "tests", "golden", "T.hmm.gz"), "rt") as f:
        thmm = f.read()
    n = args.pairs
    rng = np.random.default_rng(17)
    t_gen = time.perf_counter()

    def note(what):                      # (a 10 M-pair input takes minutes to generate: a line per stage keeps a watched run alive)
        print("[paired_run] %6.1f s  %s" % (time.perf_counter() - t_gen, what), file=sys.stderr, flush=True)
    # amplicons: a library of templates of 300-480 bases, Zipf-sampled; the differences between reads come from sequencing
    nt = max(1, n // 50)
    tb, to = synth.make_reads(thmm, nt, config=3, seed=synth.SEED + 7, fixed_len=0, len_range=(300, 480), frac_templates=1.0,
                              sub_rate=0.0, n_rate=0.0, rc_rate=0.0)
    tb = np.frombuffer(tb, np.uint8)
    w = 1.0 / np.arange(1, nt + 1) ** 1.1
    ids = rng.choice(nt, size=n, p=w / w.sum())
    comp = np.zeros(256, np.uint8)
    for a, b in zip(b"ACGTN", b"TGCAN"):
        comp[a] = b
    tmat = np.full((nt, 480), ord("A"), np.uint8)
    tlen = np.diff(to)
    for t in range(nt):
        tmat[t, :tlen[t]] = tb[to[t]:to[t + 1]]
    L = tlen[ids]
    fwd = tmat[ids, :250]
    cols = (L[:, None] - 1 - np.arange(250)[None, :])
    rev = comp[tmat[ids[:, None], cols]]
    acgt = np.frombuffer(b"ACGT", np.uint8)
    qv = np.array([2, 12, 22, 30, 37, 38], np.uint8)
    qp = [.001, .009, .03, .08, .28, .6]

    def sequenced(reads):
        q = rng.choice(qv, size=reads.shape, p=qp)
        err = rng.random(reads.shape) < 10.0 ** (-q.astype(np.float64) / 10.0)
        reads = reads.copy()
        reads[err] = acgt[rng.integers(0, 4, int(err.sum()))]
        return reads, (q + 33).astype(np.uint8)
    note("templates sampled")
    fwd, fq = sequenced(fwd)
    note("R1 sequenced")
    rev, rq = sequenced(rev)
    note("R2 sequenced")
    tmp = tempfile.mkdtemp(prefix="itsx_paired_run_")
    try:
        paths = []
        for tag, (sq, ql) in (("R1", (fwd, fq)), ("R2", (rev, rq))):
            plain = os.path.join(tmp, tag + ".fastq")
            with open(plain, "wb") as f:
                for i in range(n):
                    f.write(b"@M0:1:000:1:%d:%d %s:N:0:1\n" % (1000 + i // 1000, i % 1000, tag[1:].encode()) + sq[i].tobytes() + b"\n+\n" + ql[i].tobytes() + b"\n")
            gz = plain + ".gz"
            write_trimmed_fastq(plain, gz, np.zeros(n, np.int32), np.full(n, 1 << 30, np.int32), gzipped=True)
            os.remove(plain)
            paths.append(gz)
            note(tag + " written")
        hmm = os.path.join(tmp, "its2.hmm")
        with open(hmm, "w") as f:
            f.write(its2_profiles(thmm))
        eng = Engine(0)
        eng.load_profiles(path=hmm)
        eng.set_reads([tmat[t, :tlen[t]].tobytes().decode() for t in range(min(nt, 2000))])      # first-touch costs outside the stages
        eng.derep()
        eng.search()
        eng.finalize()
        s = SeqSamplePairedNotInterleaved(paths[0], os.path.join(tmp, "work"), paths[1])
        s._engine = eng
        t = {}
        note("staged run starts")
        t0 = time.perf_counter()
        s._merge_reads(threads=1, stagger=False)
        t["merge"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        if args.array_path:
            if getattr(s, "_reads_loaded_from", None) != s.seq_file:      # (ITSXPRESS_ARRAYS=1: the merge left its reads in the engine)
                eng.load_reads_file(s.seq_file)
            nu = eng.derep()
            t["derep"] = time.perf_counter() - t0
            t0 = time.perf_counter()
            eng.set_rows_mode("lazy")              # what SeqSample._search selects in arrays mode (coordinates only, no domtbl.txt)
            eng.search()
            eng.finalize()
        else:
            s.deduplicate(threads=1)
            nu = eng.n_unique
            t["derep"] = time.perf_counter() - t0
            t0 = time.perf_counter()
            s._search(hmmfile=hmm, threads=1)
        t["search"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        start, stop, tl, ind = s.trim_coordinates("ITS2")
        names = eng.read_names_raw()
        t["coords+names"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        o1, o2 = os.path.join(tmp, "o1.fastq.gz"), os.path.join(tmp, "o2.fastq.gz")
        nw = write_trimmed_paired(paths[0], paths[1], o1, o2, names, start, stop, tl, gzipped=True)
        t["write"] = time.perf_counter() - t0
        total = sum(t.values())
        kept = int(((start >= 0) & (stop >= 0) & (start < stop)).sum())
        assert nw == kept, (nw, kept)
        streamed = None
        note("staged run done: %.2f s" % total)
        if stream:
            eng.close()
            # (a leg that does not end is reported by every thread's stack after two minutes)
            import faulthandler
            faulthandler.dump_traceback_later(120, exit=True)
            streamed, (s1, s2) = stream_leg(paths, hmm, tmp)
            faulthandler.cancel_dump_traceback_later()
            note("streamed run done: %.2f s" % streamed["s_total"])
            streamed["pairs_per_s_file_to_file"] = round(n / streamed["s_total"])
            if check:
                for a, b in ((o1, s1), (o2, s2)):
                    with gzip.open(a, "rb") as fa, gzip.open(b, "rb") as fb:
                        while True:
                            x, y = fa.read(1 << 24), fb.read(1 << 24)
                            assert x == y, "streamed output differs from the staged one: %s" % b
                            if not x:
                                break
                streamed["outputs_equal_staged"] = True
        return ({"streamed": streamed, "pairs": n, "merged": len(names[1]) - 1, "unique": int(nu), "pairs_written": int(nw), "array_path": bool(args.array_path),
                          "input_gz_MB": round(sum(os.path.getsize(p) for p in paths) / 1e6, 1),
                          "output_gz_MB": round((os.path.getsize(o1) + os.path.getsize(o2)) / 1e6, 1),
                          "gzip_level": int(os.environ.get("ITSX_GZIP_LEVEL", 6)),
                          **{"s_" + k: round(v, 3) for k, v in t.items()}, "s_total": round(total, 3),
                          "pairs_per_s_file_to_file": round(n / total)})
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()

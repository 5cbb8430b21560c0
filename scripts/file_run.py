#!/usr/bin/env python3
"""File to file on one GPU: a gzip FASTQ of N synthetic 300-bp reads in, a compressed FASTQ of the trimmed reads out
-- what a user of `itsxpress --fastq x.fq.gz --single_end --region ITS2 --outfile y.fq.gz` waits for, stage by stage
(load = inflate + parse + upload + device packing; write = slice + block-parallel deflate; the input's text is shared
between the two through the reader's cache, ITSX_TEXT_CACHE_GB=0 turns that off).
Prints one JSON line.  usage: file_run.py [--reads 1000000] [--out-kind gz|zst|plain] [--stream [--check]]"""
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
    ap.add_argument("--reads", type=int, default=1000000)
    ap.add_argument("--out-kind", default="gz", choices=["gz", "zst", "plain"])
    ap.add_argument("--rows", default="lazy", choices=["lazy", "compact", "full"], help="the search's rows mode (lazy = what ITSXPRESS_ARRAYS=1 selects)")
    ap.add_argument("--shape", default="cfg1", choices=["cfg1", "cfg2"], help="cfg1: 300-base reads; cfg2: merged reads of 300-580 bases")
    ap.add_argument("--stream", action="store_true", help="file-order chunks scored while the file is inflated (itsxpress_amd/stream.py)")
    ap.add_argument("--chunk-mb", type=float, default=0.0, help="--stream: text per chunk (0: about a tenth of the file)")
    ap.add_argument("--stream-write", action="store_true", help="--stream: the writer inside the pipeline too (provisional thresholds per chunk)")
    ap.add_argument("--check", action="store_true", help="--stream: compare the coordinates with one context on the whole file")
    ap.add_argument("--input-dir", default="", help="keep the generated in.fastq.gz here and use it again when it is there (several arms over one file)")
    args = ap.parse_args()
    import synth
    from bench import its2_profiles
    from itsxpress_amd import Engine, _lib
    from itsxpress_amd.trim import write_trimmed_fastq, read_text
    with gzip.open(os.path.join(ROOT, "tests", "golden", "T.hmm.gz"), "rt") as f:
        thmm = f.read()
    if args.shape == "cfg2":
        blob, offs = synth.make_reads(thmm, args.reads, config=3, seed=synth.SEED + 3, fixed_len=0, len_range=(300, 580))
    else:
        blob, offs = synth.make_reads(thmm, args.reads, config=2, seed=synth.SEED + 2)
    n = args.reads
    rng = np.random.default_rng(9)
    tmp = tempfile.mkdtemp(prefix="itsx_file_run_")
    try:
        plain = os.path.join(tmp, "in.fastq")
        bases = np.frombuffer(blob, np.uint8)
        kept = os.path.join(args.input_dir, "in_%s_%d.fastq.gz" % (args.shape, n)) if args.input_dir else ""
        if kept and os.path.exists(kept) and os.path.exists(kept + ".size"):
            fq = kept
            in_bytes, in_gz = int(open(kept + ".size").read()), os.path.getsize(fq)
        else:
            with open(plain, "wb") as f:                  # Illumina-like qualities: high, decaying along the read
                qtab = [(np.clip(38 - (np.arange(600) // 25) - rng.integers(0, 6, 600), 2, 40) + 33).astype(np.uint8).tobytes() for _ in range(64)]
                for i in range(n):
                    s = bases[offs[i]:offs[i + 1]]
                    f.write(b"@read%d 1:N:0:1\n" % i + s.tobytes() + b"\n+\n" + qtab[i & 63][:len(s)] + b"\n")
            fq = kept or os.path.join(tmp, "in.fastq.gz")
            if kept:
                os.makedirs(args.input_dir, exist_ok=True)
            write_trimmed_fastq(plain, fq, np.zeros(n, np.int32), np.full(n, 1 << 30, np.int32), gzipped=True)
            in_bytes, in_gz = os.path.getsize(plain), os.path.getsize(fq)
            os.remove(plain)
            if kept:
                open(kept + ".size", "w").write(str(in_bytes))

        eng = Engine(0)
        eng.set_rows_mode(args.rows)
        eng.load_profiles(text=its2_profiles(thmm))
        # first-touch costs (context, code objects, allocator) outside the stages: one small pass of the whole path
        eng.set_reads([bases[offs[i]:offs[i + 1]].tobytes().decode() for i in range(2000)], ["w%d" % i for i in range(2000)])
        eng.derep()
        eng.search()
        eng.finalize()
        t = {}
        extra = {}
        if args.stream:
            # file-order chunks: chunk k is dereplicated and scored while chunk k + 1 is inflated and parsed (itsxpress_amd/stream.py)
            from itsxpress_amd.stream import StreamEngine
            eng.close()
            t0 = time.perf_counter()
            se = StreamEngine(0, chunk_mb=args.chunk_mb or None)
            se.set_rows_mode(args.rows)
            out = os.path.join(tmp, "trimmed.fastq" + {"gz": ".gz", "zst": ".zst", "plain": ""}[args.out_kind])
            if args.stream_write:
                se.plan_output(out, "3_", "4_", gzipped=args.out_kind == "gz", zstd_file=args.out_kind == "zst")
            se.load_reads_file(fq)
            se.derep()
            se.load_profiles(text=its2_profiles(thmm))
            se.search()
            t["load+search (streamed)"] = time.perf_counter() - t0
            t0 = time.perf_counter()
            se.finalize()
            t["finalize"] = time.perf_counter() - t0
            nu = se.n_unique
            if args.stream_write and not args.check:    # (the writer has had its rows chunk by chunk: nobody needs the per-read arrays)
                start = stop = None
            else:
                t0 = time.perf_counter()
                start, stop, tlen, ind = se.trim_coords("3_", "4_")
                t["coords"] = time.perf_counter() - t0
            if args.stream_write:
                t0 = time.perf_counter()
                sw = se.finish_output()
                t["write (the part left after the pipeline)"] = time.perf_counter() - t0
                extra_w = {"late_uniques": se._out.n_late_uniques}
            extra = {"stream_chunks": se.world, "stream_timeline_s(chunk, text ready, loaded, searched)": se.timeline,
                     "chunk_load_s": [st.get("load_s") for _, st in se._engs], "finalize_s": getattr(se, "finalize_s", None)}
            if args.check:                              # the same file through one context: identical coordinates
                e1 = Engine(0)
                e1.set_rows_mode(args.rows)
                e1.load_profiles(text=its2_profiles(thmm))
                e1.load_reads_file(fq); e1.derep(); e1.search(); e1.finalize()
                ref = e1.trim_coords("3_", "4_")
                extra["coordinates_equal_one_context"] = bool(all(np.array_equal(a, b) for a, b in zip((start, stop, tlen, ind), ref)))
                assert extra["coordinates_equal_one_context"]
                e1.close()
        else:
            t0 = time.perf_counter()
            eng.load_reads_file(fq)
            t["load"] = time.perf_counter() - t0
            t0 = time.perf_counter()
            nu = eng.derep()
            eng.search()
            eng.finalize()
            start, stop, tlen, ind = eng.trim_coords("3_", "4_")
            t["path"] = time.perf_counter() - t0
        out = os.path.join(tmp, "trimmed.fastq" + {"gz": ".gz", "zst": ".zst", "plain": ""}[args.out_kind])
        if args.stream and args.stream_write:
            nw, tot = sw
            extra["representatives_that_waited_for_the_exact_thresholds"] = extra_w["late_uniques"]
            if args.check:                              # the one-go writer on the same coordinates: the same bytes
                ref_out = out + ".ref"
                assert write_trimmed_fastq(fq, ref_out, start, stop, gzipped=args.out_kind == "gz", zstd_file=args.out_kind == "zst") == (nw, tot)
                extra["output_bytes_equal_one_go_writer"] = open(ref_out, "rb").read() == open(out, "rb").read()
                assert extra["output_bytes_equal_one_go_writer"]
        else:
            t0 = time.perf_counter()
            nw, tot = write_trimmed_fastq(fq, out, start, stop, gzipped=args.out_kind == "gz", zstd_file=args.out_kind == "zst")
            t["write"] = time.perf_counter() - t0
        total = sum(t.values())
        # the output holds exactly the kept reads, sliced: check a sample against the coordinates
        if start is None:
            start, stop, _, _ = se.trim_coords("3_", "4_")          # (after the clock stopped: for the checks below)
        kept = np.flatnonzero((start >= 0) & (stop >= 0) & (start < stop))
        assert nw == len(kept)
        big = os.path.getsize(out) > (1 << 30) or tot > (1 << 30)      # (ctypes.string_at takes a C int: the whole text of a 10 M-read run does not fit)
        lines = [] if big else read_text(out).split(b"\n")
        assert big or len(lines) == 4 * nw + 1
        for k in (() if big else (0, nw // 2, nw - 1)):
            i = int(kept[k])
            assert lines[4 * k] == b"@read%d 1:N:0:1" % i
            assert lines[4 * k + 1] == bases[offs[i]:offs[i + 1]].tobytes()[start[i]:stop[i]]
        print(json.dumps({
            "reads": n, "shape": args.shape, "rows": args.rows, "unique": int(nu), "written": int(nw), "out_kind": args.out_kind,
            "input_MB": round(in_bytes / 1e6, 1), "input_gz_MB": round(in_gz / 1e6, 1), "output_MB": round(os.path.getsize(out) / 1e6, 1),
            "stages_s": {k: round(v, 3) for k, v in t.items()}, "s_total": round(total, 3), **extra,
            "reads_per_s_file_to_file": round(n / total), "io_threads": int(os.environ.get("ITSX_IO_THREADS", 0)) or min(os.cpu_count(), 32),
            "codecs": _lib.lib().itsx_io_codecs()}))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()

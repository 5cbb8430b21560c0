"""GPU parity tests for SURVEY 8f row f4 (read orientation, vsearch --orient restated): engine == oracle on strands and
both 12-mer counts; the oriented FASTQ; the SeqSample.orient_reads mirror with the reference's own database file."""
import gzip
import os

import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu
_RC = str.maketrans("ACGTNRYKMBVDH", "TGCANYRMKVBHD")


def _fasta(path):
    seqs, cur = [], []
    with gzip.open(path, "rt") as f:
        for line in f:
            if line.startswith(">"):
                if cur:
                    seqs.append("".join(cur))
                cur = []
            else:
                cur.append(line.strip().upper())
    if cur:
        seqs.append("".join(cur))
    return seqs


def test_orient_matches_oracle(engine, gold, tmp_path):
    db = _fasta(os.path.join(gold, "universal_orient_ref_clean.fasta.gz"))
    assert engine.orient_load_db(os.path.join(gold, "universal_orient_ref_clean.fasta.gz")) == len(db) == 599
    rng = np.random.default_rng(4)
    reads = []
    for i in range(400):
        src = db[int(rng.integers(0, len(db)))]
        a = int(rng.integers(0, max(1, len(src) - 300)))
        s = list(src[a:a + int(rng.integers(40, 900))])
        for _ in range(int(rng.integers(0, 12))):
            s[int(rng.integers(0, len(s)))] = str(rng.choice(list("ACGTN")))
        s = "".join(s)
        kind = i % 5
        if kind == 1:
            s = s[::-1].translate(_RC)
        elif kind == 2:
            s = "".join(rng.choice(list("ACGT"), len(s)))                       # unrelated
        elif kind == 3:
            s = s[:len(s) // 2] + s[len(s) // 2:][::-1].translate(_RC)            # half and half
        reads.append(s)
    reads += ["ACGT", "N" * 50, db[0][:11], db[0][:12]]
    engine.set_reads(reads)
    strand, cf, cr = engine.orient()
    es, ef, er = orc.orient(db, reads)
    assert np.array_equal(strand, es) and np.array_equal(cf, ef) and np.array_equal(cr, er)
    assert (strand == 1).sum() > 100 and (strand == -1).sum() > 60 and (strand == 0).sum() > 60
    # the FASTQ vsearch --orient --fastqout would hold
    from itsxpress_amd.trim import write_oriented_fastq
    fq = tmp_path / "in.fq"
    with open(fq, "w") as f:
        for i, s in enumerate(reads):
            f.write("@r%d extra\n%s\n+\n%s\n" % (i, s, "".join(chr(33 + (k * 7 + i) % 40) for k in range(len(s)))))
    out = tmp_path / "oriented.fq"
    n = write_oriented_fastq(str(fq), str(out), strand)
    assert n == int((strand != 0).sum())
    lines = open(out).read().split("\n")
    k = 0
    for i, s in enumerate(reads):
        if strand[i] == 0:
            continue
        q = "".join(chr(33 + (j * 7 + i) % 40) for j in range(len(s)))
        assert lines[4 * k] == "@r%d extra" % i
        assert lines[4 * k + 1] == (s if strand[i] > 0 else s[::-1].translate(_RC))
        assert lines[4 * k + 3] == (q if strand[i] > 0 else q[::-1])
        k += 1


def test_orient_reads_mirror(gold, tmp_path, monkeypatch):
    """the fixture's merged ITS2 amplicons against the reference's own orientation database: a decided strand for
    almost every read, and the sample points at oriented.fq afterwards (SeqSample.py:83-86)"""
    monkeypatch.setenv("ITSXPRESS_DB_DIR", gold)
    import importlib
    import itsxpress_amd.definitions as D
    importlib.reload(D)
    from itsxpress_amd.SeqSample import SeqSampleNotPaired
    s = SeqSampleNotPaired(fastq=os.path.join(gold, "seq.fq.gz"), tempdir=str(tmp_path))
    s.orient_reads(threads=1)
    assert s.fastq == s.seq_file == s.r1 == str(tmp_path / "oriented.fq")
    n = sum(1 for i, _ in enumerate(open(s.seq_file)) if i % 4 == 0)
    assert 200 <= n <= 227

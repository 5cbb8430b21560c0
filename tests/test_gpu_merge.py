"""GPU parity tests for SURVEY 8f row f2 (paired-end merge): the HIP kernel through the C ABI against oracle/orc_merge.c --
same accept/reject reason, same merged bases and qualities, same score bits and diagonal for every pair."""
import gzip
import os

import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu
_COMP = str.maketrans("ACGTN", "TGCAN")


def _read_fastq(path):
    out = []
    with gzip.open(path, "rt") as f:
        while True:
            h = f.readline()
            if not h:
                break
            s = f.readline().strip()
            f.readline()
            q = f.readline().strip()
            out.append((h[1:].split()[0], s, q))
    return out


def _check(engine, fwd, fq, rev, rq, **kw):
    reason, merged, score, shift = engine.merge_pairs(fwd, fq, rev, rq, **kw)
    n_ok = 0
    for i in range(len(fwd)):
        o = orc.merge_pair(fwd[i], fq[i], rev[i], rq[i], maxdiffs=kw.get("maxdiffs", 40), maxee=kw.get("maxee", 2.0),
                           allow_stagger=kw.get("allow_stagger", False))
        assert orc.MERGE_REASONS[int(reason[i])] == o[0], (i, int(reason[i]), o[0])
        assert np.float64(score[i]).view(np.uint64) == np.float64(o[3]).view(np.uint64) and int(shift[i]) == o[4], i
        if o[0] == "ok":
            assert merged[i] == (o[1], o[2]), i
            n_ok += 1
        else:
            assert merged[i] is None
    return n_ok


def test_merge_fixture_pairs(engine, gold):
    r1 = _read_fastq(os.path.join(gold, "4774-1-MSITS3_R1.fastq.gz"))
    r2 = _read_fastq(os.path.join(gold, "4774-1-MSITS3_R2.fastq.gz"))
    args = ([x[1] for x in r1], [x[2] for x in r1], [x[1] for x in r2], [x[2] for x in r2])
    assert _check(engine, *args) == 236
    assert _check(engine, *args, allow_stagger=True) >= 236
    assert _check(engine, *args, maxdiffs=2, maxee=0.5) < 200


def test_merge_synthetic_pairs(engine):
    rng = np.random.default_rng(8)
    acgt = np.array(list("ACGT"))
    fwd, fq, rev, rq = [], [], [], []
    for i in range(600):
        L = int(rng.integers(120, 520))
        frag = "".join(acgt[rng.integers(0, 4, L)])
        if i % 17 == 0:
            frag = ("ACGGTCATTG" * 60)[:L]                      # tandem repeat: several alignments
        fl, rl = int(rng.integers(80, 300)), int(rng.integers(80, 300))
        f = frag[:fl]
        r = frag[max(0, L - rl):][::-1].translate(_COMP)
        if i % 11 == 0:                                         # staggered: the reverse read starts before the forward read
            f = frag[L // 3:][:fl]

        def noisy(s, rate):
            s = list(s)
            q = []
            for k in range(len(s)):
                qq = int(rng.choice([2, 8, 14, 20, 30, 38, 40], p=[.02, .05, .08, .1, .15, .5, .1]))
                if rng.random() < 10 ** (-qq / 10.0) * rate:
                    s[k] = str(acgt[rng.integers(0, 4)])
                if rng.random() < 0.004:
                    s[k] = "N"
                    qq = 0
                q.append(chr(33 + qq))
            return "".join(s), "".join(q)
        f, q1 = noisy(f, 1.0)
        r, q2 = noisy(r, 1.5)
        fwd.append(f); fq.append(q1); rev.append(r); rq.append(q2)
    n1 = _check(engine, fwd, fq, rev, rq)
    n2 = _check(engine, fwd, fq, rev, rq, allow_stagger=True)
    assert 50 < n1 < n2
    # a batch holding a long pair takes the kernel variant without the 5-mer index (LDS): same answers
    frag = "".join(acgt[rng.integers(0, 4, 4200)])
    lf, lr = frag[:2600], frag[1800:][::-1].translate(_COMP)
    assert _check(engine, fwd[:40] + [lf], fq[:40] + ["I" * len(lf)], rev[:40] + [lr], rq[:40] + ["I" * len(lr)], maxee=50.0) >= 1
    # empty input, one-base reads
    assert _check(engine, ["A"], ["I"], ["T"], ["I"]) == 0
    reason, merged, _, _ = engine.merge_pairs([], [], [], [])
    assert len(reason) == 0 and merged == []


def test_merge_files_and_mirror(engine, gold, tmp_path, mini_hmm_text):
    """SeqSamplePairedNotInterleaved._merge_reads -> seq.fq -> deduplicate -> _search, all on the engine"""
    from itsxpress_amd.SeqSample import SeqSamplePairedNotInterleaved
    r1, r2 = os.path.join(gold, "4774-1-MSITS3_R1.fastq.gz"), os.path.join(gold, "4774-1-MSITS3_R2.fastq.gz")
    s = SeqSamplePairedNotInterleaved(fastq=r1, tempdir=str(tmp_path), fastq2=r2)
    s._merge_reads(threads=1, stagger=False)
    assert s.seq_file == str(tmp_path / "seq.fq")
    recs = open(s.seq_file).read().split("\n")
    heads = [recs[i][1:] for i in range(0, len(recs) - 1, 4)]
    a, b = _read_fastq(r1), _read_fastq(r2)
    exp = []
    for (h, f, fq), (_, r, rq) in zip(a, b):
        o = orc.merge_pair(f, fq, r, rq)
        if o[0] == "ok":
            exp += ["@" + h, o[1], "+", o[2]]
    assert recs[:-1] == exp and len(heads) == 236
    hmm = tmp_path / "mini.hmm"
    hmm.write_text(mini_hmm_text)
    s.deduplicate(threads=1)
    s._search(hmmfile=str(hmm), threads=1)
    assert os.path.getsize(s.uc_file) > 0 and os.path.exists(s.dom_file)
    # the engine's read labels are the merged records' identifiers; the paired writer takes them as (blob, offsets) as well as a list
    from itsxpress_amd.trim import write_trimmed_paired
    assert s.engine.read_names() == heads
    start, stop, tlen, _ = s.trim_coordinates("ITS2")
    outs = []
    for k, names in enumerate((heads, s.engine.read_names_raw())):
        o1, o2 = str(tmp_path / ("o1_%d.fq" % k)), str(tmp_path / ("o2_%d.fq" % k))
        n = write_trimmed_paired(r1, r2, o1, o2, names, start, stop, tlen)
        outs.append((n, open(o1, "rb").read(), open(o2, "rb").read()))
    assert outs[0] == outs[1] and outs[0][0] == int(((start >= 0) & (stop >= 0) & (start < stop)).sum())
    with pytest.raises(FileNotFoundError):
        SeqSamplePairedNotInterleaved(fastq=r1, tempdir=str(tmp_path), fastq2=str(tmp_path / "nope.fq"))._merge_reads(threads=1)


def test_merge_into_the_engines_read_set(engine, gold, tmp_path, t_hmm_text, monkeypatch):
    """arrays mode: _merge_reads leaves the merged reads as the engine's read set (itsx_merge_pairs_load: gathered and packed on the
    device, no seq.fq) -- the same reads, labels, dereplication, coordinates and trimmed pairs as through the file"""
    from itsxpress_amd.SeqSample import SeqSamplePairedNotInterleaved, Dedup, ItsPosition
    r1, r2 = os.path.join(gold, "4774-1-MSITS3_R1.fastq.gz"), os.path.join(gold, "4774-1-MSITS3_R2.fastq.gz")
    from bench import its2_profiles
    hmm = tmp_path / "its2.hmm"
    hmm.write_text(its2_profiles(t_hmm_text))
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("ITSXPRESS_ARRAYS", mode)
        monkeypatch.setenv("ITSXPRESS_STREAM", "0")
        d = tmp_path / ("arrays" + mode)
        s = SeqSamplePairedNotInterleaved(fastq=r1, tempdir=str(d), fastq2=r2)
        s._merge_reads(threads=1, stagger=False)
        assert os.path.exists(d / "seq.fq") == (mode == "0")                 # nothing written in arrays mode
        s.deduplicate(threads=1)
        s._search(hmmfile=str(hmm), threads=1)
        pos = ItsPosition(domtable=s.dom_file, region="ITS2")
        dd = Dedup(uc_file=s.uc_file, rep_file=s.rep_file, seq_file=s.seq_file, fastq=s.r1, fastq2=s.fastq2)
        o1, o2 = str(d / "o1.fq"), str(d / "o2.fq")
        dd.create_paired_trimmed_seqs(o1, o2, gzipped=False, zstd_file=False, itspos=pos, wri_file=True)
        # the reference's other route for a paired sample (main.py:596-624: --fastq2 without --outfile2): the MERGED reads trimmed, one
        # file -- in arrays mode seq.fq was never written, the merged records are materialised when this consumer asks for them
        om = str(d / "merged_trimmed.fq")
        dd.create_trimmed_seqs(om, gzipped=False, zstd_file=False, itspos=pos, wri_file=True, tempdir=str(d))
        assert os.path.exists(d / "seq.fq")
        rep_of, strand, uniq_of = s.engine.get_derep()
        res[mode] = (s.engine.read_names(), rep_of.copy(), strand.copy(), [np.asarray(c).copy() for c in s.trim_coordinates("ITS2")],
                     open(o1, "rb").read(), open(o2, "rb").read(), open(om, "rb").read(), open(d / "seq.fq", "rb").read())
        s._engine.close()
    a, b = res["0"], res["1"]
    assert a[0] == b[0] and len(a[0]) == 236
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert all(np.array_equal(x, y) for x, y in zip(a[3], b[3]))
    assert a[4] == b[4] and a[5] == b[5] and len(a[4]) > 1000
    assert a[6] == b[6] and len(a[6]) > 1000 and a[7] == b[7]


"""Per-sample batching (q2_itsxpress.py:273-333 runs the path once per sample): many samples share one
read set and one pass of every kernel, and every sample must still get exactly the results of a run of its own --
checked against the CPU oracle run per sample, against the engine run per sample, and on the written files byte for
byte.  `pytest -m gpu`."""
import os

import numpy as np
import pytest

import orc
import synth
from test_gpu_parity import _bits, _its2_subset

pytestmark = pytest.mark.gpu

_RC = str.maketrans("ACGTNRYKMSWBDHV", "TGCANYRMKSWVHDB")


def _rc(s):
    return s.translate(_RC)[::-1]


def _samples(t_hmm_text):
    """Six samples of very different size that share sequences (also across strands), incl. an empty one and one read."""
    blob, offs = synth.make_reads(t_hmm_text, 900, seed=31)
    pool = synth.to_strings(blob, offs)
    rng = np.random.default_rng(32)
    cuts = [0, 350, 350, 351, 600, 820, 900]        # sample 1 is empty, sample 2 has one read
    smp = [list(pool[cuts[i]:cuts[i + 1]]) for i in range(6)]
    # the same sequences turn up in several samples, first in the other orientation in some of them
    smp[3] = [_rc(s) for s in smp[0][:40]] + smp[3] + smp[0][:60]
    smp[4] = smp[0][10:30] + smp[4] + [_rc(s) for s in smp[4][:25]]
    smp[5] = smp[5] + smp[3][:30] + ["ACGT" * 5, "N" * 40]      # a read below the clustering commands' --minseqlength (kept: derep runs with 1), an all-N read
    for s in smp:
        rng.shuffle(s)
    return smp


def _oracle_sample(hs, seqs):
    """derep + search of ONE sample on the CPU oracle: rep_of, strand, uniques, SearchResult"""
    if not seqs:
        return np.zeros(0, np.int64), np.zeros(0, np.int8), [], None
    codes, o = orc.digitize(seqs)
    _, rep_of, strand = orc.derep(codes, o, strand_both=True)
    seeds = [i for i in range(len(seqs)) if rep_of[i] == i]
    if not seeds:
        return rep_of, strand, seeds, None
    ucodes, uo = orc.digitize([seqs[i] for i in seeds])
    return rep_of, strand, seeds, orc.SearchResult(hs, ucodes, uo, threads=8, keep_trace=0)


def test_batch_equals_each_sample_alone_on_the_oracle(engine, t_hmm_text):
    smp = _samples(t_hmm_text)
    hmm = _its2_subset(t_hmm_text, 30, 30)
    hs = orc.HmmSet(text=hmm)
    allseqs = [s for x in smp for s in x]
    sample_of = np.concatenate([np.full(len(x), i, np.int32) for i, x in enumerate(smp)])
    first = np.concatenate([[0], np.cumsum([len(x) for x in smp])])
    engine.load_profiles(text=hmm)
    engine.set_reads(allseqs)
    engine.set_samples(sample_of, len(smp))
    assert engine.n_samples == 6
    engine.derep()
    engine.search()
    domz = engine.get_domz().reshape(6, -1)
    engine.finalize()
    rep_of, strand, uniq_of = engine.get_derep()
    seed, _ = engine.get_uniques()
    dom = engine.domains()
    usample = sample_of[seed]
    coords = engine.trim_coords("3_", "4_")
    n_checked = 0
    for i, seqs in enumerate(smp):
        lo, hi = int(first[i]), int(first[i + 1])
        orep, ostrand, oseeds, res = _oracle_sample(hs, seqs)
        # a1/a7: the grouping never leaves the sample and equals the sample's own dereplication
        local = rep_of[lo:hi].copy()
        local[local >= 0] -= lo
        assert np.array_equal(local, orep) and np.array_equal(strand[lo:hi], ostrand)
        mine = np.flatnonzero(usample == i)
        assert [int(seed[u]) - lo for u in mine] == oseeds
        if res is None:
            assert not np.any(np.isin(dom["rep"], mine)) and domz[i].sum() == 0
            continue
        # a4: the sample's domain rows, thresholds included (dom_reported depends on the sample's own domZ)
        d = dom[np.isin(dom["rep"], mine)]
        od = res.domains
        assert len(d) == len(od)
        lut = {int(u): k for k, u in enumerate(mine)}
        assert np.array_equal(np.array([lut[int(r)] for r in d["rep"]]), od["seq"])
        for f in ("prof", "tlen", "ienv", "jenv", "dom_idx", "ndom", "seq_reported", "dom_reported"):
            assert np.array_equal(d[f], od[f]), (i, f)
        for f in ("envsc", "bitscore", "seq_score"):
            assert np.array_equal(_bits(d[f]), _bits(od[f])), (i, f)
        # domZ[sample][profile] = targets of THIS sample reported for the profile
        cnt = np.zeros(domz.shape[1], np.int64)
        first_dom = od[(od["dom_idx"] == 0) & (od["seq_reported"] == 1)]
        np.add.at(cnt, first_dom["prof"], 1)
        assert np.array_equal(domz[i], cnt), i
        # a5/a6 composed with a7: per-read coordinates
        ostart, ostop, otlen, oind = res.positions("3_", "4_")
        for r in range(len(seqs)):
            u = orep[r]
            if u < 0:
                assert coords[3][lo + r] == 0
                continue
            k = oseeds.index(int(u))
            assert (coords[0][lo + r], coords[1][lo + r], coords[2][lo + r], coords[3][lo + r]) == \
                (ostart[k], ostop[k], otlen[k], oind[k])
            n_checked += 1
    assert n_checked > 900
    # the same reads as ONE sample group across the old sample borders and count domZ once
    engine.set_samples(None, 1)
    assert engine.n_samples == 1
    nu1 = engine.derep()
    assert nu1 < len(seed)
    engine.search()
    assert engine.get_domz().shape[0] == domz.shape[1]


def test_set_samples_argument_errors(engine):
    from itsxpress_amd import EngineError
    engine.set_reads(["ACGT" * 10, "ACGA" * 10])
    with pytest.raises(EngineError):
        engine.set_samples(np.array([0, 2], np.int32), 2)
    engine.set_samples(np.array([0, 1], np.int32), 2)
    assert engine.derep() == 2
    with pytest.raises(EngineError):
        engine.cluster(0.97)                       # greedy clustering is sequential per sample: not batched
    with pytest.raises(EngineError):
        engine.unique_keys()
    with pytest.raises(EngineError):
        engine.select_sample(2)
    engine.set_reads(["ACGT" * 10])                 # a new read set is one sample again
    assert engine.L.itsx_num_samples(engine.h) == 1


def test_batch_files_equal_single_sample_runs(engine, fixture_reads, t_hmm_text, mini_hmm_text, tmp_path):
    """SampleBatch over SeqSample objects: uc.txt / rep.fa / domtbl.txt of every sample are byte-identical to the files
    the sample's own deduplicate() / _search() write, and the per-sample coordinate arrays are equal too."""
    from itsxpress_amd import SeqSampleNotPaired
    from itsxpress_amd.batch import SampleBatch
    names, seqs = fixture_reads
    blob, offs = synth.make_reads(t_hmm_text, 500, seed=33)
    syn = synth.to_strings(blob, offs)
    # reads of 5 and 31 bases (below vsearch's clustering default --minseqlength 32, kept by --fastx_uniques: SeqSample.py:96
    # passes no such option) in two samples, twice each: a batch and a solo run must treat them alike
    short = ["ACGTA", "ACGTTGCAAGGCTTACCGGATTTACGCAGTC", "ACGTA", "GACTGCGTAAATCCGGTAAGCCTTGCAACGT"]
    parts = [(names[:120] + ["short%d" % i for i in range(4)], seqs[:120] + short), (names[120:], seqs[120:]),
             (["s%05d extra words" % i for i in range(len(syn))] + ["t%d" % i for i in range(2)], syn + short[:2]),
             (names[:50][::-1], [_rc(s) for s in seqs[:50]][::-1])]
    hmm = tmp_path / "its2.hmm"
    hmm.write_text(mini_hmm_text + _its2_subset(t_hmm_text, 12, 12))
    fqs = []
    for k, (nm, sq) in enumerate(parts):
        fq = tmp_path / ("sample%d.fq" % k)
        with open(fq, "w") as f:
            for n, s in zip(nm, sq):
                f.write("@%s\n%s\n+\n%s\n" % (n, s, "I" * len(s)))
        fqs.append(str(fq))
    solo_dir = tmp_path / "solo"
    batch_dir = tmp_path / "batch"
    solo, solo_coords = [], []
    for k, fq in enumerate(fqs):
        d = solo_dir / str(k)
        os.makedirs(d)
        s = SeqSampleNotPaired(fq, str(d))
        s._engine = engine
        s.deduplicate(threads=1)
        s._search(hmmfile=str(hmm), threads=1)
        solo.append({f: open(getattr(s, f), "rb").read() for f in ("uc_file", "rep_file", "dom_file")})
        if k == 0:      # the short reads are in uc.txt (S, S, H +, H -): nothing below 32 bases vanished
            rows = [ln.split("\t") for ln in solo[0]["uc_file"].decode().splitlines()]
            assert sorted(r[0] + r[4] for r in rows if r[0] in "SH" and r[8].startswith("short")) == ["H+", "H-", "S*", "S*"]
        solo_coords.append([x.copy() for x in s.trim_coordinates("ITS2")])
    os.makedirs(batch_dir)
    objs = [SeqSampleNotPaired(fq, str(batch_dir)) for fq in fqs]
    b = SampleBatch(objs, engine=engine)
    b.deduplicate(threads=1)
    b._search(hmmfile=str(hmm), threads=1)
    assert list(b.counts) == [len(p[0]) for p in parts]
    per = b.trim_coordinates("ITS2")
    for k, s in enumerate(objs):
        assert len({s.uc_file, s.rep_file, s.dom_file}) == 3 and os.path.dirname(s.uc_file) != str(batch_dir)
        for f in ("uc_file", "rep_file", "dom_file"):
            assert open(getattr(s, f), "rb").read() == solo[k][f], (k, f)
        for g, e in zip(per[k], solo_coords[k]):
            assert np.array_equal(g, e)
    assert sum(len(x["dom_file"]) for x in solo) > 20000
    with pytest.raises(Exception):
        b.cluster(threads=1, cluster_id=0.99)


def test_load_reads_files_mixed_formats_and_names(engine, fixture_reads, tmp_path):
    """itsx_load_reads_files: plain FASTQ, gzip FASTQ, FASTA and an empty file as four samples; labels and sample borders"""
    import gzip
    names, seqs = fixture_reads
    a = tmp_path / "a.fq"
    a.write_text("".join("@%s x\n%s\n+\n%s\n" % (n, s, "I" * len(s)) for n, s in zip(names[:30], seqs[:30])))
    b = tmp_path / "b.fq.gz"
    b.write_bytes(gzip.compress("".join("@%s\n%s\n+\n%s\n" % (n, s, "I" * len(s)) for n, s in zip(names[30:80], seqs[30:80])).encode()))
    c = tmp_path / "c.fa"
    c.write_text("".join(">%s\n%s\n" % (n, s) for n, s in zip(names[:20], seqs[:20])))
    d = tmp_path / "empty.fq"
    d.write_text("")
    counts = engine.load_reads_files([str(a), str(b), str(c), str(d)])
    assert list(counts) == [30, 50, 20, 0] and engine.n_samples == 4 and engine.n_reads == 100
    assert engine.read_names() == names[:30] + names[30:80] + names[:20]
    engine.derep()
    rep_of, _, _ = engine.get_derep()
    # sample c repeats sample a's first 20 reads: they group inside c, never with a
    assert np.all(rep_of[80:] >= 80) and np.all(rep_of[:30] < 30)
    engine.set_reads(["ACGT" * 10, "TTTT" * 10])
    assert engine.read_names() == ["r000000000", "r000000001"]
    with pytest.raises(FileNotFoundError):
        engine.load_reads_files([str(a), str(tmp_path / "nope.fq")])

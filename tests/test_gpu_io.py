"""Host-side plumbing that needs a context (`pytest -m gpu`): the domain-table writer's thread pool and the record-parallel
FASTA / FASTQ parser behind itsx_load_reads_file / itsx_merge_pairs_files must give exactly what one thread gives."""
import numpy as np
import pytest

import synth
from test_gpu_parity import _its2_subset

pytestmark = pytest.mark.gpu


def test_domtbl_written_by_the_pool_equals_one_thread(engine, t_hmm_text, tmp_path, monkeypatch):
    """domtbl.txt of a few hundred thousand rows: blocks formatted by a pool of threads are the file one thread writes."""
    blob, offs = synth.make_reads(t_hmm_text, 4000, seed=35)
    seqs = synth.to_strings(blob, offs)
    engine.load_profiles(text=_its2_subset(t_hmm_text))
    engine.set_reads(seqs)
    engine.derep()
    engine.search()
    engine.finalize()
    files = {}
    for threads in ("1", "7"):
        monkeypatch.setenv("ITSX_IO_THREADS", threads)
        p = tmp_path / ("domtbl_%s.txt" % threads)
        engine.write_domtbl(str(p))
        files[threads] = p.read_bytes()
    assert files["1"] == files["7"]
    rows = [ln for ln in files["1"].split(b"\n") if ln and not ln.startswith(b"#")]
    assert len(rows) > 150000                                      # several blocks of 32768 rows
    # file order = profile order, then targets in rep.fa order, then domains
    names = engine.profile_names()
    seen = [(names.index(r.split()[3].decode()), int(r.split()[0][1:]), int(r.split()[9])) for r in rows[::97]]
    assert seen == sorted(seen)


def test_parallel_fastx_parse_equals_serial(engine, tmp_path, monkeypatch):
    """Large FASTQ / FASTA texts are cut at record starts and parsed by the I/O pool: same reads, same labels, same order as
    one thread -- with quality lines that start with '@' and '+', CRLF line ends and multi-line FASTA in the input."""
    rng = np.random.default_rng(41)
    n = 3000
    seqs = ["".join(rng.choice(list("ACGTN"), int(rng.integers(40, 300)), p=[.245, .245, .245, .245, .02])) for _ in range(n)]
    fq = tmp_path / "in.fq"
    with open(fq, "w", newline="") as f:
        for i, s in enumerate(seqs):
            q = "".join(rng.choice(list("@+I5#>"), len(s)))         # qualities that look like headers
            eol = "\r\n" if i % 7 == 0 else "\n"
            f.write("@r%d some words%s%s%s+%s%s%s" % (i, eol, s, eol, eol, q, eol))
    fa = tmp_path / "in.fa"
    with open(fa, "w") as f:
        for i, s in enumerate(seqs):
            f.write(">r%d desc\n" % i + "\n".join(s[j:j + 60] for j in range(0, len(s), 60)) + "\n")
    monkeypatch.setenv("ITSX_TEXT_CACHE_GB", "0")
    for path in (fq, fa):
        got = {}
        for threads, minmb in (("1", "1000"), ("5", "0"), ("16", "0")):
            monkeypatch.setenv("ITSX_IO_THREADS", threads)
            monkeypatch.setenv("ITSX_PARSE_MIN_MB", minmb)
            assert engine.load_reads_file(str(path)) == n
            names = engine.read_names()
            engine.derep(strand_both=False, minseqlength=0)
            rep_of, _, _ = engine.get_derep()
            got[threads] = (names, rep_of.copy())
        assert got["1"][0] == ["r%d" % i for i in range(n)]
        for t in ("5", "16"):
            assert got[t][0] == got["1"][0] and np.array_equal(got[t][1], got["1"][1])
        # and the reads are the reads: identical sequences group, different ones do not
        first = {}
        exp = np.array([first.setdefault(s, i) for i, s in enumerate(seqs)])
        assert np.array_equal(got["5"][1], exp)
    # a damaged record is still reported
    bad = tmp_path / "bad.fq"
    bad.write_text(open(fq, newline="").read().replace("\n+\n", "\n-\n", 1))
    from itsxpress_amd import EngineError
    with pytest.raises(EngineError):
        engine.load_reads_file(str(bad))

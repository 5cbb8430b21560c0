"""BASELINE configs[4] as a tested configuration: 2x250-merged-shaped reads (300-480 bases), Fungi `--region ALL`
(`1_` / `4_` profiles: main.py:200-208), `cluster_id 0.995` (SeqSample.cluster, SeqSample.py:133-176; main.py:534-537).

* oracle size: the whole configuration -- greedy clustering, then the HMM stages on the centroids, then every read's
  coordinates -- equals the CPU oracle (orc_cluster.c + orc_search.c), value for value;
* full size (the largest that stays inside a few minutes on one GPU): properties the greedy procedure guarantees whatever
  the size -- clustering the centroids again changes nothing; every member meets the threshold against ITS centroid when
  re-aligned by the oracle's aligner, with exactly the identity the engine reported; the window size of the speculative
  scheme does not change a single outcome; per-read coordinates are the centroid's.
`pytest -m gpu`."""
import numpy as np
import pytest

import orc
import synth

pytestmark = pytest.mark.gpu

CID = 0.995
N_FULL = 6_000_000          # about half of one GPU's share of configs[4] (12.5 M): 43 s of clustering on an MI355X (profiles/round3_cluster_curve.jsonl)


def _region_all(hmm_text):
    blocks = [b + "//\n" for b in hmm_text.split("//\n") if "NAME  " in b]
    return "".join(b for b in blocks if b.split("NAME  ")[1][:2] in ("1_", "4_"))


def _cfg4_reads(hmm_text, n, seed=None, as_array=False):
    return synth.make_reads(hmm_text, n, config=5, left="1_", right="4_", fixed_len=0, len_range=(300, 480), seed=seed, as_array=as_array)


def test_cfg4_shape_equals_oracle(engine, t_hmm_text):
    """clustering at 0.995 + the `1_` / `4_` profiles on merged-pair-shaped reads, at oracle size"""
    hmm = _region_all(t_hmm_text)
    assert hmm.count("NAME  1_") >= 5 and hmm.count("NAME  4_") >= 50
    blob, offs = _cfg4_reads(t_hmm_text, 2500, seed=505)
    reads = synth.to_strings(blob, offs)
    assert min(map(len, reads)) >= 300 and max(map(len, reads)) <= 480 and len(set(map(len, reads))) > 20
    names = ["r%09d" % i for i in range(len(reads))]
    engine.load_profiles(text=hmm)
    engine.set_reads(reads, names)
    ncl = engine.cluster(CID, strand_both=True)
    rep_of, strand, uniq_of = engine.get_derep()
    pct, order = engine.get_cluster()
    codes, off = orc.digitize(reads)
    o = orc.cluster(codes, off, names, CID)
    assert ncl == o["n_centroids"] and 0 < ncl < len(reads)
    assert np.array_equal(order, o["order"]) and np.array_equal(rep_of, o["rep_of"]) and np.array_equal(strand, o["strand"])
    assert np.array_equal(pct.view(np.uint64), o["pct_id"].view(np.uint64))
    assert (strand < 0).sum() > 20
    # the HMM stages run on the centroids (rep.fa of SeqSample.cluster), every read takes its centroid's coordinates
    engine.search()
    engine.finalize()
    start, stop, tlen, ind = engine.trim_coords("1_", "4_")
    seeds = [i for i in range(len(reads)) if o["rep_of"][i] == i]
    c2, o2 = orc.digitize([reads[i] for i in seeds])
    res = orc.SearchResult(orc.HmmSet(text=hmm), c2, o2, threads=8, keep_trace=0)
    us, ue, ut, ui = res.positions("1_", "4_")
    uniq = np.cumsum(o["rep_of"] == np.arange(len(reads))) - 1
    uo = uniq[o["rep_of"]]
    assert np.array_equal(start, us[uo]) and np.array_equal(stop, ue[uo]) and np.array_equal(tlen, ut[uo]) and np.array_equal(ind, ui[uo])
    both = (start >= 0) & (stop >= 0)
    assert both.mean() > 0.9 and (stop[both] > start[both]).all()
    # --region ALL keeps SSU-end .. LSU-start: the kept part spans the spacer and both motifs' inner ends
    assert np.median(stop[both] - start[both]) > 100


@pytest.fixture(scope="module")
def full4(engine, t_hmm_text):
    blob, offs = _cfg4_reads(t_hmm_text, N_FULL, as_array=True)
    engine.load_profiles(text=_region_all(t_hmm_text))
    engine.set_reads_buffer(blob, offs)
    ncl = engine.cluster(CID, strand_both=True)
    rep_of, strand, uniq_of = engine.get_derep()
    pct, order = engine.get_cluster()
    st = engine.stats()
    engine.set_rows_mode("lazy")                 # (the centroids' full table would hold ~100 rows each; the coordinates are what is checked)
    try:
        engine.search()
        engine.finalize()
        coords = engine.trim_coords("1_", "4_")
        rcoords = engine.rep_coords("1_", "4_")
    finally:
        engine.set_rows_mode(None)
    return dict(blob=blob, offs=offs, ncl=ncl, rep_of=rep_of, strand=strand, uniq_of=uniq_of, pct=pct, order=order, stats=st,
                coords=coords, rcoords=rcoords)


def _seq(full, i):
    return bytes(full["blob"][full["offs"][i]:full["offs"][i + 1]]).decode()


_RC = str.maketrans("ACGTN", "TGCAN")


def test_full_size_cluster_structure(full4):
    r, n = full4["rep_of"], N_FULL
    assert (r >= 0).all()                                        # nothing below --minseqlength here
    assert np.array_equal(r[r], r)                               # a centroid is its own centroid
    assert (r <= np.arange(n)).all()                             # labels are in input order: a centroid precedes its members
    seeds = np.flatnonzero(r == np.arange(n))
    assert len(seeds) == full4["ncl"] and 0.1 * n < len(seeds) < 0.6 * n
    assert np.array_equal(full4["order"], np.arange(n))
    members = r != np.arange(n)
    assert (full4["pct"][members] >= 100.0 * CID).all() and (full4["pct"][~members] == -1.0).all()
    assert (full4["strand"][~members] == 1).all() and 0.02 < (full4["strand"][members] < 0).mean() < 0.2
    assert full4["stats"]["ms_cluster"] < 80000.0                 # wall-time guard: 43 s on an MI355X in round 3


def test_full_size_members_meet_the_threshold_by_the_oracles_aligner(full4):
    """an independent check of the accepted hits: the oracle's global alignment of (member, its centroid) gives exactly the
    identity the engine reported, on the strand it reported, and that identity passes"""
    rng = np.random.default_rng(7)
    members = np.flatnonzero(full4["rep_of"] != np.arange(N_FULL))
    minus = members[full4["strand"][members] < 0]
    below100 = members[full4["pct"][members] < 100.0]
    pick = np.concatenate([rng.choice(members, 400, replace=False), rng.choice(minus, 100, replace=False), rng.choice(below100, 200, replace=False)])
    for i in pick:
        q, t = _seq(full4, int(i)), _seq(full4, int(full4["rep_of"][i]))
        if full4["strand"][i] < 0:
            q = q[::-1].translate(_RC)
        sc, m, cols = orc.align_identity(q, t)
        pid = 100.0 * m / cols
        assert pid == full4["pct"][i] and pid >= 100.0 * CID, (int(i), pid, float(full4["pct"][i]))


def test_full_size_clustering_the_centroids_again_changes_nothing(engine, full4):
    """greedy clustering is idempotent on its own centroids: every centroid met, when it was a query, exactly the centroids
    that precede it now, and found no hit among them"""
    seeds = np.flatnonzero(full4["rep_of"] == np.arange(N_FULL))
    sub = seeds[:300000]                                         # a prefix of the centroids is closed under "precedes"
    offs = full4["offs"]
    lens = (offs[sub + 1] - offs[sub]).astype(np.int64)
    o2 = np.zeros(len(sub) + 1, np.int64)
    np.cumsum(lens, out=o2[1:])
    idx = np.repeat(offs[sub] - o2[:-1], lens) + np.arange(int(o2[-1]))
    blob2 = np.ascontiguousarray(full4["blob"][idx])
    engine.set_reads_buffer(blob2, o2)
    assert engine.cluster(CID, strand_both=True) == len(sub)
    rep2, _, _ = engine.get_derep()
    assert np.array_equal(rep2, np.arange(len(sub)))


def test_full_size_window_size_does_not_change_the_outcome(engine, full4, monkeypatch):
    n = 150000
    blob = np.ascontiguousarray(full4["blob"][:int(full4["offs"][n])])
    offs = np.ascontiguousarray(full4["offs"][:n + 1])
    out = []
    for w in ("4096", "1000"):
        monkeypatch.setenv("ITSX_CL_WINDOW", w)
        engine.set_reads_buffer(blob, offs)
        engine.cluster(CID, strand_both=True)
        rep_of, strand, _ = engine.get_derep()
        pct, _ = engine.get_cluster()
        out.append((rep_of.copy(), strand.copy(), pct.copy()))
    assert all(np.array_equal(a, b) for a, b in zip(out[0], out[1]))
    # and a prefix of the input clusters like the same reads inside the full run (the procedure is sequential)
    assert np.array_equal(out[0][0], full4["rep_of"][:n]) and np.array_equal(out[0][2].view(np.uint64), full4["pct"][:n].view(np.uint64))


def test_full_size_reads_take_their_centroids_coordinates(full4):
    start, stop, tlen, ind = full4["coords"]
    rs, re_, rt, ri = full4["rcoords"]
    u = full4["uniq_of"]
    assert np.array_equal(start, rs[u]) and np.array_equal(stop, re_[u]) and np.array_equal(tlen, rt[u])
    both = (start >= 0) & (stop >= 0)
    assert both.mean() > 0.9
    # tlen is the CENTROID's length (SeqSample.py:429 records the representative's tlen), not the read's
    seeds = np.flatnonzero(full4["rep_of"] == np.arange(N_FULL))
    clen = (full4["offs"][seeds + 1] - full4["offs"][seeds])
    assert np.array_equal(rt[rt >= 0], clen[rt >= 0])

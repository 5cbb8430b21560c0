"""Size-independent properties at BASELINE.json's full single-GPU size (configs[1]: 1M x 300 bp).
The oracle cannot run at this size in seconds, so the checks are invariants of the path itself:
cluster maps against an independent numpy/hash grouping, idempotence, permutation and
reverse-complement invariance, and agreement between the text-file route and the array route."""
import hashlib

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu

N_FULL = 1_000_000
_COMP = np.zeros(256, np.uint8)
for a, b in zip(b"ACGTN", b"TGCAN"):
    _COMP[a] = b


def _its2(hmm_text):
    blocks = [b + "//\n" for b in hmm_text.split("//\n") if "NAME  " in b]
    return "".join(b for b in blocks if b.split("NAME  ")[1][:2] in ("3_", "4_"))


@pytest.fixture(scope="module")
def full(engine, t_hmm_text):
    blob, offs = synth.make_reads(t_hmm_text, N_FULL, config=2)
    engine.load_profiles(text=_its2(t_hmm_text))
    engine.set_reads_buffer(blob, offs)
    nu = engine.derep()
    engine.search()
    engine.finalize()
    rep_of, strand, uniq_of = engine.get_derep()
    coords = engine.trim_coords("3_", "4_")
    return dict(blob=blob, offs=offs, nu=nu, rep_of=rep_of, strand=strand, uniq_of=uniq_of, coords=coords,
                stats=engine.stats(), domz=engine.get_domz())


def test_full_size_cluster_map_matches_independent_grouping(full):
    reads = np.frombuffer(full["blob"], np.uint8).reshape(N_FULL, 300)
    rc = _COMP[reads[:, ::-1]]
    # canonical orientation = lexicographically smaller of (read, revcomp); group by a digest of it
    first_diff = (reads != rc).argmax(axis=1)
    pick_rc = rc[np.arange(N_FULL), first_diff] < reads[np.arange(N_FULL), first_diff]
    canon = np.where(pick_rc[:, None], rc, reads)
    first = {}
    exp_rep = np.empty(N_FULL, np.int64)
    for i in range(N_FULL):
        k = hashlib.blake2b(canon[i].tobytes(), digest_size=12).digest()
        exp_rep[i] = first.setdefault(k, i)
    assert full["nu"] == len(first)
    assert np.array_equal(full["rep_of"], exp_rep)
    same = (reads == reads[exp_rep]).all(axis=1)
    assert np.array_equal(full["strand"], np.where(same, 1, -1).astype(np.int8))
    # idempotence and order: seeds are their own representative and precede their members
    r = full["rep_of"]
    assert np.array_equal(r[r], r) and (r <= np.arange(N_FULL)).all()


def test_full_size_coordinates_are_cluster_consistent_and_in_range(full):
    start, stop, tlen, ind = full["coords"]
    r = full["rep_of"]
    for a in (start, stop, tlen, ind):
        assert np.array_equal(a, a[r])                      # every read carries its representative's result
    both = (start >= 0) & (stop >= 0)
    assert both.mean() > 0.9
    assert (tlen[both] == 300).all() and (start[both] >= 45).all() and (stop[both] <= 300 - 44).all()
    st = full["stats"]
    assert st["n_past_msv"] >= st["n_past_bias"] >= st["n_past_fwd"] > 0
    assert st["n_env_unique"] <= st["n_domains"] and st["hash_reseeds"] == 0 and st["n_domain_overflow"] == 0


def test_search_is_invariant_under_permutation_revcomp_and_dereplication(engine, t_hmm_text, full):
    """Same answers per sequence when (a) only the unique sequences are given, (b) the reads are shuffled --
    a checksum of the whole path.  Forward-strand dereplication is used: with --strand both the first
    occurrence fixes the cluster's orientation, so a shuffle legitimately changes which strand is scored."""
    n = 150_000
    reads = np.frombuffer(full["blob"], np.uint8).reshape(N_FULL, 300)[:n]
    offs = np.arange(n + 1, dtype=np.int64) * 300
    engine.load_profiles(text=_its2(t_hmm_text))

    def run(mat):
        engine.set_reads_buffer(mat.tobytes(), np.arange(mat.shape[0] + 1, dtype=np.int64) * 300)
        engine.derep(strand_both=False)
        engine.search()
        engine.finalize()
        rep_of, strand, uniq_of = engine.get_derep()
        return rep_of, strand, np.stack(engine.trim_coords("3_", "4_"), axis=1), engine.get_domz()

    rep0, strand0, c0, z0 = run(reads)
    # (b) permutation: domZ identical and per-read coordinates follow the permutation
    rng = np.random.default_rng(4)
    perm = rng.permutation(n)
    rep1, strand1, c1, z1 = run(reads[perm])
    assert np.array_equal(z0, z1)
    assert np.array_equal(c1, c0[perm])
    # (a) dereplication is idempotent: searching only the seeds gives the seeds' own rows
    seeds = np.flatnonzero(rep0 == np.arange(n))
    rep2, strand2, c2, z2 = run(reads[seeds])
    assert np.array_equal(rep2, np.arange(len(seeds))) and np.array_equal(z2, z0)
    assert np.array_equal(c2, c0[seeds])


# ------------------------------------------------------------------------------------------------------------
# BASELINE configs[2]: 10 M merged reads of 300-580 bases on one GPU -- the size bench.py's default line is quoted on.
N_CFG2 = 10_000_000


@pytest.fixture(scope="module")
def full2(engine, t_hmm_text):
    import time
    blob, offs = synth.make_reads(t_hmm_text, N_CFG2, config=3, fixed_len=0, len_range=(300, 580), as_array=True)
    engine.load_profiles(text=_its2(t_hmm_text))
    t0 = time.time()
    engine.set_reads_buffer(blob, offs)
    nu = engine.derep()
    engine.search()
    engine.finalize()
    coords = engine.trim_coords("3_", "4_")
    wall = time.time() - t0
    rep_of, strand, uniq_of = engine.get_derep()
    return dict(blob=blob, offs=offs, nu=nu, rep_of=rep_of, strand=strand, uniq_of=uniq_of, coords=coords,
                stats=engine.stats(), domz=engine.get_domz(), wall=wall)


def test_cfg2_size_runs_in_chunks_within_its_time_budget(full2):
    st = full2["stats"]
    assert st["n_reads"] == N_CFG2 and st["n_unique"] == full2["nu"] > N_CFG2 // 10
    assert st["msv_launches"] > 1                                # the unique list went through several chunks
    assert st["n_domain_overflow"] == 0 and st["hash_reseeds"] == 0
    assert st["n_past_msv"] >= st["n_past_bias"] >= st["n_past_fwd"] > 0
    assert st["n_reads_region_cap"] == 0
    assert full2["wall"] < 90.0, "one cold pass over 10 M reads took %.1f s" % full2["wall"]     # ~10 s warm; first-use allocations included


def test_cfg2_size_coordinates_are_cluster_consistent_and_in_range(full2):
    start, stop, tlen, ind = full2["coords"]
    r = full2["rep_of"]
    lens = np.diff(full2["offs"])
    assert (r >= 0).all() and np.array_equal(r[r], r) and (r <= np.arange(N_CFG2)).all()
    for a in (start, stop, tlen, ind):
        assert np.array_equal(a, a[r])                           # every read carries its representative's result
    both = (start >= 0) & (stop >= 0)
    assert both.mean() > 0.9
    assert np.array_equal(tlen[both], lens[r][both])             # tlen = length of the representative
    assert (lens == lens[r]).all()                               # exact dereplication: members have the seed's length
    assert (start[both] >= 45).all() and (stop[both] <= tlen[both] - 44).all()
    hit = ind > 0
    assert ((start >= 0) | (stop >= 0))[~hit].sum() == 0         # no coordinates without a reported domain


def test_cfg2_size_first_million_matches_an_independent_grouping(full2):
    """first occurrences never look forward, so the engine's grouping of all 10 M reads, restricted to the first
    million, must equal an independent grouping of that million alone"""
    n = 1_000_000
    blob, offs = full2["blob"], full2["offs"]
    first = {}
    exp_rep = np.empty(n, np.int64)
    exp_strand = np.empty(n, np.int8)
    raw = blob[:int(offs[n])].tobytes()
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    for i in range(n):
        s = raw[offs[i]:offs[i + 1]]
        k = first.get(s)
        if k is None:
            rcs = s.translate(comp)[::-1]
            k = first.get(rcs)
            if k is None:
                first[s] = (i, s)
                exp_rep[i] = i; exp_strand[i] = 1
                continue
        exp_rep[i] = k[0]
        exp_strand[i] = 1 if k[1] == s else -1
    assert np.array_equal(full2["rep_of"][:n], exp_rep)
    assert np.array_equal(full2["strand"][:n], exp_strand)


def test_cfg2_shape_is_invariant_under_permutation_and_dereplication(engine, t_hmm_text, full2):
    """the checksum-of-the-whole-path properties of the 1 M test on half a million of the ragged reads"""
    n = 500_000
    blob, offs = full2["blob"], full2["offs"]
    lens = np.diff(offs[:n + 1])
    engine.load_profiles(text=_its2(t_hmm_text))

    def take(idx):
        ln = lens[idx]
        o = np.zeros(len(idx) + 1, np.int64)
        np.cumsum(ln, out=o[1:])
        src = np.repeat(offs[idx] - o[:-1], ln) + np.arange(o[-1])
        return blob[src], o

    def run(b, o):
        engine.set_reads_buffer(b, o)
        engine.derep(strand_both=False)
        engine.search()
        engine.finalize()
        rep_of, strand, uniq_of = engine.get_derep()
        return rep_of, np.stack(engine.trim_coords("3_", "4_"), axis=1), engine.get_domz()

    rep0, c0, z0 = run(blob[:int(offs[n])], offs[:n + 1])
    perm = np.random.default_rng(6).permutation(n)
    rep1, c1, z1 = run(*take(perm))
    assert np.array_equal(z0, z1) and np.array_equal(c1, c0[perm])
    seeds = np.flatnonzero(rep0 == np.arange(n))
    rep2, c2, z2 = run(*take(seeds))
    assert np.array_equal(rep2, np.arange(len(seeds))) and np.array_equal(z2, z0) and np.array_equal(c2, c0[seeds])


# ------------------------------------------------------------------------------------------------------------
# BASELINE configs[3]: 50 M reads against --taxa All (814 ITS2 profiles), read-sharded over 8 GPUs = 6.25 M reads x 814 profiles per
# GPU.  One GPU's share at a fifth of its size (the whole share takes 26 s and is a bench run: `bench.py --taxa all --reads 6250000`),
# with the row compaction the bench runs with: the properties that do not depend on the size, and the sharding identity the N > 1 path
# relies on (two shards + summed domZ == one run).
N_CFG3 = 1_250_000


def test_cfg3_shape_all_taxa_profiles_compacted_rows_and_two_shards(t_hmm_text, all_its2_hmm_text, monkeypatch):
    from itsxpress_amd import Engine
    engine = Engine(0)          # a context of its own, closed at the end: its work buffers (which only grow) go back to the device
    try:
        _cfg3_body(engine, t_hmm_text, all_its2_hmm_text, monkeypatch)
    finally:
        engine.close()


def _cfg3_body(engine, t_hmm_text, all_its2_hmm_text, monkeypatch):
    blob, offs = synth.make_reads(t_hmm_text, N_CFG3, config=4, fixed_len=0, len_range=(300, 580), as_array=True)
    assert engine.load_profiles(text=all_its2_hmm_text) == 814
    monkeypatch.setenv("ITSX_COMPACT_ROWS", "1")

    def run(b, o, domz=None):
        engine.set_reads_buffer(b, o)
        engine.derep(strand_both=True, minseqlength=1)
        engine.search()
        z = engine.get_domz().copy()
        if domz is not None:
            engine.set_domz(domz)
        engine.finalize()
        return [a.copy() for a in engine.trim_coords("3_", "4_")], z, engine.get_derep()[0].copy(), engine.stats()

    c, z, rep, st = run(blob, offs)
    start, stop, tlen, ind = c
    assert st["n_profiles"] == 814 and st["n_pairs"] == st["n_unique"] * 814
    assert st["n_rows_resident"] * 20 < st["n_domains"]           # ~1.5 rows per representative and side instead of ~200 per representative
    assert st["n_domain_overflow"] == 0 and st["n_mr_failed"] == 0
    for a in c:
        assert np.array_equal(a, a[rep])                         # every read carries its representative's result
    both = (start >= 0) & (stop >= 0)
    assert both.mean() > 0.9 and (stop[both] > start[both]).all()
    assert np.array_equal(tlen[both], np.diff(offs)[rep][both])
    # two contiguous shards searched on their own, finalized with the SUMMED domZ (what the all-reduce hands every rank), give the
    # coordinates of the single run for every read whose representative lies in its own shard
    h = N_CFG3 // 2
    oa = offs[:h + 1]
    ob = offs[h:] - offs[h]
    ba, bb = blob[:int(offs[h])], blob[int(offs[h]):]
    _, za, _, _ = run(ba, oa)
    _, zb, _, _ = run(bb, ob)
    zsum = za + zb
    ca, _, _, _ = run(ba, oa, zsum)
    cb, _, repb, _ = run(bb, ob, zsum)
    # (per-shard dereplication counts a sequence present in both shards twice in domZ: section 7's option 1; the single run's domZ is smaller)
    assert (zsum >= z).all()
    same_a = all(np.array_equal(x[:h], y) for x, y in zip(c, ca))
    own = rep[h:] >= h                                           # reads of shard b whose first occurrence is not in shard a
    frac = np.mean([(x[h:][own] == y[own]).mean() for x, y in zip(c, cb)])
    assert same_a or np.mean([(x[:h] == y).mean() for x, y in zip(c, ca)]) > 0.9999
    assert frac > 0.9999                                         # only domains within a fraction of a bit of the E-value threshold may move

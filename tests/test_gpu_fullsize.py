"""Size-independent properties at BASELINE.json's full single-GPU size (configs[1]: 1M x 300 bp).
The oracle cannot run at this size in seconds, so the checks are invariants of the path itself:
cluster maps against an independent numpy/hash grouping, idempotence, permutation and
reverse-complement invariance, and agreement between the text-file route and the array route."""
import hashlib
import time

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


def test_cfg2_size_lazy_stage_equals_the_full_table(engine, t_hmm_text, full2):
    """the bench's mode at the bench's size: the lazy domain stage's coordinates == the full table's for all 10 M reads, with a
    twentieth of the pairs evaluated (undecided rows, if any, settled by counting their profiles -- never by the full search)"""
    engine.set_rows_mode("lazy")
    try:
        engine.load_profiles(text=_its2(t_hmm_text))
        t0 = time.time()
        engine.set_reads_buffer(full2["blob"], full2["offs"])
        engine.derep()
        engine.search()
        engine.finalize()
        c = engine.trim_coords("3_", "4_")
        wall = time.time() - t0
        st = engine.stats()
    finally:
        engine.set_rows_mode(None)
    assert all(np.array_equal(a, b) for a, b in zip(c, full2["coords"]))
    assert st["lazy"] == 1 and st["n_lazy_reruns"] == 0
    assert st["n_past_msv"] == full2["stats"]["n_past_msv"]
    assert st["n_lazy_evaluated"] * 10 < st["n_past_msv"] and st["n_lazy_completed_profiles"] < st["n_profiles"] // 4
    assert st["n_rows_resident"] < 0.05 * full2["stats"]["n_rows_resident"]
    assert wall < 45.0, wall


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
# GPU: ONE GPU's FULL share, in the mode the bench runs (the lazy domain stage): the properties that do not depend on the size, and
# the sharding identity the N > 1 path relies on -- two shards with EXACT global dereplication and summed counters == one run, for
# every read (`==`, not "nearly": round 3's per-shard variant could only promise 0.9999).
N_CFG3 = 6_250_000


def test_cfg3_full_share_all_taxa_profiles_lazy_and_two_exact_shards(t_hmm_text, all_its2_hmm_text):
    from itsxpress_amd import Engine
    import shards
    engines = [Engine(0), Engine(0)]   # contexts of their own, closed at the end: their work buffers (which only grow) go back to the device
    try:
        blob, offs = synth.make_reads(t_hmm_text, N_CFG3, config=4, fixed_len=0, len_range=(300, 580), as_array=True)
        e = engines[0]
        assert e.load_profiles(text=all_its2_hmm_text) == 814
        e.set_rows_mode("lazy")
        t0 = time.time()
        e.set_reads_buffer(blob, offs)
        e.derep(strand_both=True, minseqlength=1)
        e.search()
        e.finalize()
        c = [a.copy() for a in e.trim_coords("3_", "4_")]
        wall = time.time() - t0
        rep = e.get_derep()[0].copy()
        st = e.stats()
        e.set_rows_mode(None)
        start, stop, tlen, ind = c
        assert st["n_profiles"] == 814 and st["n_pairs"] == st["n_unique"] * 814 and st["lazy"] == 1 and st["n_lazy_reruns"] == 0
        assert st["n_lazy_evaluated"] * 10 < st["n_past_msv"]           # the stage really skips: < 10 % of the pairs reach Backward
        assert st["n_rows_resident"] < 4 * st["n_unique"] + 1000         # ~1 row per representative and side
        assert st["n_domain_overflow"] == 0 and st["n_mr_failed"] == 0
        for a in c:
            assert np.array_equal(a, a[rep])                         # every read carries its representative's result
        both = (start >= 0) & (stop >= 0)
        assert both.mean() > 0.9 and (stop[both] > start[both]).all()
        assert np.array_equal(tlen[both], np.diff(offs)[rep][both])
        assert wall < 60.0, wall                                       # round 3's full pipeline took 26 s for this share
        # two contiguous shards, exact global dereplication (owner's step), summed counters: the single run's rows for EVERY read
        h = N_CFG3 // 2
        parts = [(blob[:int(offs[h])], offs[:h + 1]), (blob[int(offs[h]):], offs[h:] - offs[h])]
        rows, z, sts = shards.run_shards(engines, parts, all_its2_hmm_text)
        assert np.array_equal(rows, np.stack(c, axis=1))
        assert sum(s["n_lazy_evaluated"] for s in sts) < 0.1 * sum(s["n_past_msv"] for s in sts)
    finally:
        for e in engines:
            e.close()

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

"""GPU parity tests: the HIP engine (through the C ABI) against the CPU oracle and the golden
fixtures.  Bar: bit-exact -- trim coordinates, cluster maps, MSV bytes and every float score
are compared for equality, not closeness.  Run on the GPU box with `pytest -m gpu`.
"""
import json
import os

import numpy as np
import pytest

import orc
import synth

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _its2_subset(hmm_text, n3=None, n4=None):
    blocks = [b + "//\n" for b in hmm_text.split("//\n") if "NAME  " in b]
    l3 = [b for b in blocks if b.split("NAME  ")[1].startswith("3_")]
    l4 = [b for b in blocks if b.split("NAME  ")[1].startswith("4_")]
    return "".join(l3[:n3] + l4[:n4])


# --------------------------------------------------------------------------------------------
def test_library_loads_and_reports_device(engine):
    assert engine.L.itsx_abi_version() == 6


def test_detmath_device_matches_oracle_bitwise(engine):
    rng = np.random.default_rng(1)
    x = np.concatenate([np.exp(rng.uniform(-80, 80, 200000)), rng.uniform(1e-3, 4e4, 100000),
                        np.float32(rng.uniform(0.5, 1.5, 100000)).astype(np.float64)])
    dl, _ = engine.debug_detmath(x)
    L = orc.lib()
    ol = np.array([L.orc_det_log(float(v)) for v in x[:50000]])
    assert np.array_equal(dl[:50000].view(np.uint64), ol.view(np.uint64))
    y = rng.uniform(-60, 60, 50000)
    _, de = engine.debug_detmath(y)
    oe = np.array([L.orc_det_exp(float(v)) for v in y])
    assert np.array_equal(de.view(np.uint64), oe.view(np.uint64))
    # and both agree with libm after rounding to float (what the pipeline stores)
    assert np.array_equal(np.float32(dl), np.float32(np.log(x)))
    # the bias filter's table-driven logarithm == (float)det_log((double)x) on the device, bit for bit: 3 M floats over the whole range,
    # crowded around 1 (where the result is tiny) and on the table's interval borders (checked for EVERY positive float on the
    # host when it was written: 0 mismatches, 203 arguments fall back to det_log)
    f = np.concatenate([np.float32(np.exp(rng.uniform(-87, 88, 1500000))), np.float32(rng.uniform(1e-4, 4.0, 1000000)),
                        np.float32(1.0 + rng.uniform(-1e-3, 1e-3, 400000)),
                        (np.uint32(0x3f3504f3) + np.arange(0, 100000, dtype=np.uint32) * np.uint32(167)).view(np.float32)])
    dlf, _ = engine.debug_detmath(f.astype(np.float64))
    assert np.array_equal(engine.debug_logf(f).view(np.uint32), np.float32(dlf).view(np.uint32))


def test_profile_tables_match_oracle(engine, mini_hmm_text):
    n = engine.load_profiles(text=mini_hmm_text)
    hs = orc.HmmSet(text=mini_hmm_text)
    assert n == hs.n and engine.profile_names() == hs.names
    for i in range(n):
        t = engine.profile_tables(i)
        assert t["M"] == hs.M[i]
        assert np.array_equal(t["rbv"], hs.rbv(i))
        assert np.array_equal(_bits(t["rfv"]), _bits(hs.rfv(i)))
        assert np.array_equal(_bits(t["tfv"]), _bits(hs.tfv(i)))
        assert {k: t[k] for k in ("base", "bias", "tbm", "tec")} == hs.msvparams(i)


def test_packed_keys_hash_like_xxhash(engine):
    import xxhash
    rng = np.random.default_rng(5)
    seqs = []
    for L in [32, 33, 47, 48, 49, 63, 64, 65, 127, 128, 129, 300, 301, 441, 1000]:
        s = "".join(rng.choice(list("ACGT"), L))
        seqs.append(s)
        s2 = list(s)
        for p in rng.choice(L, 3, replace=False):
            s2[p] = rng.choice(list("NRYKMSWBDHV"))
        seqs.append("".join(s2))
    engine.set_reads(seqs)
    hf, hr = engine.debug_read_hashes()
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N", "R": "Y", "Y": "R", "M": "K", "K": "M",
            "S": "S", "W": "W", "H": "D", "D": "H", "B": "V", "V": "B"}
    code = {c: i for i, c in enumerate("ACGT-RYMKSWHBVDN")}

    def key_bytes(s):
        nw = max(1, (len(s) + 15) // 16)
        w = np.zeros(nw, np.uint32)
        exc = []
        for i, ch in enumerate(s):
            c = code[ch]
            if c <= 3:
                w[i >> 4] |= np.uint32(c << (2 * (i & 15)))
            else:
                exc.append((i << 4) | c)
        return w.tobytes() + np.array(exc, np.uint32).tobytes() + np.uint32(len(s)).tobytes()

    for i, s in enumerate(seqs):
        w, e = engine.debug_packed_read(i)
        assert (w.tobytes() + e.tobytes() + np.uint32(len(s)).tobytes()) == key_bytes(s)
        assert int(hf[i]) == xxhash.xxh64(key_bytes(s), seed=0).intdigest(), (i, len(s))
        rc = "".join(comp[c] for c in reversed(s))
        assert int(hr[i]) == xxhash.xxh64(key_bytes(rc), seed=0).intdigest(), (i, len(s))


# --------------------------------------------------------------------------------------------
def test_derep_reproduces_reference_fixture(engine, fixture_reads, gold, tmp_path):
    """a1 + a7: the frozen vsearch output of the reference's own tests (ex_tmpdir/uc.txt, rep.fa)."""
    names, seqs = fixture_reads
    engine.set_reads(seqs, names)
    nu = engine.derep()
    assert nu == 137
    rep_of, strand, uniq_of = engine.get_derep()
    md = json.load(open(os.path.join(gold, "matchdict.json")))
    assert len(md) == 227
    for i, nm in enumerate(names):
        assert names[int(rep_of[i])] == md[nm]
    assert (strand == 1).all()
    uc, fa = str(tmp_path / "uc.txt"), str(tmp_path / "rep.fa")
    engine.write_uc(uc)
    engine.write_rep_fasta(fa)
    assert open(uc).read() == open(os.path.join(gold, "fixture_uc.txt")).read()
    assert open(fa).read() == open(os.path.join(gold, "fixture_rep.fa")).read()


def test_derep_matches_oracle_on_synthetic(engine, t_hmm_text):
    blob, offs = synth.make_reads(t_hmm_text, 50000, seed=11)
    seqs = synth.to_strings(blob, offs)
    # edge cases: empty-ish/short reads (dropped), palindromes, an exact reverse complement pair with N
    seqs += ["ACGT" * 7, "ACGT" * 8, "ACGT" * 8, "A" * 31, "ACGTNACGT" * 5, "ACGTNACGT"[::-1].translate(str.maketrans("ACGT", "TGCA")) * 5]
    seqs += ["ACGTA", ""]
    engine.set_reads(seqs)
    codes, o = orc.digitize(seqs)
    # --minseqlength 32 (vsearch's default for the clustering commands) and 1 (--fastx_uniques: SeqSample.deduplicate's value)
    for minlen, ndropped in ((32, 4), (1, 1)):
        nu = engine.derep(minseqlength=minlen)
        rep_of, strand, _ = engine.get_derep()
        nc, orep, ostrand = orc.derep(codes, o, minlen=minlen)
        assert nu == nc
        assert np.array_equal(rep_of, orep)
        assert np.array_equal(strand, ostrand)
        assert (strand == -1).sum() > 100 and (rep_of == -1).sum() == ndropped
        st = engine.stats()
        assert st["hash_reseeds"] == 0


def test_derep_forward_only_and_ragged_lengths(engine):
    rng = np.random.default_rng(3)
    base = ["".join(rng.choice(list("ACGT"), int(L))) for L in rng.integers(32, 700, 300)]
    seqs = [base[i] for i in rng.integers(0, 300, 2000)]
    engine.set_reads(seqs)
    for both in (True, False):
        nu = engine.derep(strand_both=both)
        rep_of, strand, _ = engine.get_derep()
        codes, o = orc.digitize(seqs)
        nc, orep, ostrand = orc.derep(codes, o, strand_both=both)
        assert nu == nc and np.array_equal(rep_of, orep) and np.array_equal(strand, ostrand)


# --------------------------------------------------------------------------------------------
def _run_both(engine, hmm_text, seqs, threads=8, **flags):
    engine.load_profiles(text=hmm_text)
    engine.set_reads(seqs)
    engine.derep()
    engine.search(**flags)
    engine.finalize()
    seed, _ = engine.get_uniques()
    useqs = [seqs[int(i)] for i in seed]
    codes, o = orc.digitize(useqs)
    res = orc.SearchResult(orc.HmmSet(text=hmm_text), codes, o, threads=threads, keep_trace=1, **flags)
    return res


def _compare(engine, res, left="3_", right="4_"):
    tr = engine.pairtraces()
    ot = res.trace
    assert len(tr) == len(ot), (len(tr), len(ot))
    assert np.array_equal(tr["rep"], ot["seq"]) and np.array_equal(tr["prof"], ot["prof"])
    assert np.array_equal(tr["msv_xj"], ot["msv_xj"])
    for f in ("msv_sc", "nullsc", "filtersc"):
        assert np.array_equal(_bits(tr[f]), _bits(ot[f])), f
    assert np.array_equal(tr["pass_bias"], ot["pass_bias"])
    assert np.array_equal(tr["ran_vit"], ot["ran_vit"]) and np.array_equal(tr["pass_vit"], ot["pass_vit"])
    rv = ot["ran_vit"] == 1
    assert np.array_equal(_bits(tr["vitsc"][rv]), _bits(ot["vitsc"][rv]))
    pb = ot["pass_vit"] == 1
    assert np.array_equal(_bits(tr["fwdsc"][pb]), _bits(ot["fwdsc"][pb]))
    assert np.array_equal(tr["pass_fwd"], ot["pass_fwd"])
    pf = ot["pass_fwd"] == 1
    assert np.array_equal(_bits(tr["bcksc"][pf]), _bits(ot["bcksc"][pf]))
    assert np.array_equal(tr["nregions"][pf], ot["nregions"][pf])
    bad = np.flatnonzero((tr["ndom"] != ot["ndom"]) & pf)
    assert len(bad) == 0, [(int(tr["rep"][i]), int(tr["prof"][i]), int(tr["ndom"][i]), int(ot["ndom"][i]), int(tr["nregions"][i])) for i in bad[:8]]
    d, od = engine.domains(), res.domains
    assert len(d) == len(od)
    for f in ("rep", "prof", "tlen", "ienv", "jenv", "dom_idx", "ndom", "seq_reported", "dom_reported"):
        assert np.array_equal(d[f], od["seq" if f == "rep" else f]), f
    assert np.array_equal(d["flags"] & 1, od["flags"] & 1)
    for f in ("envsc", "domcorrection", "dombias", "bitscore", "seq_score", "seq_bias"):
        assert np.array_equal(_bits(d[f]), _bits(od[f])), f
    assert np.array_equal(d["lnP"].view(np.uint64), od["lnP"].view(np.uint64))
    got = engine.rep_coords(left, right)
    exp = res.positions(left, right)
    for g, e in zip(got, exp):
        assert np.array_equal(g, e)
    c = res.counts
    s = engine.stats()
    assert (s["n_past_msv"], s["n_past_bias"], s["n_past_fwd"]) == (c["past_msv"], c["past_bias"], c["past_fwd"])
    assert s["n_multidomain"] == c["multidomain"]


def test_search_fixture_mini_profiles(engine, fixture_reads, mini_hmm_text):
    """a4-a6 on the reference's fixture reads, 10 profiles incl. the two short (M=25, 11) models."""
    names, seqs = fixture_reads
    res = _run_both(engine, mini_hmm_text, seqs)
    _compare(engine, res)
    _compare(engine, res, "1_", "2_")
    _compare(engine, res, "1_", "4_")


def test_search_synthetic_tracheophyta_its2(engine, t_hmm_text):
    """stand-in taxon (Tracheophyta), ITS2 profiles, synthetic reads incl. N and reverse complements."""
    blob, offs = synth.make_reads(t_hmm_text, 600, seed=21)
    seqs = synth.to_strings(blob, offs)
    hmm = _its2_subset(t_hmm_text)
    res = _run_both(engine, hmm, seqs)
    assert res.counts["past_fwd"] > 1000
    _compare(engine, res)


def test_search_ragged_lengths_and_edge_cases(engine, t_hmm_text):
    blob, offs = synth.make_reads(t_hmm_text, 300, seed=22, fixed_len=0, len_range=(120, 520))
    seqs = synth.to_strings(blob, offs)
    rng = np.random.default_rng(9)
    seqs += ["".join(rng.choice(list("ACGT"), 40)), "N" * 64, "ACGTRYKMSWBDHVN" * 8, seqs[0] + seqs[1]]
    hmm = _its2_subset(t_hmm_text, 20, 20)
    res = _run_both(engine, hmm, seqs)
    _compare(engine, res)


def test_search_mixed_lengths_and_degenerate_bases_across_tiles(engine, t_hmm_text):
    """k_msv takes 256 representatives per block, a lane each: lengths from 20 to 620 inside one wave, 2 % N (a per-lane
    exception list the lane steps through on its own), random reads that fail at different rows, reads with the motif
    twice -- several blocks' worth, every stage equal to the oracle (scripts/parity_stress.py is the 12 000-read version)."""
    blob, offs = synth.make_reads(t_hmm_text, 2500, config=3, seed=synth.SEED + 11, fixed_len=0, len_range=(300, 580), n_rate=0.02, sub_rate=0.01)
    rng = np.random.default_rng(11)
    seqs = []
    for s in synth.to_strings(blob, offs):
        r = rng.random()
        if r < 0.35:
            k = int(rng.integers(20, len(s)))
            s = s[:k] if rng.random() < 0.5 else s[-k:]
        elif r < 0.45:
            s = "".join(rng.choice(list("ACGTNRYKMSWBDHV"), size=int(rng.integers(20, 620))))
        elif r < 0.50:
            s = s + s[: int(rng.integers(10, 200))]
        seqs.append(s)
    res = _run_both(engine, _its2_subset(t_hmm_text, 30, 30), seqs)
    assert engine.n_unique > 2000 and res.counts["past_msv"] > 20000
    _compare(engine, res)


def test_search_fuzz_odd_reads(engine, mini_hmm_text, monkeypatch):
    """Reads nobody sequences: shorter than the models, homopolymers, consensus repeated many times (several domains per
    read, strong scores that force rescaling), IUPAC soup, motif fragments glued in both orientations.  Every stage must
    still equal the oracle bit for bit."""
    monkeypatch.setenv("ITSX_KEEP_TRACE", "1")
    rng = np.random.default_rng(77)
    cons3 = synth.consensus_motifs(mini_hmm_text, "3_")
    cons4 = synth.consensus_motifs(mini_hmm_text, "4_")
    rc = lambda s: s[::-1].translate(str.maketrans("ACGTN", "TGCAN"))
    rnd = lambda n: "".join(rng.choice(list("ACGT"), n))
    seqs = []
    for n in (33, 34, 40, 44, 45, 46, 60):
        seqs.append(rnd(n))
    seqs += ["A" * 200, "AC" * 150, "G" * 33, cons3[0] * 7, (cons3[0] + rnd(30) + cons4[0]) * 3, cons4[0][:20] + cons3[0] + cons4[0][20:]]
    seqs += [rnd(50) + cons3[0] + rnd(120) + cons4[0] + rnd(40) + cons3[1] + rnd(100) + cons4[1] + rnd(30)]
    seqs += [rc(cons3[0]) + rnd(80) + rc(cons4[0]), cons3[0][5:40] + rnd(10) + cons4[0][3:], "ACGTRYKMSWBDHVN" * 20 + cons3[0]]
    for _ in range(40):
        parts = []
        for _ in range(int(rng.integers(1, 6))):
            k = int(rng.integers(0, 5))
            parts.append([rnd(int(rng.integers(1, 90))), cons3[int(rng.integers(0, len(cons3)))], cons4[int(rng.integers(0, len(cons4)))],
                          "N" * int(rng.integers(1, 12)), rc(cons4[0])][k])
        s = "".join(parts)
        if len(s) >= 33:
            s = list(s)
            for _ in range(int(rng.integers(0, 6))):
                s[int(rng.integers(0, len(s)))] = str(rng.choice(list("ACGTNRY")))
            seqs.append("".join(s))
    res = _run_both(engine, mini_hmm_text, seqs)
    assert res.counts["past_fwd"] > 40 and (res.domains["ndom"] > 1).any()
    _compare(engine, res)
    _compare(engine, res, "1_", "2_")
    assert engine.stats()["n_domain_overflow"] == 0
    # (the limits -- 8 envelopes per (representative, profile), the ensemble's bookkeeping -- are refused, not capped: tests/test_gpu_caps.py)


def test_trim_coords_per_read_follow_matchdict(engine, fixture_reads, mini_hmm_text):
    names, seqs = fixture_reads
    res = _run_both(engine, mini_hmm_text, seqs)
    start, stop, tlen, ind = engine.trim_coords("3_", "4_")
    _, _, uniq_of = engine.get_derep()
    us, ue, ut, ui = res.positions("3_", "4_")
    assert np.array_equal(start, us[uniq_of]) and np.array_equal(stop, ue[uniq_of])
    assert np.array_equal(tlen, ut[uniq_of]) and np.array_equal(ind, ui[uniq_of])


def test_domz_override_changes_only_domain_reporting(engine, fixture_reads, mini_hmm_text):
    """multi-GPU hook: the all-reduced domZ is applied by itsx_search_finalize."""
    names, seqs = fixture_reads
    res = _run_both(engine, mini_hmm_text, seqs)
    z = engine.get_domz()
    big = z * 10 ** 9
    engine.set_domz(big)
    engine.finalize()
    d = engine.domains()
    res2 = orc.SearchResult(res.hs, *res._keep, threads=4, domZ=big)
    assert np.array_equal(d["dom_reported"], res2.domains["dom_reported"])
    assert d["dom_reported"].sum() < res.domains["dom_reported"].sum()


def test_file_compatible_outputs_feed_the_reference_parsers(engine, fixture_reads, mini_hmm_text, tmp_path):
    """SeqSample.deduplicate/_search write uc.txt / rep.fa / domtbl.txt that the mirror parsers read back
    into the same dicts the array fast path gives."""
    from itsxpress_amd import Dedup, ItsPosition, SeqSampleNotPaired
    names, seqs = fixture_reads
    fq = tmp_path / "seq.fq"
    with open(fq, "w") as f:
        for n, s in zip(names, seqs):
            f.write("@%s\n%s\n+\n%s\n" % (n, s, "I" * len(s)))
    hmm = tmp_path / "mini.hmm"
    hmm.write_text(mini_hmm_text)
    s = SeqSampleNotPaired(str(fq), str(tmp_path))
    s.deduplicate(threads=1)
    s._search(hmmfile=str(hmm), threads=1)
    assert os.path.exists(s.uc_file) and os.path.exists(s.rep_file) and os.path.exists(s.dom_file)
    dd = Dedup(s.uc_file, s.rep_file, s.seq_file)
    assert len(dd.matchdict) == 227
    ip = ItsPosition(s.dom_file, "ITS2")
    seed, _ = s.engine.get_uniques()
    unames = [names[int(i)] for i in seed]
    ip2 = ItsPosition.from_engine(s.engine, "ITS2", unames)
    assert ip.ddict == ip2.ddict
    start, stop, tlen, ind = s.trim_coordinates("ITS2")
    for i, nm in enumerate(names):
        rep = dd.matchdict[nm]
        if rep in ip.ddict:
            a, b, t = ip.get_position(rep)
            assert (a if a is not None else -1, b if b is not None else -1, t if t is not None else -1) == \
                (int(start[i]), int(stop[i]), int(tlen[i]))
        else:
            assert ind[i] == 0


def test_search_long_reads(engine, mini_hmm_text, monkeypatch):
    """PacBio-length targets (1.2-2.5 kb): slab rows, per-length tables and the MSV thresholds far from the 300-bp case"""
    monkeypatch.setenv("ITSX_KEEP_TRACE", "1")
    blob, offs = synth.make_reads(mini_hmm_text, 60, seed=31, fixed_len=0, len_range=(1200, 2500))
    res = _run_both(engine, mini_hmm_text, synth.to_strings(blob, offs))
    _compare(engine, res)


def test_search_all_taxa_its2(engine, fixture_reads, all_its2_hmm_text, monkeypatch):
    """BASELINE configs[3] profile set: --taxa All --region ITS2 = 814 profiles (13 groups of 64 lanes), every stage
    bit-identical to the oracle on the reference's fixture reads."""
    monkeypatch.setenv("ITSX_KEEP_TRACE", "1")
    names, seqs = fixture_reads
    res = _run_both(engine, all_its2_hmm_text, seqs[:90], threads=64)
    assert engine.stats()["n_profiles"] == 814
    _compare(engine, res)


def test_errors_are_loud(engine, tmp_path):
    from itsxpress_amd import EngineError
    with pytest.raises(EngineError):
        engine.load_profiles(text="HMMER3/f\nNAME x\nLENG 3\n")
    with pytest.raises(EngineError):
        engine.set_reads(["ACGT!ACGT" * 5])
    with pytest.raises(FileNotFoundError):
        engine.load_profiles(path=str(tmp_path / "nope.hmm"))
    engine.set_reads(["ACGT" * 10])
    with pytest.raises(EngineError):
        engine.cluster(1.5)                      # the identity must lie in (0, 1]
    with pytest.raises(EngineError):
        engine.get_cluster()                     # no clustering has run
    engine.derep()
    engine.search(F1=1e-6, F2=1e-3)          # F2 above F1: the Viterbi filter simply never runs (refused before round 2)
    assert engine.stats()["ms_vit_kernel"] == 0


def test_tiny_slab_budget_only_adds_batches(engine, t_hmm_text, monkeypatch):
    """The DP slabs are cut into batches by an HBM budget; the smallest budget the engine accepts (0.25 GB, several batches
    here instead of one) must not change a bit."""
    monkeypatch.setenv("ITSX_SLAB_GB", "0.25")
    monkeypatch.setenv("ITSX_KEEP_TRACE", "1")
    blob, offs = synth.make_reads(t_hmm_text, 2500, seed=29, fixed_len=0, len_range=(150, 400))
    res = _run_both(engine, _its2_subset(t_hmm_text, 30, 30), synth.to_strings(blob, offs), threads=64)
    assert engine.stats()["n_batches"] >= 3
    _compare(engine, res)


def test_chunked_search_is_identical_to_one_chunk(engine, t_hmm_text, monkeypatch):
    """Large inputs are searched in chunks of unique reads; a forced tiny chunk size must not change anything
    (domZ, thresholds and the argmax span chunks)."""
    blob, offs = synth.make_reads(t_hmm_text, 500, seed=23, fixed_len=0, len_range=(200, 420))
    seqs = synth.to_strings(blob, offs)
    hmm = _its2_subset(t_hmm_text, 25, 25)
    monkeypatch.setenv("ITSX_CHUNK_UNIQUES", "37")
    monkeypatch.setenv("ITSX_KEEP_TRACE", "1")
    res = _run_both(engine, hmm, seqs)
    _compare(engine, res)
    start, stop, tlen, ind = engine.trim_coords("3_", "4_")
    _, _, uniq_of = engine.get_derep()
    us, ue, ut, ui = res.positions("3_", "4_")
    assert np.array_equal(start, us[uniq_of]) and np.array_equal(stop, ue[uniq_of])


@pytest.mark.parametrize("layout", ["one_per_wave", "two_per_wave", "seven_per_wave", "sixty_four_per_wave"])
def test_multidomain_regions_are_resolved_by_traceback_clustering(engine, t_hmm_text, monkeypatch, layout):
    """hmmsearch resolves a region whose posterior suggests several domains by 200 stochastic tracebacks + clustering
    (region_trace_ensemble).  Engine (k_ensemble.hip) and oracle must agree on every envelope, every per-residue null2
    sum (domcorrection, seq_bias) and every score -- the random stream included -- on a workload with hundreds of such
    regions; and with the stage switched off on both sides they must agree on the old one-envelope behaviour.
    The kernel lays a batch out in four ways (a small batch one region per wave; a lazy round's few thousand regions 2 - 8 per wave, the
    other lanes helping; a full table's 64 per wave): every layout must give the same bits."""
    if layout != "one_per_wave":
        monkeypatch.setenv("ITSX_MR_ONE_MAX", "0")
        monkeypatch.setenv("ITSX_MR_WAVES", {"two_per_wave": "4096", "seven_per_wave": "128", "sixty_four_per_wave": "0"}[layout])
    blob, offs = synth.make_reads(t_hmm_text, 3000, seed=5, fixed_len=0, len_range=(300, 580))
    seqs = synth.to_strings(blob, offs)
    hmm = _its2_subset(t_hmm_text)
    res = _run_both(engine, hmm, seqs, threads=os.cpu_count() or 8)
    _compare(engine, res)
    if layout != "one_per_wave":
        return
    st = engine.stats()
    assert res.counts["multidomain"] > 300
    assert st["n_mr_clustered"] == res.counts["multidomain"] and st["n_mr_failed"] == 0
    flagged = res.domains[(res.domains["flags"] & 1) == 1]
    assert st["n_mr_envelopes"] == len(flagged) > 300
    # a clustered region may yield several envelopes, or a narrower one than the region
    d = engine.domains()
    assert len(d) == len(res.domains)
    monkeypatch.setenv("ITSX_NO_ENSEMBLE", "1")
    monkeypatch.setenv("ORC_NO_ENSEMBLE", "1")
    res0 = _run_both(engine, hmm, seqs, threads=os.cpu_count() or 8)
    _compare(engine, res0)
    assert engine.stats()["n_mr_clustered"] == 0 and res0.counts["multidomain"] == res.counts["multidomain"]


def test_viterbi_filter_with_hmmsearch_default_thresholds(engine, t_hmm_text, mini_hmm_text, fixture_reads):
    """--F1 0.02 --F2 1e-3 --F3 1e-5 (hmmsearch's own defaults; the reference passes 1e-6 three times, which skips this filter):
    the 16-bit Viterbi filter runs between the bias filter and Forward for every pair above F2, and every later stage sees only
    its survivors.  Engine == oracle on the filter's score bits, its verdicts and everything downstream."""
    flags = dict(F1=0.02, F2=1e-3, F3=1e-5)
    blob, offs = synth.make_reads(t_hmm_text, 500, seed=61, fixed_len=0, len_range=(150, 420))
    seqs = synth.to_strings(blob, offs) + ["ACGTRYKMSWBDHVN" * 10, "A" * 64]
    res = _run_both(engine, _its2_subset(t_hmm_text, 25, 25), seqs, **flags)
    ran = res.trace["ran_vit"] == 1
    assert ran.sum() > 500 and (res.trace["pass_vit"][ran] == 0).sum() > 50 and (res.trace["pass_vit"][ran] == 1).sum() > 50
    _compare(engine, res)
    assert engine.stats()["ms_vit_kernel"] > 0
    names, fseqs = fixture_reads
    res = _run_both(engine, mini_hmm_text, fseqs, **flags)          # incl. the short (M = 25, 11) models
    assert (res.trace["ran_vit"] == 1).sum() > 100
    _compare(engine, res)
    _compare(engine, res, "1_", "2_")
    # the reference's flags: the filter never runs
    res = _run_both(engine, mini_hmm_text, fseqs)
    assert (res.trace["ran_vit"] == 0).all() and engine.stats()["ms_vit_kernel"] == 0
    _compare(engine, res)

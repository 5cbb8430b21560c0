"""The engine's loud limits, at the edge: the longest model the kernels hold (46 nodes), the longest read (65 535 bases), a DP
slab budget the device cannot supply, an output device that is full."""
import numpy as np
import pytest

import orc
import synth
from test_gpu_parity import _compare, _run_both

pytestmark = pytest.mark.gpu


def _stretch(block, M_new):
    """a profile of M_new nodes made from a 45-node HMMER3/f block: the last node is repeated (the repeats take an inner
    node's transitions, the final node keeps the terminal ones)"""
    lines = block.split("\n")
    i_leng = next(k for k, ln in enumerate(lines) if ln.startswith("LENG"))
    M = int(lines[i_leng].split()[1])
    lines[i_leng] = "LENG  %d" % M_new
    i_hmm = next(k for k, ln in enumerate(lines) if ln.startswith("HMM "))
    first = i_hmm + 5                         # HMM, header, COMPO, node-0 inserts, node-0 transitions
    node = lambda k: lines[first + 3 * (k - 1): first + 3 * k]
    inner_t = node(M - 1)[2]
    out = lines[:first + 3 * (M - 1)]
    last = node(M)
    for k in range(M, M_new + 1):
        m = last[0].split()
        m[0] = str(k)
        out += ["  " + "  ".join(m), last[1], inner_t if k < M_new else last[2]]
    out += lines[first + 3 * M:]
    return "\n".join(out).replace("NAME  ", "NAME  ", 1)


def test_model_of_46_nodes_runs_and_47_is_refused(engine, mini_hmm_text):
    from itsxpress_amd import EngineError
    blocks = [b + "//\n" for b in mini_hmm_text.split("//\n") if "NAME  " in b]
    b45 = next(b for b in blocks if "LENG  45" in b)
    hmm46 = _stretch(b45, 46) + blocks[1]
    assert orc.HmmSet(text=hmm46).M[0] == 46
    blob, offs = synth.make_reads(mini_hmm_text, 400, seed=3)
    seqs = synth.to_strings(blob, offs)
    res = _run_both(engine, hmm46, seqs)
    assert res.counts["past_fwd"] > 50
    _compare(engine, res)
    with pytest.raises(EngineError, match="more than 46 nodes") as ei:
        engine.load_profiles(text=_stretch(b45, 47))
    assert ei.value.code == -5


def test_longest_read_is_65535_bases(engine, mini_hmm_text):
    from itsxpress_amd import EngineError
    rng = np.random.default_rng(1)
    cons = synth.consensus_motifs(mini_hmm_text, "3_")[0]
    long_read = "".join(rng.choice(list("ACGT"), 65535 - len(cons))) + cons
    seqs = [long_read, long_read[:40000], cons * 3]
    res = _run_both(engine, mini_hmm_text, seqs)
    _compare(engine, res)
    with pytest.raises(EngineError, match="65535") as ei:
        engine.set_reads([long_read + "A"])
    assert ei.value.code == -5


def test_slab_budget_beyond_the_device_shrinks_instead_of_failing(engine, t_hmm_text, monkeypatch):
    """ITSX_SLAB_GB above the free HBM: the budget is halved until the allocation succeeds, the answers do not change"""
    blob, offs = synth.make_reads(t_hmm_text, 400000, seed=9)
    blocks = [b + "//\n" for b in t_hmm_text.split("//\n") if "NAME  " in b]
    hmm = "".join(b for b in blocks if b.split("NAME  ")[1][:2] in ("3_", "4_"))

    def run():
        from itsxpress_amd import Engine
        e = Engine(0)                                         # a context of its own: the slab it ends up with is its own
        e.load_profiles(text=hmm)
        e.set_reads_buffer(blob, offs)
        e.derep(); e.search(); e.finalize()
        out = np.stack(e.trim_coords("3_", "4_"), axis=1), e.stats()
        e.close()
        return out

    c0, s0 = run()
    monkeypatch.setenv("ITSX_SLAB_GB", "400")
    c1, s1 = run()
    assert np.array_equal(c0, c1) and s1["n_slab_shrinks"] >= 1 and s0["n_slab_shrinks"] == 0
    assert s1["n_domains"] == s0["n_domains"] > 0


def test_full_output_device_is_an_error_not_a_short_file(engine, fixture_reads):
    from itsxpress_amd import EngineError
    names, seqs = fixture_reads
    engine.set_reads(seqs, names)
    engine.derep()
    for writer in (engine.write_uc, engine.write_rep_fasta):
        with pytest.raises(EngineError) as ei:
            writer("/dev/full")
        assert ei.value.code == -2

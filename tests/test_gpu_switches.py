"""The library's environment switches (csrc/switches.cpp).  The reference passes fixed flags to its engines (itsxpress/SeqSample.py:191-209)
and has nothing a stray environment variable could change; here a switch that changes results is a TEST HOOK and is honoured only
under ITSX_TEST_HOOKS=1.  This test sets every hook WITHOUT the gate and must get the default coordinates and counters; with the gate
the same settings do change the run (so the test tests something), and `Engine.switches()` reports what a search saw.  `pytest -m gpu`."""
import numpy as np
import pytest

from test_gpu_compact import PAIRS, _same
from test_gpu_parity import _its2_subset
from test_gpu_share import _bench_reads

pytestmark = pytest.mark.gpu

HOOKS = {"ITSX_NO_ENSEMBLE": "1", "ITSX_LAZY_ZUB_SCALE": "500000", "ITSX_LAZY_FORCE_PENDING": "3", "ITSX_LAZY_NO_RERUN": "1", "ITSX_LAZY_NO_COMPLETE": "1",
         "ITSX_COMPACT_ZMAX": "1", "ITSX_COMPACT_DOME_MIN": "10", "ITSX_PASSA_DBG": "1"}
KEYS = ("n_past_msv", "n_lazy_evaluated", "n_lazy_pending", "n_lazy_completed", "n_lazy_reruns", "n_domains", "n_mr_clustered")


def _run(engine, hmm, seqs):
    engine.set_rows_mode("lazy")
    try:
        engine.load_profiles(text=hmm)
        engine.set_reads(seqs)
        engine.derep()
        engine.search()
        engine.finalize()
        st = engine.stats()
        return [tuple(a.copy() for a in engine.trim_coords(l, r)) for l, r in PAIRS], {k: st[k] for k in KEYS}, engine.switches()
    finally:
        engine.set_rows_mode(None)


def test_result_changing_hooks_need_the_gate(engine, t_hmm_text, monkeypatch):
    hmm = _its2_subset(t_hmm_text, 20, 20)
    # multidomain targets among the reads, so that ITSX_NO_ENSEMBLE has something to change
    seqs = _bench_reads(t_hmm_text, 6000, seed=9)
    seqs += [s + s[40:] for s in seqs[:40]]
    for k in list(HOOKS) + ["ITSX_TEST_HOOKS"]:
        monkeypatch.delenv(k, raising=False)
    ref, st0, sw0 = _run(engine, hmm, seqs)
    assert not any(k in sw0 for k in HOOKS)
    # every hook at once, no gate: ignored, and reported as ignored
    for k, v in HOOKS.items():
        monkeypatch.setenv(k, v)
    got, st1, sw1 = _run(engine, hmm, seqs)
    assert _same(ref, got) and st1 == st0
    assert all(k in sw1 and "ignored" in sw1[k] for k in HOOKS), sw1
    # one by one as well
    for k in HOOKS:
        monkeypatch.delenv(k)
    for k, v in HOOKS.items():
        monkeypatch.setenv(k, v)
        got, st1, sw1 = _run(engine, hmm, seqs)
        assert _same(ref, got) and st1 == st0, k
        assert "ignored" in sw1[k]
        monkeypatch.delenv(k)
    # with the gate the hooks are live: forced-pending rows make the completion run
    monkeypatch.setenv("ITSX_TEST_HOOKS", "1")
    monkeypatch.setenv("ITSX_LAZY_FORCE_PENDING", "3")
    got, st2, sw2 = _run(engine, hmm, seqs)
    assert sw2["ITSX_LAZY_FORCE_PENDING"] == "3" and sw2["ITSX_TEST_HOOKS"] == "1"
    assert st2["n_lazy_reruns"] == 1 and st0["n_lazy_reruns"] == 0
    assert _same(ref, got)                                   # (the safety net is exact: the hook only forces it to run)
    monkeypatch.delenv("ITSX_LAZY_FORCE_PENDING")
    monkeypatch.setenv("ITSX_NO_ENSEMBLE", "1")
    _, st3, _ = _run(engine, hmm, seqs)
    assert st3["n_mr_clustered"] == 0 and st0["n_mr_clustered"] > 0

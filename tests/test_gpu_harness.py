"""scripts/parity_harness.py (the tool that pins the engine against real vsearch / hmmsearch where they exist) in its
--reference-dir mode: reference outputs identical to the engine's give 100 % concordance and exit status 0; a
changed envelope in the reference's domtbl is reported and fails."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_harness_reference_dir_mode(gold, tmp_path):
    from itsxpress_amd.SeqSample import SeqSampleNotPaired
    fq = os.path.join(gold, "seq.fq.gz")
    hmm = os.path.join(gold, "mini.hmm")
    ref = tmp_path / "ref"
    ref.mkdir()
    s = SeqSampleNotPaired(fastq=fq, tempdir=str(ref))
    s.deduplicate(threads=1)
    s._search(hmmfile=hmm, threads=1)
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "parity_harness.py"), "--fastq", fq, "--hmm", hmm, "--region", "ITS2",
           "--reference-dir", str(ref)]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    rep = json.loads(p.stdout)
    assert rep["reads"] == 227 and rep["reads_agreeing"] == 227 and rep["same_representative"] == 227
    assert rep["domain_rows_identical"] == rep["domain_keys"] > 50
    # move one right-hand envelope start in the "reference": the harness must notice
    lines = open(ref / "domtbl.txt").read().split("\n")
    k = next(i for i, ln in enumerate(lines) if ln and not ln.startswith("#") and ln.split()[3].startswith("4_"))
    ll = lines[k].split()
    ll[13] = "999.9"
    ll[19] = str(int(ll[19]) + 7)
    lines[k] = " ".join(ll)
    open(ref / "domtbl.txt", "w").write("\n".join(lines))
    p = subprocess.run(cmd, capture_output=True, text=True)
    rep = json.loads(p.stdout)
    assert p.returncode == 1 and rep["reads_agreeing"] < 227 and rep["first_differences"]
    # without the binaries --run-tools says so (FileNotFoundError, the reference's own error class for a missing engine)
    p = subprocess.run(cmd[:-2] + ["--run-tools"], capture_output=True, text=True)
    import shutil
    if shutil.which("vsearch") is None or shutil.which("hmmsearch") is None:
        assert p.returncode != 0 and "not on PATH" in p.stderr

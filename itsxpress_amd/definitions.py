"""Constants of the path's Python surface (mirror of itsxpress/definitions.py:32-82).

Values are the reference's (same keys, same file names, same order -- the order of
`taxa_dict` is the order in which `create_runtime_hmm("All", ...)` concatenates taxon files,
which decides ties between equal-scoring profiles).  ROOT_DIR points at the directory that
holds `ITSx_db/HMMs`: $ITSXPRESS_DB_DIR if set, else an installed `itsxpress` package, else
this package's own directory.
"""
import importlib.util
import os
from typing import Dict, List


def _find_root() -> str:
    env = os.environ.get("ITSXPRESS_DB_DIR")
    if env:
        # accept either .../ITSx_db/HMMs, .../ITSx_db or the package root
        p = os.path.abspath(env)
        if os.path.basename(p) == "HMMs":
            return os.path.dirname(os.path.dirname(p))
        if os.path.basename(p) == "ITSx_db":
            return os.path.dirname(p)
        return p
    try:
        spec = importlib.util.find_spec("itsxpress")
        if spec and spec.submodule_search_locations:
            return list(spec.submodule_search_locations)[0]
    except (ImportError, ValueError):
        pass
    return os.path.dirname(os.path.abspath(__file__))


ROOT_DIR: str = _find_root()


def hmm_path(taxon: str = "Fungi"):
    """Path of a taxon's profile file (`ITSx_db/HMMs/<letter>.hmm` under $ITSXPRESS_DB_DIR, an installed `itsxpress`
    package or this package), or None when it is not there.  `F.hmm` -- the Fungi set every BASELINE config names -- is
    absent from the reference mount this engine was built against; tests/test_fungi_pin.py and bench.py switch to it the
    moment this returns a path.  Looked up at call time, so a test may set ITSXPRESS_DB_DIR first."""
    fn = taxa_dict.get(taxon, taxon)
    for root in (_find_root(), ROOT_DIR):
        p = os.path.join(root, "ITSx_db", "HMMs", fn)
        if os.path.isfile(p) and os.path.getsize(p) > 0:
            return p
    return None

_TAXA = [("Alveolata", "A"), ("Bryophyta", "B"), ("Bacillariophyta", "C"), ("Amoebozoa", "D"),
         ("Euglenozoa", "E"), ("Fungi", "F"), ("Chlorophyta", "G"), ("Rhodophyta", "H"),
         ("Phaeophyceae", "I"), ("Marchantiophyta", "L"), ("Metazoa", "M"), ("Oomycota", "O"),
         ("Haptophyceae", "P"), ("Raphidophyceae", "Q"), (" Rhizaria", "R"), ("Synurophyceae", "S"),
         ("Tracheophyta", "T"), ("Eustigmatophyceae", "U"), ("Parabasalia", "Y")]

taxa_choices: List[str] = [t for t, _ in _TAXA] + ["All"]
taxa_dict: Dict[str, str] = {t: letter + ".hmm" for t, letter in _TAXA}
taxa_dict["All"] = "all.hmm"

maxmismatches: int = 40
maxratio: float = 0.3
vsearch_fastq_qmax: int = 93

#!/usr/bin/env python3
"""The shared schedules against the unshared kernels on the bench workload itself (the review's "0 differences in >= 1e8
pairs" for the MSV filter's join): one lazy search with ITSX_SHARE_CHECK=1 -- every (representative, profile) cell of the MSV filter and
every pass-A score is computed a second time from row 1 by the unshared kernels and compared (MSV cells and the scores of chains
that run to their last row bit for bit; joined scores within 2e-3 nats).  usage: share_check.py [reads]  -> one JSON line"""
import gzip, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from itsxpress_amd import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
thmm = gzip.open(os.path.join(ROOT, "tests", "golden", "T.hmm.gz"), "rt").read()
blocks = [b + "//\n" for b in thmm.split("//\n") if "NAME  " in b]
hmm = "".join(b for b in blocks if b.split("NAME  ")[1][:2] in ("3_", "4_"))
blob, offs = synth.make_reads(thmm, n, config=3, fixed_len=0, len_range=(300, 580), as_array=True)
os.environ["ITSX_SHARE_CHECK"] = "1"
eng = Engine(0)
eng.load_profiles(text=hmm)
eng.set_reads_buffer(blob, offs)
eng.derep()
eng.set_rows_mode("lazy")
eng.search()
st = eng.stats()
print(json.dumps({"reads": n, "representatives": int(st["n_unique"]), "profiles": int(st["n_profiles"]), "msv_cells_compared": int(st["n_pairs"]),
                  "pass_a_scores_compared": int(st["n_past_msv"]), "joined_representatives": int(st["n_joined"]), "share_mismatch": int(st["share_mismatch"]),
                  "join_maxdiff_nats": float(st["join_maxdiff"]), "two_sided": int(st["two_sided"]), "switches": eng.switches()}))

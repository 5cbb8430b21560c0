#!/usr/bin/env python3
"""diagnostic: engine vs oracle on a workload with many multidomain regions; prints the pairs whose domains differ"""
import gzip, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import orc, synth
from itsxpress_amd import Engine
t = gzip.open(os.path.join(ROOT, "tests/golden/T.hmm.gz"), "rt").read()
blocks = [b + "//\n" for b in t.split("//\n") if "NAME  " in b]
hmm = "".join([b for b in blocks if b.split("NAME  ")[1].startswith("3_")] + [b for b in blocks if b.split("NAME  ")[1].startswith("4_")])
blob, offs = synth.make_reads(t, int(sys.argv[1]) if len(sys.argv) > 1 else 3000, seed=5, fixed_len=0, len_range=(300, 580))
seqs = synth.to_strings(blob, offs)
eng = Engine(0)
eng.load_profiles(text=hmm); eng.set_reads(seqs); eng.derep(); eng.search(); eng.finalize()
seed, _ = eng.get_uniques()
useqs = [seqs[int(i)] for i in seed]
codes, o = orc.digitize(useqs)
res = orc.SearchResult(orc.HmmSet(text=hmm), codes, o, threads=os.cpu_count() or 8, keep_trace=1)
st = eng.stats()
print({k: st[k] for k in ("n_multidomain", "n_mr_clustered", "n_mr_failed", "n_mr_envelopes", "ms_ensemble", "n_domain_overflow")}, res.counts)
d, od = eng.domains(), res.domains
print("domains", len(d), len(od))
key = lambda a, s: {}
from collections import defaultdict
E, O = defaultdict(list), defaultdict(list)
for r in d: E[(int(r["rep"]), int(r["prof"]))].append((int(r["ienv"]), int(r["jenv"]), int(r["flags"]) & 1, float(r["bitscore"]), float(r["domcorrection"]), int(r["dom_idx"])))
for r in od: O[(int(r["seq"]), int(r["prof"]))].append((int(r["ienv"]), int(r["jenv"]), int(r["flags"]) & 1, float(r["bitscore"]), float(r["domcorrection"]), int(r["dom_idx"])))
bad = [k for k in set(E) | set(O) if E.get(k) != O.get(k)]
print("pairs differing:", len(bad))
for k in sorted(bad)[:15]:
    print(k, "L", len(useqs[k[0]]), "\n   engine", E.get(k), "\n   oracle", O.get(k))
tr, ot = eng.pairtraces(), res.trace
pf = ot["pass_fwd"] == 1
dn = np.flatnonzero((tr["ndom"] != ot["ndom"]) & pf)
print("trace ndom differs for", len(dn), "pairs; nregions differs for", int(((tr["nregions"] != ot["nregions"]) & pf).sum()))
for i in dn[:10]:
    k = (int(tr["rep"][i]), int(tr["prof"][i]))
    print(k, "engine ndom", tr["ndom"][i], "nregions", tr["nregions"][i], "| oracle ndom", ot["ndom"][i], "nregions", ot["nregions"][i], "\n   engine rows", E.get(k), "\n   oracle rows", O.get(k))
da = eng.domains()


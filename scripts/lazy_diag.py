#!/usr/bin/env python3
"""Diagnostics of the lazy domain stage on the bench workload (GPU): pairs evaluated, undecided rows that matter and their profiles."""
import argparse, gzip, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
ap = argparse.ArgumentParser(); ap.add_argument("--reads", type=int, default=1000000); ap.add_argument("--exact-z", action="store_true")
args = ap.parse_args()
pass
import synth
from itsxpress_amd import Engine
with gzip.open(os.path.join(ROOT, "tests", "golden", "T.hmm.gz"), "rt") as f: thmm = f.read()
blocks = [b + "//\n" for b in thmm.split("//\n") if "NAME  " in b]
hmm = "".join(b for b in blocks if b.split("NAME  ")[1][:2] in ("3_", "4_"))
blob, offs = synth.make_reads(thmm, args.reads, config=3, seed=synth.SEED + 3, as_array=True, fixed_len=0, len_range=(300, 580))
e = Engine(0); e.load_profiles(text=hmm); e.set_reads_buffer(blob, offs); e.derep()
out = {}
if args.exact_z:
    e.set_rows_mode("compact"); e.search(); zex = e.get_domz().copy(); e.finalize(); ref = np.stack(e.trim_coords("3_", "4_"))
e.set_rows_mode("lazy")
t0 = time.time(); e.search(); dt = time.time() - t0
z = e.get_domz(); P = e.n_profiles
if args.exact_z:
    out["z_lb_over_exact"] = float(z[:P].sum() / zex.sum()); out["z_ub_over_exact"] = float(z[P:].sum() / zex.sum())
t1 = time.time(); e.finalize(); out["finalize_s"] = time.time() - t1
st = e.stats()
out.update({k: st[k] for k in ("n_unique", "n_past_msv", "n_lazy_evaluated", "n_lazy_round1", "n_lazy_pending", "n_lazy_pending_profiles", "n_lazy_completed", "n_lazy_completed_profiles", "ms_lazy_complete", "n_lazy_reruns", "ms_bound_kernel", "ms_lazy_select", "ms_fwd_kernel", "ms_bwd_kernel", "ms_ensemble", "ms_msv")})
out["search_s"] = dt
if args.exact_z:
    e.search(); e.set_domz(np.concatenate([zex, zex])); e.finalize(); st2 = e.stats()
    out["pending_with_exact_z"] = st2["n_lazy_pending"]
    if st2["n_lazy_pending"] == 0:
        out["coords_equal_with_exact_z"] = bool(np.array_equal(ref, np.stack(e.trim_coords("3_", "4_"))))
print(json.dumps(out))

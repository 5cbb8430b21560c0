#!/usr/bin/env python3
"""Where does pass A's per-wave start-up go?  Runs the bench workload's search with the kernel's diagnostic bits (k_api.h: ShareLaunch::dbg;
honoured only under ITSX_TEST_HOOKS=1; the scores are garbage, only ms_bound_kernel is read): 1 = no rows, 2 = no state restore, 4 = no join.
usage: passa_dbg.py [reads]"""
import gzip, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from itsxpress_amd import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
thmm = gzip.open(os.path.join(ROOT, "tests", "golden", "T.hmm.gz"), "rt").read()
blocks = [b + "//\n" for b in thmm.split("//\n") if "NAME  " in b]
hmm = "".join(b for b in blocks if b.split("NAME  ")[1][:2] in ("3_", "4_"))
blob, offs = synth.make_reads(thmm, n, config=3, fixed_len=0, len_range=(300, 580), as_array=True)
eng = Engine(0)
eng.load_profiles(text=hmm)
eng.set_reads_buffer(blob, offs)
eng.derep()
eng.set_rows_mode("lazy")
os.environ["ITSX_TEST_HOOKS"] = "1"
for bits in (0, 1, 3, 5, 7, 0):
    os.environ["ITSX_PASSA_DBG"] = str(bits)
    try:
        eng.search()
    except Exception as e:
        print("bits", bits, "search raised", str(e)[:100])
    st = eng.stats()
    print(json.dumps({"dbg_bits": bits, "ms_bound_kernel": round(float(st["ms_bound_kernel"]), 1), "bound_launches": int(st["n_bound_launches"]), "bwd_launches": int(st["n_bwd_launches"])}), flush=True)

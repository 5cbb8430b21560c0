#!/usr/bin/env python3
"""Round 6 (two-sided sharing for pass A and the MSV filter): HBM traffic of this round's kernels from rocprofv3 PMC passes
(MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE in separate --pmc passes, counters in KiB, FETCH_SIZE doubled on gfx950 --
profiles/round2_fetch_calibration.md confirmed the factor on the slab's 4-byte-per-lane pattern).

On the GPU box (each counter its own run; the program itself after `--`):
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc6/fetch -- python3 $R/bench.py --reads N \\
      --steps 1 --warmup 0 --cpu-sample 0 --handover-steps 0 --full-steps 0 --alone-steps 0 > $R/gpurun_out/pmc6/fetch.json
  rocprofv3 --pmc WRITE_SIZE ... -d $R/gpurun_out/pmc6/write ... > $R/gpurun_out/pmc6/write.json
Then here:  scripts/pmc_round6.py gpurun_out/pmc6 TAG  ->  profiles/round6_pmc_bytes_per_row.json (what bench.py's roofline.traffic
multiplies by the rows of a launch; written for TAG = 1M) and profiles/round6_pmc_hbm_traffic_TAG.md (every kernel)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(d):
    out = {}
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(fn) as f:
            for row in csv.DictReader(f):
                name = row.get("Kernel_Name") or row.get("Kernel Name") or ""
                v = float(row.get("Counter_Value") or row.get("Counter Value") or 0)
                k = name.split("(")[0].replace("void ", "").replace("itsx::", "")
                e = out.setdefault(k, [0, 0.0])
                e[0] += 1
                e[1] += v
    return out


def main():
    d, tag = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "1M")
    line = json.loads([ln for ln in open(os.path.join(d, "fetch.json")) if ln.startswith("{")][0])
    fe, wr = per_kernel(os.path.join(d, "fetch")), per_kernel(os.path.join(d, "write"))
    names = sorted(set(fe) | set(wr), key=lambda k: -(2 * fe.get(k, [0, 0])[1] + wr.get(k, [0, 0])[1]))
    gb = lambda kib: kib * 1024.0 / 1e9
    k, v, cfgl = line["kernels"], line["valu"], line["config"]
    rows_fwd = v["fwd_rows_per_s"] * k["k_filters_fwd"]["ms"] * 1e-3 if v.get("fwd_rows_per_s") else None
    lr = cfgl["lane_rows"]
    per_row = {}

    def tot(kern, exact=False):
        c = [x for x in names if (x == kern or x.startswith(kern + "<")) or (not exact and x.startswith(kern))]
        return sum(2 * fe.get(x, [0, 0])[1] + wr.get(x, [0, 0])[1] for x in c) * 1024.0
    if rows_fwd:
        for kern in ("k_filters_fwd", "k_bwd_decode", "k_decode"):
            per_row[kern] = round(tot(kern) / rows_fwd, 3)
    if lr["bound_rows"]:
        per_row["k_fwd_bound"] = round(tot("k_fwd_bound") / lr["bound_rows"], 4)
    if lr["bwd_rows"]:
        per_row["k_bwd_bound"] = round(tot("k_bwd_bound") / lr["bwd_rows"], 4)
    # the MSV filter's lane-rows: its Forward chains' and Backward chains' together (the stats count both in msv_rows)
    per_row["k_msv"] = round((tot("k_msv", True) + tot("k_msv_bwd")) / max(1, lr["msv_rows"]), 4)
    per_row["_GB_fetched_plus_written"] = {kn: round(tot(kn, kn == "k_msv") / 1e9, 2) for kn in ("k_msv", "k_msv_bwd", "k_fwd_bound", "k_bwd_bound")}
    per_row["_collected_with"] = ("rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --kernel-trace --output-format csv -- python3 bench.py --reads %d --steps 1 --warmup 0 "
                                  "--cpu-sample 0 --handover-steps 0 --full-steps 0 --alone-steps 0; FETCH_SIZE x 2 (gfx950), KiB -> bytes; scripts/pmc_round6.py" % cfgl["reads_rank0"])
    per_row["_rows"] = {"slab kernels (pairs the lazy stage evaluates)": rows_fwd, **lr}
    if tag == "1M":
        with open(os.path.join(ROOT, "profiles", "round6_pmc_bytes_per_row.json"), "w") as f:
            json.dump(per_row, f, indent=1)
    with open(os.path.join(ROOT, "profiles", "round6_pmc_hbm_traffic_%s.md" % tag), "w") as f:
        f.write("# HBM-side traffic per kernel, round 6 (two-sided sharing: pass A and the MSV filter) -- rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes\n\n")
        f.write("Command (each counter its own run): `%s`\n" % per_row["_collected_with"])
        f.write("Workload: configs[2]'s shape at %d reads (%d unique, %d pairs past MSV, %d through the domain pipeline).\n"
                "Counters in KiB; FETCH_SIZE doubled (MI355X_MICROARCH.md; profiles/round2_fetch_calibration.md).\n\n" %
                (cfgl["reads_rank0"], cfgl["unique"], cfgl["pairs_past_msv"], cfgl["pairs_evaluated"]))
        f.write("Bytes per lane-row the kernel computed: %s\n\n" % json.dumps({a: b for a, b in per_row.items() if not a.startswith("_")}))
        f.write("| kernel | launches | read (GB, FETCH_SIZE x 2) | written (GB) |\n|---|---|---|---|\n")
        tr = tw = 0.0
        for kname in names:
            r, w = fe.get(kname, [0, 0.0]), wr.get(kname, [0, 0.0])
            tr += gb(2 * r[1]); tw += gb(w[1])
            if gb(2 * r[1]) + gb(w[1]) < 0.005:
                continue
            f.write("| `%s` | %d | %.2f | %.2f |\n" % (kname[:90], max(r[0], w[0]), gb(2 * r[1]), gb(w[1])))
        f.write("| **all kernels** | | **%.1f** | **%.1f** |\n" % (tr, tw))
        f.write("\nThe step's wall time in the counter runs is not the bench's (counter collection serialises the kernels); the bytes are what is read here.\n")
    print(json.dumps(per_row, indent=1))


if __name__ == "__main__":
    main()

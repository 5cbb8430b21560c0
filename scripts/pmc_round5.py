#!/usr/bin/env python3
"""Round 5 (prefix sharing, MSV words staged through LDS): HBM traffic of this round's kernels from rocprofv3 PMC passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE in separate --pmc
passes, counters in KiB, FETCH_SIZE doubled on gfx950 -- profiles/round2_fetch_calibration.md confirmed the factor on the slab's
4-byte-per-lane pattern).

On the GPU box (each counter its own run; the program itself after `--`):
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc/fetch -- python3 $R/bench.py --reads 1000000 \\
      --steps 1 --warmup 0 --cpu-sample 0 --handover-steps 0 --full-steps 0 > $R/gpurun_out/pmc/fetch.json
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc/write -- python3 $R/bench.py --reads 1000000 \\
      --steps 1 --warmup 0 --cpu-sample 0 --handover-steps 0 --full-steps 0 > $R/gpurun_out/pmc/write.json
Then here:  scripts/pmc_round5.py gpurun_out/pmc  ->  profiles/round5_pmc_bytes_per_row.json (what bench.py's roofline.traffic
multiplies by the rows of a launch) and profiles/round5_pmc_hbm_traffic_1M.md (every kernel)."""
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
    d = sys.argv[1]
    line = json.loads([ln for ln in open(os.path.join(d, "fetch.json")) if ln.startswith("{")][0])
    fe, wr = per_kernel(os.path.join(d, "fetch")), per_kernel(os.path.join(d, "write"))
    names = sorted(set(fe) | set(wr), key=lambda k: -(2 * fe.get(k, [0, 0])[1] + wr.get(k, [0, 0])[1]))
    gb = lambda kib: kib * 1024.0 / 1e9
    rows_bound = None
    # rows of the three slab kernels / of the bound pass: from the line's own counters (valu.* x the kernels' times)
    k = line["kernels"]
    v = line["valu"]
    rows_fwd = v["fwd_rows_per_s"] * k["k_filters_fwd"]["ms"] * 1e-3 if v.get("fwd_rows_per_s") else None
    bound_ms = k.get("k_fwd_bound", {}).get("ms")
    cfgl = line["config"]
    per_row = {}

    def tot(kern):
        c = [x for x in names if x.startswith(kern)]
        return sum(2 * fe.get(x, [0, 0])[1] + wr.get(x, [0, 0])[1] for x in c) * 1024.0
    if rows_fwd:
        for kern in ("k_filters_fwd", "k_bwd_decode", "k_decode"):
            per_row[kern] = round(tot(kern) / rows_fwd, 3)
    if line["roofline"]["kernel"] == "k_fwd_bound":
        rows_bound = line["roofline"]["alg_flops_per_launch"] * line["roofline"]["launches_per_step"] / line["roofline"]["alg_flops_per_lane_row"]
        per_row["k_fwd_bound"] = round(tot("k_fwd_bound") / rows_bound, 4)
    # k_msv: lane-rows it computed = (unique x profiles x length) x (1 - rows_shared_frac)
    sh = cfgl.get("rows_shared_frac") or {}
    rows_msv = cfgl["unique"] * cfgl["profiles"] * cfgl["mean_length"] * (1.0 - (sh.get("k_msv") or 0.0))
    per_row["k_msv"] = round(tot("k_msv") / rows_msv, 4)
    per_row["_k_msv_GB_fetched_plus_written"] = round(tot("k_msv") / 1e9, 2)
    per_row["_collected_with"] = "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --kernel-trace --output-format csv -- python3 bench.py --reads 1000000 --steps 1 --warmup 0 --cpu-sample 0 --handover-steps 0 --full-steps 0; FETCH_SIZE x 2 (gfx950), KiB -> bytes; scripts/pmc_round5.py"
    per_row["_rows"] = {"slab kernels (pairs the lazy stage evaluates)": rows_fwd, "k_fwd_bound (the chains' own rows)": rows_bound, "k_msv (the chains' own rows, approximate: mean length)": rows_msv}
    with open(os.path.join(ROOT, "profiles", "round5_pmc_bytes_per_row.json"), "w") as f:
        json.dump(per_row, f, indent=1)
    with open(os.path.join(ROOT, "profiles", "round5_pmc_hbm_traffic_1M.md"), "w") as f:
        f.write("# HBM-side traffic per kernel, round 5 (lazy domain stage, prefix sharing) -- rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes\n\n")
        f.write("Command (each counter its own run): `%s`\n" % per_row["_collected_with"])
        f.write("Workload: configs[2]'s shape at 1 M reads (%d unique, %d pairs past MSV, %d through the domain pipeline).\n"
                "Counters in KiB; FETCH_SIZE doubled (MI355X_MICROARCH.md; profiles/round2_fetch_calibration.md).\n\n" %
                (cfgl["unique"], cfgl["pairs_past_msv"], cfgl["pairs_evaluated"]))
        f.write("| kernel | launches | read (GB, FETCH_SIZE x 2) | written (GB) |\n|---|---|---|---|\n")
        for kname in names:
            r, w = fe.get(kname, [0, 0.0]), wr.get(kname, [0, 0.0])
            if gb(2 * r[1]) + gb(w[1]) < 0.005:
                continue
            f.write("| %s | %d | %.2f | %.2f |\n" % (kname, max(r[0], w[0]), gb(2 * r[1]), gb(w[1])))
        f.write("\nBytes per lane-row (profiles/round5_pmc_bytes_per_row.json): %s\n" % json.dumps({k2: v2 for k2, v2 in per_row.items() if not k2.startswith("_")}))
    print(json.dumps(per_row, indent=1))


if __name__ == "__main__":
    main()

"""Turn the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE CSVs of bench.py runs (1e9 rows, one step) into
profiles/r2_pmc_traffic_1e9.json: HBM bytes per launch of every hot-path kernel, = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
(FETCH_SIZE counts 32-B... units of KB on gfx950 and under-reports by 2x: MI355X_MICROARCH.md, HBM / rocprofv3 section).
Usage: pmc_to_json.py <dir with pmc_<workload>_{fetch,write}/...counter_collection.csv> <out.json>"""
import csv
import glob
import json
import os
import sys

GROUPS = {  # bench.py roofline["kernel"] prefix -> kernels whose traffic it sums (substring match)
    "partitioned probe": ["k_pj_hist", "k_pj_offsets", "k_pj_scatter", "k_pj_probe_pipe"],
    "k_part_hist + k_part_scatter + k_part_aggregate": ["k_part_hist", "k_part_offsets", "k_part_scatter", "k_part_aggregate"],
}
SINGLE = ["k_hy_hist", "k_msd_pass", "k_plan2", "k_local_sort", "k_pj_hist", "k_pj_scatter", "k_pj_probe_pipe", "k_pj_build",
          "k_part_hist", "k_part_scatter", "k_part_aggregate", "k_part_minmax", "k_lookback_scan", "k_stream_reduce"]


def read(root, workload, counter):
    """{kernel name: [value per dispatch, in dispatch order]} for one counter of one workload"""
    out = {}
    for f in glob.glob(os.path.join(root, f"pmc_{workload}_{counter}", "**", "*counter_collection.csv"), recursive=True):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
        for r in rows:
            if r["Counter_Name"].startswith(counter.upper()):
                out.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return out


def main(root, out_path):
    kernels = {}
    for wl in ("sort", "join", "groupby", "scan", "reduce", "groupby_minmax"):
        fetch, write = read(root, wl, "fetch_size"), read(root, wl, "write_size")
        if not fetch and not write:
            continue
        per = {}
        for name in set(fetch) | set(write):
            short = next((s for s in SINGLE if s in name), None)
            if short is None:
                continue
            f, w = fetch.get(name, []), write.get(name, [])
            m = min(len(f), len(w))
            # the two passes launch the same sequence: pair the dispatches, keep the LARGEST one (the 1e9-row launch;
            # the same kernel also runs on the 1e8-row build side, or as an early-exit no-op)
            pairs = [(2 * a + b, a, b) for a, b in zip(f[:m], w[:m])]
            if not pairs:
                continue
            tot, a, b = max(pairs)
            tag = short
            if short == "k_msd_pass":
                tag = "k_msd_pass level 1" if ", 9>" in name else "k_msd_pass level 0"
            if short == "k_hy_hist" and tot * 1024 < 1e9:
                continue
            e = {"FETCH_SIZE_KB": a, "WRITE_SIZE_KB": b, "hbm_bytes_per_launch": tot * 1024, "dispatches_seen": m, "workload": wl}
            if tag in per and per[tag]["hbm_bytes_per_launch"] >= e["hbm_bytes_per_launch"]:
                continue
            per[tag] = e
        for tag, e in per.items():
            kernels[tag if tag not in kernels else f"{tag} [{wl}]"] = e
        for prefix, members in GROUPS.items():
            have = [m for m in members if m in per]
            if len(have) >= 3:
                kernels[prefix] = {"hbm_bytes_per_launch": sum(per[m]["hbm_bytes_per_launch"] for m in have), "members": have,
                                   "workload": wl, "note": "one step = one 1e9-row launch of each member"}
    json.dump({"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (separate runs) of "
                         "python bench.py --workload <w> --rows 1e9 --steps 1 --warmup 0 (scripts/gpu_r2_run18.sh)",
               "rows": 1000000000, "correction": "hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024", "kernels": kernels},
              open(out_path, "w"), indent=1)
    for k, v in kernels.items():
        print(f"{k:60s} {v['hbm_bytes_per_launch'] / 1e9:9.2f} GB/launch")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

"""Turn the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE CSVs of bench.py runs (1e9 rows, one step) into
profiles/r<round>_pmc_traffic_1e9.json: HBM bytes per launch of every hot-path kernel, = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
(FETCH_SIZE is reported in KB on gfx950 and under-reports by 2x: MI355X_MICROARCH.md, HBM / rocprofv3 section).
Round 3: kernels are keyed by their base name (k_hf_scatter / k_msd_pass with their level), and the steps bench.py times are
summed into groups -- "sort local stage", "sort", "join probe phase", "groupby" -- whose names bench.py's roofline objects
carry as `traffic_key`.
Usage: pmc_to_json.py <dir with pmc_<workload>_{fetch_size,write_size}/...counter_collection.csv> <out.json> [source note]"""
import csv
import glob
import json
import os
import re
import sys

def csrc_sha16():
    """first 16 hex digits of the sha256 over the kernel sources (cudf_amd/csrc/*.hip, *.hpp, sorted by name): bench.py attaches a
    committed traffic figure to its roofline object only while the kernels it was measured on are the kernels it runs"""
    import hashlib
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cudf_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(root)):
        if f.endswith(".hip") or f.endswith(".hpp"):
            h.update(f.encode())
            h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()[:16]


GROUPS = {  # group -> (workload, base names whose LARGEST dispatch is summed)
    "sort local stage": ("sort", ["k_local_place", "k_local_sort"]),
    "sort": ("sort", ["k_hf_sample", "k_hf_plan", "k_hf_scatter level 0", "k_hf_scatter level 1", "k_hy_hist", "k_msd_pass level 0",
                      "k_msd_pass level 1", "k_plan2", "k_local_place", "k_local_sort"]),
    "join probe phase": ("join", ["k_pj2_scatter", "k_pj2_scatter_rec", "k_pj2_offsets", "k_pj2_probe_pipe", "k_pj2_probe_rare"]),  # (round 6: the record-form scatter)
    "join build": ("join", ["k_pj_hist", "k_pj_offsets", "k_pj_scatter", "k_bw_split", "k_bw_build", "k_bw_fixup"]),
    # round 6: the argsort is a keys-only word sort (gx_order.hip) -- its own kernels + the cursor path's; the round-3 pairs kernels stay listed
    # for runs with --sort-order-map 0 (base names that did not run contribute nothing)
    "sorted_order": ("sorted_order", ["k_om_sample", "k_om_plan", "k_om_count", "k_om_plan2", "k_om_map", "k_om_finish_a", "k_om_finish_b", "k_om_long",
                                      "k_hf_sample", "k_hf_plan", "k_hf_scatter level 0", "k_hf_scatter level 1",
                                      "k_hy_hist", "k_msd_pass level 0", "k_msd_pass level 1", "k_plan2", "k_local_place", "k_local_sort"]),
    "join probe phase (exact two-pass)": ("join", ["k_pj_hist", "k_pj_offsets", "k_pj_scatter", "k_pj_probe_pipe"]),
    # (round 6: the dense-id path's kernels -- k_dense_* -- run INSTEAD of k_part_aggregate where the ids allow; both listed)
    "groupby": ("groupby", ["k_slot_sample", "k_slot_plan", "k_part_reset_cursors", "k_part_scatter", "k_part_aggregate", "k_dense_sample", "k_dense_plan",
                            "k_dense_aggregate", "k_compact", "k_chunk_scan", "k_chunk_reduce"]),
    "groupby (exact two-pass)": ("groupby", ["k_part_hist", "k_part_offsets", "k_part_scatter", "k_part_aggregate"]),
    "groupby_minmax": ("groupby_minmax", ["k_slot_sample", "k_slot_plan", "k_part_reset_cursors", "k_part_scatter", "k_part_minmax"]),
}


def base_name(kernel):
    """gx::sort::k_hf_scatter<unsigned long, 1, 0, 8>(...) -> 'k_hf_scatter level 0'; None for kernels outside gx::"""
    m = re.search(r"\bgx::(?:\w+::)*(k_\w+)", kernel)
    if not m:
        return None
    name = m.group(1)
    targs = re.search(re.escape(name) + r"<([^()]*?)>\(", kernel)
    args = [a.strip() for a in targs.group(1).split(",")] if targs else []
    if name == "k_hf_scatter" and len(args) >= 3:
        return f"{name} level {args[2]}"
    if name == "k_msd_pass" and args:
        return f"{name} level {1 if args[-1] == '9' else 0}"  # level 1 of the 1e9-row sort is the 9-bit instantiation
    if name == "k_hf_sample" and args:
        return f"{name} {'histogram' if args[-1] == 'true' else 'masks'}"
    return name


def read(root, workload, counter):
    """{kernel name: [value per dispatch, in dispatch order]} for one counter of one workload"""
    out = {}
    for f in glob.glob(os.path.join(root, f"pmc_{workload}_{counter}", "**", "*counter_collection.csv"), recursive=True):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
        for r in rows:
            if r["Counter_Name"].startswith(counter.upper()):
                out.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return out


def main(root, out_path, note):
    kernels, per_wl = {}, {}
    for wl in ("sort", "sorted_order", "join", "groupby", "scan", "reduce", "groupby_minmax"):
        fetch, write = read(root, wl, "fetch_size"), read(root, wl, "write_size")
        if not fetch and not write:
            continue
        per = {}
        for name in set(fetch) | set(write):
            short = base_name(name)
            if short is None:
                continue
            f, w = fetch.get(name, []), write.get(name, [])
            m = min(len(f), len(w))
            # the two passes launch the same sequence: pair the dispatches, keep the LARGEST one (the 1e9-row launch; the
            # same kernel also runs on the 1e8-row build side, or as an early-exit no-op of a speculative branch)
            pairs = [(2 * a + b, a, b) for a, b in zip(f[:m], w[:m])]
            if not pairs:
                continue
            tot, a, b = max(pairs)
            e = {"FETCH_SIZE_KB": a, "WRITE_SIZE_KB": b, "hbm_bytes_per_launch": tot * 1024, "dispatches_seen": m, "workload": wl}
            if short in per and per[short]["hbm_bytes_per_launch"] >= e["hbm_bytes_per_launch"]:
                continue
            per[short] = e
        per_wl[wl] = per
        for short, e in per.items():
            if e["hbm_bytes_per_launch"] >= 1e6:
                kernels[short if short not in kernels else f"{short} [{wl}]"] = e
    groups = {}
    for g, (wl, members) in GROUPS.items():
        per = per_wl.get(wl, {})
        have = [m for m in members if any(k == m or k.startswith(m + " ") for k in per)]
        if not have:
            continue
        tot = sum(e["hbm_bytes_per_launch"] for k, e in per.items() if any(k == m or k.startswith(m + " ") for m in members))
        groups[g] = {"hbm_bytes_per_launch": tot, "members": have, "workload": wl,
                     "note": "one step = the largest launch of each member (speculative branches that did not run contribute ~0)"}
    json.dump({"source": note, "rows": 1000000000, "csrc_sha16": csrc_sha16(), "correction": "hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024",
               "kernels": kernels, "groups": groups}, open(out_path, "w"), indent=1)
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"]):
        print(f"{k:60s} {v['hbm_bytes_per_launch'] / 1e9:9.2f} GB/launch  [{v['workload']}]")
    for k, v in groups.items():
        print(f"GROUP {k:54s} {v['hbm_bytes_per_launch'] / 1e9:9.2f} GB/step   {v['members']}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else
         "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (separate runs) of python bench.py --workload <w> "
         "--rows 1e9 --steps 1 --warmup 0")

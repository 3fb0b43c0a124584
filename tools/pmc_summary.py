#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per kernel name, mean counter value per dispatch.
usage: python tools/pmc_summary.py gpurun_out/pmc_dir [more dirs...]"""
import csv, glob, os, sys, collections

def source_sha():
    """digest of the kernel sources the counters were collected on (ddp-generator_amd/evidence.py)"""
    import importlib.util
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("ilqg_evidence", os.path.join(here, "..", "ddp-generator_amd", "evidence.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.source_sha()


def load(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    return rows

def short(n):
    n = n.replace("(anonymous namespace)::", "")
    return n.split("(")[0]

KERNEL_LABELS = {  # (kernel, grid) -> bench.py's kernel label; rollout modes differ by grid size only
    "k_derivs": "k_derivs",
    "void k_backward<0>": "k_backward",
    "void k_backward<2>": "k_backward[fused derivs]",
    "void k_search<0, true>": "k_search[stage 1]",
    "void k_search<0, false>": "k_search[stage 1]",
    "void k_search<1, true>": "k_search[stage 2]",
    "void k_search<1, false>": "k_search[stage 2]",
}


def main():
    argv = sys.argv[1:]
    traffic_json = None
    config5_json, config5_iters = None, 1
    if argv and argv[0] == "--traffic-json":
        traffic_json, argv = argv[1], argv[2:]
    if argv and argv[0] == "--config5-json":  # --config5-json FILE ITERATIONS: per-iteration traffic of a bench.py --workload synth run
        config5_json, config5_iters, argv = argv[1], int(argv[2]), argv[3:]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in argv:
        per_dispatch = collections.defaultdict(dict)
        for r in load(d):
            key = (r["Dispatch_Id"], short(r["Kernel_Name"]), r.get("Grid_Size", ""))
            per_dispatch[key][r["Counter_Name"]] = per_dispatch[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        for (did, k, g), cs in per_dispatch.items():
            for c, v in cs.items():
                acc[(k, g)][c].append(v)
    for (k, g) in sorted(acc):
        if not any(x in k for x in ("k_rollout", "k_backward", "k_derivs")):
            continue
        print("%s grid=%s" % (k, g))
        for c in sorted(acc[(k, g)]):
            v = acc[(k, g)][c]
            print("    %-32s n=%3d mean=%.6g" % (c, len(v), sum(v) / len(v)))

    if config5_json:
        import json
        out, total = {}, 0.0
        for (k, g), cs in sorted(acc.items()):
            if "FETCH_SIZE" not in cs or "WRITE_SIZE" not in cs or "k_rollout<1>" in k:  # (<1>: the initial roll-out, not part of an iteration)
                continue
            # every kernel an iteration dispatches (derivatives, backward, roll-outs, selection, adoption of the kept
            # roll-outs, update); not the set-up of a solve: layout kernels, reset, runtime fills / copies
            if not any(x in k for x in ("k_rollout", "k_backward", "k_derivs", "k_adopt", "k_select", "k_update", "k_search", "k_commit", "k_multipliers")):
                continue
            f = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"])
            w = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
            per_launch = 2.0 * f * 1024 + w * 1024
            per_iter = len(cs["FETCH_SIZE"]) / config5_iters
            out["%s grid=%s" % (k, g)] = {"hbm_bytes_per_launch": per_launch, "launches_per_iteration": per_iter,
                                          "fetch_size_KiB_raw": f, "write_size_KiB_raw": w}
            total += per_launch * per_iter
        out["iteration"] = {"hbm_bytes": total, "correction": "FETCH_SIZE x2 (gfx950 wide coalesced reads), both KiB -> bytes",
                            "iterations_profiled": config5_iters,
                            "kernels_included": "every dispatch of k_derivs*, k_backward*, k_rollout* (but the initial roll-out), k_search*, k_select, k_adopt*, k_commit, k_update, k_multipliers"}
        out["_source_sha"] = source_sha()
        json.dump(out, open(config5_json, "w"), indent=1, sort_keys=True)
        print("wrote", config5_json)

    if traffic_json:
        import json
        # per bench label: mean over ALL launches of that kernel (the batch runs as several groups of trajectories, so
        # grids differ slightly).  The general roll-out kernel is launched two ways, told apart by their grids: the
        # first search stage has ls_split (3) lanes per trajectory of the group (grid = 3 Bp); the second launch is the
        # winner pass of the trajectories that stage settled (Bp lanes) side by side with the second stage, n_alpha -
        # ls_split (5) lanes per trajectory for the WORST case (most of those blocks return at once): grid = 6 Bp.
        ls_split, n_alpha = 3, 8
        grids = sorted({int(g) for (k, g) in acc if k == "void k_rollout<0>" and g})
        second = n_alpha - ls_split + 1
        bases = [g // ls_split for g in grids if g % ls_split == 0 and (g // ls_split) * second in grids]
        roll_label = {}
        for b in bases:
            roll_label[b * ls_split] = "k_rollout[search]"
            roll_label[b * second] = "k_rollout[stage 2 | winner]"
        per_label = collections.defaultdict(lambda: collections.defaultdict(list))
        for (k, g), cs in acc.items():
            label = KERNEL_LABELS.get(k)
            if k == "void k_rollout<0>" and g:
                label = roll_label.get(int(g))
            if not label:
                continue
            for cname in ("FETCH_SIZE", "WRITE_SIZE"):
                per_label[label][cname] += cs.get(cname, [])
        out = {}
        for label, cs in per_label.items():
            f = sum(cs["FETCH_SIZE"]) / max(1, len(cs["FETCH_SIZE"]))
            w = sum(cs["WRITE_SIZE"]) / max(1, len(cs["WRITE_SIZE"]))
            out[label] = {"fetch_size_KiB_raw": f, "write_size_KiB_raw": w, "launches": len(cs["FETCH_SIZE"]),
                          "hbm_bytes_per_launch": 2.0 * f * 1024 + w * 1024,
                          "correction": "FETCH_SIZE x2 (gfx950 wide coalesced reads), both KiB -> bytes"}
        out["_source_sha"] = source_sha()
        json.dump(out, open(traffic_json, "w"), indent=1, sort_keys=True)
        print("wrote", traffic_json)


if __name__ == "__main__":
    main()

"""per-kernel sums of the counters of rocprofv3 --pmc passes: python pmc_kernels.py [--issue-json FILE STEPS [--wave-steps X ROWS]] <dir> [<dir> ...]
--issue-json FILE STEPS: also write, per kernel, the vector instructions a wavefront issues per time step (SQ_INSTS_VALU /
SQ_WAVES / STEPS — STEPS = steps a wavefront of a sweep kernel walks, the horizon) and the fraction of a wavefront's cycles
in which it issues a vector instruction (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES), stamped with the digest of the sources
(ddp-generator_amd/evidence.py); bench.py puts them into roofline.issue."""
import csv, glob, json, os, sys, re, collections
argv = sys.argv[1:]
issue_json, steps, sweep_steps = None, 1, None
if argv and argv[0] == "--issue-json":
    issue_json, steps, argv = argv[1], int(argv[2]), argv[3:]
if argv and argv[0] == "--wave-steps":  # steps the wavefronts of the persistent backward kernel walked in the counted run, and the
    sweep_steps, rows, argv = float(argv[1]), int(argv[2]), argv[3:]  # trajectories a wavefront serves per step (quad mapping: 4)
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for d in argv:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"^void |\(anonymous namespace\)::|\(.*", "", r["Kernel_Name"])[:30]
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in tot.items():
    print(k, {a: "%.3g" % b for a, b in sorted(c.items())})
if issue_json:
    import importlib.util
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("ilqg_evidence", os.path.join(here, "..", "ddp-generator_amd", "evidence.py"))
    ev = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ev)
    out = {}
    for k, c in tot.items():
        if not k.startswith("k_") or not c.get("SQ_WAVES") or not c.get("SQ_WAVE_CYCLES"):
            continue
        out[k] = {"valu_insts_per_wave": c.get("SQ_INSTS_VALU", 0.0) / c["SQ_WAVES"],
                  "valu_insts_per_wave_and_step": c.get("SQ_INSTS_VALU", 0.0) / c["SQ_WAVES"] / steps,
                  "active_valu_frac": c.get("SQ_ACTIVE_INST_VALU", 0.0) / c["SQ_WAVE_CYCLES"],
                  "wait_any_frac": c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"],
                  "waves": c["SQ_WAVES"], "valu_insts_total": c.get("SQ_INSTS_VALU", 0.0)}
        # the wave mapping's backward kernels are PERSISTENT (wavefronts take trajectories from a queue and walk their sweeps,
        # lambda retries included): instructions / (wavefronts x horizon) is not a per-step figure for them; only the step
        # count of the profile build (--wave-steps) makes one, and bench.py reports none without it
        if re.match(r"k_backward_(quad|wave)", k):
            out[k]["persistent"] = True
            del out[k]["valu_insts_per_wave_and_step"]
        if sweep_steps and k.startswith("k_backward"):
            # persistent wavefronts (a worker walks many trajectories, sweeps are cut short by lambda retries): per step the
            # wavefronts really walked — counted by the kernel itself in the -DILQG_PROFILE_SECTIONS build of the same sources
            # (tools/section_profile_quad.py, the same first iteration) — and per trajectory step with all rows at work
            out[k]["valu_insts_per_wavefront_step"] = c.get("SQ_INSTS_VALU", 0.0) / sweep_steps
            out[k]["valu_insts_per_trajectory_step"] = c.get("SQ_INSTS_VALU", 0.0) / sweep_steps / rows
            out[k]["wavefront_steps"] = sweep_steps
            out[k]["trajectories_per_wavefront"] = rows
    out["_steps"] = steps
    out["_source_sha"] = ev.source_sha()
    out["_counters"] = "rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY (separate passes), summed over the run's dispatches of the kernel"
    json.dump(out, open(issue_json, "w"), indent=1, sort_keys=True)
    print("wrote", issue_json)

"""What ties a committed profile to the code it was collected on.

`profiles/traffic*.json` (HBM bytes per launch from rocprofv3 PMC passes) and `profiles/issue*.json` (vector
instructions per step, issue activity) are collected in separate profiler runs and read by bench.py, which cannot run
the counters inside its timed window.  Each file carries `_source_sha`: the digest below of the kernel sources and of
the two benchmark problems' generated files at collection time.  bench.py reports a figure from such a file only while
the digest still matches; after a kernel change the figure is `null` with the reason, until the profile is collected
again (tools/round_profile.sh)."""
import glob
import hashlib
import os

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def source_files():
    files = [f for f in glob.glob(os.path.join(HERE, "csrc", "*")) if os.path.isfile(f)]
    for prob in ("carparking", "synth16x8"):
        files += glob.glob(os.path.join(ROOT, "problems", prob, "iLQG_*"))
    return sorted(files)


def source_sha():
    h = hashlib.sha256()
    for f in source_files():
        h.update(os.path.relpath(f, ROOT).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def load_stamped(path):
    """(content, None) of a stamped profile whose digest matches the current sources, else (None, reason)"""
    import json
    if not os.path.exists(path):
        return None, "%s not present" % os.path.relpath(path, ROOT)
    j = json.load(open(path))
    have, want = j.get("_source_sha"), source_sha()
    if have != want:
        return None, ("%s was collected on sources %s, the current ones are %s: collect it again (tools/round_profile.sh)"
                      % (os.path.relpath(path, ROOT), have, want))
    return j, None

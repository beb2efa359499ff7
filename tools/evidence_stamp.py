#!/usr/bin/env python3
"""
Ties the static evidence files under profiles/ (profiles/kernel_flops.json: instruction / flop counts from the ISA;
profiles/pmc_traffic.json: HBM bytes per launch from the PMC passes) to the kernel sources they were taken from.

Every record carries `source = {"files": [...], "sha256_16": "..."}`: the first 16 hex digits of the SHA-256 over the listed
csrc files (in the listed order).  bench.py recomputes the digest of the tree it runs from and reports a figure as null
(with `"stale": true` beside it) when the sources have changed since the figure was measured -- a constant can no longer
outlive the kernel it describes.

    python tools/evidence_stamp.py --check            exit 1 and list what is stale
    python tools/evidence_stamp.py --stamp NAME ...   re-stamp the named top-level records (after re-measuring them!)
"""
import argparse
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "multiple-quadrotor-slam_amd", "csrc")

# which sources decide which record
# (mqs_common.h -- host-side declarations shared by every translation unit -- is deliberately not part of a digest: it changes
# with every new entry point and holds no kernel arithmetic)
SOURCES = {
    "ba": ["ba.hip", "ba_math.h", "wave_reduce.h", "tri_math.h", "peer_dev.h"],
    "tri": ["triangulate.hip", "tri_math.h"],
}
RECORD_SOURCES = {          # file -> record key -> source set
    "kernel_flops.json": {"ba_linearize_kernel<4>": "ba", "ba_linearize_lane_kernel<4>": "ba"},
    "pmc_traffic.json": {"ba": "ba", "tri": "tri"},
}


def digest(files):
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def source_record(name):
    return {"files": SOURCES[name], "sha256_16": digest(SOURCES[name])}


def is_current(src):
    """src: a `source` record.  True when the tree's files still hash to the recorded digest."""
    try:
        return bool(src) and digest(src["files"]) == src["sha256_16"]
    except (OSError, KeyError, TypeError):
        return False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--stamp", nargs="*", default=None, help="record keys to re-stamp (kernel_flops.json keys, or ba / tri for pmc_traffic.json)")
    args = ap.parse_args()
    stale = []
    for fname, recs in RECORD_SOURCES.items():
        path = os.path.join(ROOT, "profiles", fname)
        data = json.load(open(path))
        changed = False
        for key, srcname in recs.items():
            holder = data.setdefault("sources", {}) if fname == "pmc_traffic.json" else data.get(key)
            if holder is None:
                continue
            cur = holder.get(key if fname == "pmc_traffic.json" else "source")
            if args.stamp is not None and key in args.stamp:
                holder[key if fname == "pmc_traffic.json" else "source"] = source_record(srcname)
                changed = True
            elif not is_current(cur):
                stale.append("%s: %s" % (fname, key))
        if changed:
            json.dump(data, open(path, "w"), indent=1)
    if args.check:
        for s in stale:
            print("stale:", s)
        return 1 if stale else 0
    return 0


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""Register / scratch report of every gfx950 kernel in the library: compiles each HIP source with
`-Rpass-analysis=kernel-resource-usage` (no GPU needed) and prints the kernels that spill or use scratch memory --
the check that found the 964 SGPR spills of the sparse Cholesky's diagonal-block kernel and the scratch-resident index
arrays of the matcher.  python tools/kernel_resources.py [--all] [source.hip ...]"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FIELDS = ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs Spill", "VGPRs Spill")


def report(src):
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", src, "-o", os.devnull,
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-2000:])
    kernels, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            kernels.append(cur)
            continue
        for f in FIELDS:
            m = re.search(r"remark:\s+" + re.escape(f) + r": (\d+)", line)
            if m and cur is not None:
                cur[f] = int(m.group(1))
    return kernels


def demangle(names):
    if not names:                      # c++filt without arguments would wait on stdin
        return names
    try:
        out = subprocess.run(["c++filt"] + names, capture_output=True, text=True, stdin=subprocess.DEVNULL).stdout.splitlines()
        return out if len(out) == len(names) else names
    except OSError:
        return names


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    show_all = "--all" in sys.argv
    srcs = args or sorted(glob.glob(os.path.join(ROOT, "multiple-quadrotor-slam_amd", "csrc", "*.hip")))
    flagged = 0
    for src in srcs:
        ks = report(src)
        names = demangle([k["name"] for k in ks])
        for k, nm in zip(ks, names):
            bad = k.get("SGPRs Spill", 0) or k.get("VGPRs Spill", 0) or k.get("ScratchSize [bytes/lane]", 0)
            if bad or show_all:
                flagged += bool(bad)
                print("%-18s %-90s vgpr %3d  occ %d  scratch %4d B  spill sgpr %4d vgpr %4d" % (
                    os.path.basename(src), re.sub(r"\(anonymous namespace\)::", "", nm)[:90], k.get("VGPRs", 0),
                    k.get("Occupancy [waves/SIMD]", 0), k.get("ScratchSize [bytes/lane]", 0), k.get("SGPRs Spill", 0),
                    k.get("VGPRs Spill", 0)))
    print("%d kernel(s) spill or use scratch" % flagged)


if __name__ == "__main__":
    main()

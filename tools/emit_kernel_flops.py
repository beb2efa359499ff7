#!/usr/bin/env python3
"""
Re-emits profiles/kernel_flops.json from the gfx950 ISA of the tree's kernels (no GPU needed): the four-landmark chunk body of
ba_linearize_wave_kernel<4> without and with the lens-distortion model, counted by tools/isa_mix.py, divided by the four
landmarks of a chunk, and stamped with the digest of the sources (tools/evidence_stamp.py) so that bench.py can tell when the
figures no longer describe the kernel it runs.

    python tools/emit_kernel_flops.py            # writes profiles/kernel_flops.json and profiles/rNN-style copy if --copy DIR
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import evidence_stamp  # noqa: E402
import isa_mix  # noqa: E402

SRC = os.path.join(ROOT, "multiple-quadrotor-slam_amd", "csrc", "ba.hip")


def count(defines, kernel_substr):
    asm = isa_mix.compile_asm(SRC, defines)
    kernels = isa_mix.split_kernels(asm)
    names = list(kernels)
    pretty = isa_mix.demangle(names)
    for name, nice in zip(names, pretty):
        if kernel_substr in nice:
            total, detail, _ = isa_mix.histogram(kernels[name])
            return total, detail
    raise SystemExit("kernel %r not found" % kernel_substr)


def per_landmark(total, detail, landmarks):
    fma = detail.get("v_fma_f64", 0) + detail.get("v_fmac_f64", 0)
    fp64 = total.get("fp64_arith", 0)
    other64 = fp64 - fma
    valu = sum(total.get(k, 0) for k in ("fp64_arith", "fp64_trans", "fp64_cmp", "lane_exchange", "mov_select", "valu_other", "mfma"))
    swaps = detail.get("v_permlane32_swap_b32", 0) + detail.get("v_permlane16_swap_b32", 0)
    n = sum(total.values())
    return {"fp64_flop_per_landmark": round((2 * fma + other64) / landmarks), "fp64_instructions_per_landmark": round(fp64 / landmarks),
            "fma_instructions_per_landmark": round(fma / landmarks), "valu_instructions_per_landmark": round(valu / landmarks),
            "valu_issue_slots_per_landmark": round((valu + swaps) / landmarks), "instructions_per_landmark_all_kinds": round(n / landmarks),
            "instructions_per_chunk": n, "mix_per_chunk": dict(total)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--copy", help="also write the record to this path (e.g. profiles/r03/01_kernel_flops.json)")
    args = ap.parse_args()
    nodist = per_landmark(*count(["MQS_WL_ONLY_L4", "MQS_WL_ONLY_NODIST=1"], "ba_linearize_wave_kernel<4, true>"), landmarks=4)
    dist = per_landmark(*count(["MQS_WL_ONLY_L4", "MQS_WL_ONLY_NODIST=0"], "ba_linearize_wave_kernel<4, true>"), landmarks=4)
    rec = {
        "ba_linearize_kernel<4>": dict(
            nodist,
            kernel="ba_linearize_wave_kernel<4, true>, the four-landmark chunk body for cameras without lens distortion (the benchmark's "
                   "Cal3DS2(480,480,0,320,240,0,0,0,0); 15 of the 15.26 rows per wave at 1e6 x 4 run in it)",
            with_lens_distortion={k: dist[k] for k in ("fp64_flop_per_landmark", "fp64_instructions_per_landmark",
                                                        "valu_instructions_per_landmark", "instructions_per_chunk")},
            method="static count of the fully unrolled chunk body (tools/emit_kernel_flops.py = tools/isa_mix.py with --define "
                   "MQS_WL_ONLY_L4 --define MQS_WL_ONLY_NODIST=1) / 4 landmarks; v_fma/v_fmac_f64 = 2 flop, other fp64 arithmetic = 1; "
                   "v_permlane*_swap issue in two passes (tools/probes/valu_issue.hip); one wave per SIMD issues ONE instruction of any "
                   "kind per slot, so every instruction of the body counts against time",
            source=evidence_stamp.source_record("ba")),
    }
    path = os.path.join(ROOT, "profiles", "kernel_flops.json")
    json.dump(rec, open(path, "w"), indent=1)
    if args.copy:
        json.dump(rec, open(os.path.join(ROOT, args.copy), "w"), indent=1)
    print(json.dumps({k: v for k, v in rec["ba_linearize_kernel<4>"].items() if k not in ("mix_per_chunk", "method", "kernel")}))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Static instruction-mix histogram of the gfx950 kernels: compiles a HIP source to assembly (no GPU needed), splits the
listing into kernels and counts the instruction classes that decide an fp64-VALU-bound kernel's time:

  fp64 arithmetic   v_fma_f64 / v_mul_f64 / v_add_f64 (+ min/max/ldexp/frexp/cmp on doubles)
  fp64 transcend.   v_rcp_f64 / v_rsq_f64 / v_sqrt_f64 (quarter rate)
  lane exchange     DPP moves, v_permlane*_swap, v_readlane / v_writelane, ds_swizzle / ds_bpermute
  moves / selects   v_mov_b32 / v_mov_b64 / v_accvgpr_* / v_cndmask_b32
  other VALU        integer / address arithmetic, conversions, 32-bit compares
  LDS               ds_read* / ds_write*
  global            global_load* / global_store* / buffer_*
  scalar            s_* except waits;  waits = s_waitcnt / s_nop / s_barrier

The count is STATIC (each instruction once): for the straight-line, fully unrolled per-landmark bodies of the BA lineariser
and the triangulation kernels that is the per-batch count up to the loop prologue; loops with a dynamic trip count are
reported with their per-iteration body separately by giving --loops (basic blocks that branch backwards).

    python tools/isa_mix.py [--kernel SUBSTR] [--json out.json] [--define NAME=V ...] source.hip
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

FP64_ARITH = re.compile(r"^v_(fma|mul|add|fmac|min|max|ldexp|frexp_mant|frexp_exp_i32|trunc|floor|ceil|rndne|fract|div_fixup|div_fmas|div_scale)_f64")
FP64_TRANS = re.compile(r"^v_(rcp|rsq|sqrt)_f64")
FP64_CMP = re.compile(r"^v_cmp[a-z_]*_f64")
EXCHANGE = re.compile(r"^(v_permlane|v_readlane|v_writelane|v_readfirstlane|ds_swizzle|ds_bpermute|ds_permute)")
MOVES = re.compile(r"^(v_mov_b32|v_mov_b64|v_accvgpr|v_cndmask_b32|v_swap_b32)")


def classify(op, operands):
    if op.startswith("v_"):
        dpp = ("row_" in operands or "quad_perm" in operands or "wave_" in operands) and "dpp" in op or "_dpp" in op \
            or re.search(r"\b(row_shl|row_shr|row_ror|quad_perm|row_mirror|row_half_mirror|row_bcast|row_newbcast)", operands)
        if dpp and not FP64_ARITH.match(op):
            return "lane_exchange"
        if FP64_ARITH.match(op):
            return "fp64_arith"
        if FP64_TRANS.match(op):
            return "fp64_trans"
        if FP64_CMP.match(op):
            return "fp64_cmp"
        if EXCHANGE.match(op):
            return "lane_exchange"
        if MOVES.match(op):
            return "mov_select"
        if op.startswith("v_mfma") or op.startswith("v_smfmac"):
            return "mfma"
        return "valu_other"
    if op.startswith("ds_"):
        return "lane_exchange" if EXCHANGE.match(op) else "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "scratch" if op.startswith("scratch_") else "global"
    if op.startswith("s_"):
        if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep")):
            return "wait"
        return "scalar"
    return "other"


def compile_asm(src, defines):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src, "-o", out]
        cmd += ["-D" + d for d in defines]
        cmd += os.environ.get("MQS_ISA_MIX_FLAGS", "").split()          # extra compiler flags for what-if counts
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-3000:])
        return open(out).read()


def split_kernels(asm):
    """name -> list of (label, [ (op, operands) ]) basic blocks."""
    kernels = {}
    cur = None
    blocks = None
    for line in asm.splitlines():
        s = line.strip()
        m = re.match(r"^([A-Za-z_.$][\w.$]*):", s)
        if m and not s.startswith(".L"):
            name = m.group(1)
            if name.startswith("_Z") or name.startswith("mqs"):
                cur = name
                blocks = [("entry", [])]
                kernels[cur] = blocks
            continue
        if cur is None:
            continue
        if s.startswith(".Lfunc_end") or s.startswith(".section") or s.startswith(".rodata"):
            cur = None
            continue
        m = re.match(r"^(\.LBB[\w.]+):", s)
        if m:
            blocks.append((m.group(1), []))
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        s = s.split(";")[0].strip()
        parts = s.split(None, 1)
        op = parts[0]
        if not re.match(r"^[a-z]", op):
            continue
        blocks[-1][1].append((op, parts[1] if len(parts) > 1 else ""))
    return kernels


def histogram(blocks):
    total = collections.Counter()
    detail = collections.Counter()
    per_block = []
    labels = [b[0] for b in blocks]
    for idx, (label, ins) in enumerate(blocks):
        h = collections.Counter()
        back = None
        for op, operands in ins:
            cls = classify(op, operands)
            h[cls] += 1
            detail[re.sub(r"_e(32|64)$|_dpp$|_sdwa$", "", op)] += 1
            if op.startswith(("s_cbranch", "s_branch")):
                tgt = operands.strip()
                if tgt in labels and labels.index(tgt) <= idx:
                    back = tgt
        total.update(h)
        per_block.append({"label": label, "n": sum(h.values()), "backward_branch_to": back, "mix": dict(h)})
    return total, detail, per_block


def demangle(names):
    try:
        out = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
        return out if len(out) == len(names) else names
    except OSError:
        return names


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("source")
    ap.add_argument("--kernel", action="append", default=[], help="only kernels whose demangled name contains this")
    ap.add_argument("--json")
    ap.add_argument("--define", action="append", default=[])
    ap.add_argument("--loops", action="store_true", help="list loop bodies (blocks with a backward branch)")
    ap.add_argument("--top", type=int, default=14)
    args = ap.parse_args()
    asm = compile_asm(args.source, args.define)
    kernels = split_kernels(asm)
    names = list(kernels)
    pretty = [re.sub(r"\(anonymous namespace\)::", "", d) for d in demangle(names)]
    out = {}
    for name, nice in zip(names, pretty):
        if args.kernel and not any(k in nice for k in args.kernel):
            continue
        total, detail, per_block = histogram(kernels[name])
        n = sum(total.values())
        if n < 8:
            continue
        valu = sum(v for k, v in total.items() if k in ("fp64_arith", "fp64_trans", "fp64_cmp", "lane_exchange", "mov_select",
                                                         "valu_other", "mfma"))
        short = nice.split("(")[0][:100]
        print("%s\n  instructions %d  VALU-issued %d" % (short, n, valu))
        for k in ("fp64_arith", "fp64_trans", "fp64_cmp", "lane_exchange", "mov_select", "valu_other", "mfma", "lds", "global",
                  "scratch", "scalar", "wait"):
            if total.get(k):
                print("    %-14s %6d  %5.1f %%" % (k, total[k], 100.0 * total[k] / n))
        print("    top: " + ", ".join("%s %d" % kv for kv in detail.most_common(args.top)))
        if args.loops:
            for b in per_block:
                if b["backward_branch_to"]:
                    print("    loop end %s -> %s (%d instr in its last block)" % (b["label"], b["backward_branch_to"], b["n"]))
        out[short] = {"instructions": n, "valu_issued": valu, "mix": dict(total), "ops": dict(detail.most_common(60)),
                      "blocks": per_block if args.loops else None}
    if args.json:
        with open(args.json, "w") as f:
            json.dump({"source": os.path.relpath(os.path.abspath(args.source), ROOT), "defines": args.define, "kernels": out}, f,
                      indent=1)


if __name__ == "__main__":
    main()

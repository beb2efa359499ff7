#!/usr/bin/env python3
"""
Static estimate of where a single wavefront of a straight-line kernel parks in s_waitcnt: walks one kernel of a device
assembly listing (hipcc --offload-device-only -S), keeps the queue of outstanding LGKM operations (ds_*, s_load*; LDS operations
return in order, a scalar load anywhere in the queue forces the wait to drain everything older too) and of outstanding vector
memory operations, and reports for every s_waitcnt the AGE, in issued instructions, of the youngest operation it has to wait
for.  At one wave per SIMD an instruction issues every ~4 cycles, so an age below latency / 4 (LDS ~ 90-130 cycles, scalar
cache ~ 200-400, L2 ~ 500+) is latency the wave sits out.

    python tools/isa_wait_exposure.py build/asm/ba_l4.s KERNEL_SUBSTRING [--lds 110 --smem 375 --vmem 2000]
"""
import argparse
import re
import sys


def kernel_body(path, substr):
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        head = l.split(";")[0].rstrip()
        if head.endswith(":") and substr in head and not l.startswith((".", ";", "\t")):
            start = i
            break
    if start is None:
        raise SystemExit("kernel not found")
    out = []
    for l in lines[start + 1:]:
        t = l.strip()
        if t.startswith("s_endpgm"):
            break
        if not t or t.startswith(".") or t.split(";")[0].rstrip().endswith(":"):
            continue
        if t.startswith(";"):
            if "MQS_MARK" in t:
                out.append(("mark", t))
            continue
        out.append(("inst", t.split(";")[0].strip()))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("kernel")
    ap.add_argument("--lds", type=float, default=110.0)
    ap.add_argument("--smem", type=float, default=375.0)
    ap.add_argument("--vmem", type=float, default=2000.0)
    ap.add_argument("--cpi", type=float, default=4.3, help="cycles per issued instruction of a lone wave")
    ap.add_argument("--list", action="store_true")
    a = ap.parse_args()
    body = kernel_body(a.asm, a.kernel)
    lgkm, vm = [], []                    # (index of issue, kind)
    n = 0
    exposed = {"lds": 0.0, "smem": 0.0, "vmem": 0.0}
    sites = []
    phase = "?"
    per_phase = {}
    lat = {"lds": a.lds, "smem": a.smem, "vmem": a.vmem}
    clock = 0.0                          # modelled time: issue + waits sat out
    issue_time = {}
    for kind, t in body:
        if kind == "mark":
            phase = t.split("MQS_MARK")[1].strip()
            continue
        n += 1
        clock += a.cpi
        op = t.split()[0]
        if op.startswith("ds_") and not op.startswith(("ds_swizzle", "ds_bpermute", "ds_permute")) or op.startswith(("ds_swizzle", "ds_bpermute", "ds_permute")):
            lgkm.append((n, "lds", clock))
        elif op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_memtime") or op.startswith("s_dcache"):
            lgkm.append((n, "smem", clock))
        elif op.startswith(("global_load", "buffer_load", "global_store", "buffer_store", "global_atomic", "flat_")):
            vm.append((n, "vmem", clock))
        elif op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", t)
            worst = 0.0
            what = None
            if m:
                keep = int(m.group(1))
                has_smem = any(k == "smem" for _, k, _ in lgkm)
                waited = lgkm if (has_smem or keep == 0) else lgkm[:max(0, len(lgkm) - keep)]
                for (i, k, tc) in waited:
                    rem = tc + lat[k] - clock
                    if rem > worst:
                        worst, what = rem, (k, n - i)
                lgkm = [] if (has_smem or keep == 0) else lgkm[len(lgkm) - keep:] if keep else []
            m = re.search(r"vmcnt\((\d+)\)", t)
            if m:
                keep = int(m.group(1))
                waited = vm[:max(0, len(vm) - keep)]
                for (i, k, tc) in waited:
                    rem = tc + lat[k] - clock
                    if rem > worst:
                        worst, what = rem, (k, n - i)
                vm = vm[len(vm) - keep:] if keep else []
            if worst > 0:
                clock += worst
                exposed[what[0]] += worst
                per_phase.setdefault(phase, {"lds": 0.0, "smem": 0.0, "vmem": 0.0})[what[0]] += worst
                sites.append((n, t, what, round(worst), phase))
    print("instructions", n, "modelled cycles", round(clock), "issue", round(n * a.cpi))
    print("exposed cycles by kind", {k: round(v) for k, v in exposed.items()})
    for ph, d in per_phase.items():
        print("  after mark %-22s" % ph, {k: round(v) for k, v in d.items()})
    if a.list:
        for s in sites:
            print(s)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""
Summarises rocprofv3 `--pmc` output (…_counter_collection.csv, one or several passes) per kernel:
mean counter values per dispatch, mean dispatch duration, and the derived figures quoted in DESIGN.md.

    python tools/pmc_summary.py gpurun_out/pmc_valu [more dirs] > profiles/rNN/xx_pmc_valu_summary.json

Derived (gfx950, 256 CUs x 4 SIMDs; a wave64 VALU instruction holds its SIMD's vector issue for 4 cycles --
MI355X_MICROARCH.md, cycle-constants table, row 'vector-instruction ISSUE cost'):
  valu_insts_per_wave        SQ_INSTS_VALU / SQ_WAVES
  valu_issue_us_at_2.4GHz    SQ_INSTS_VALU x 4 cycles / 1024 SIMDs / 2.4 GHz: the time the kernel's vector instructions
                             need if every SIMD issued one every 4 cycles at the nominal clock
  valu_issue_frac_at_2.4GHz  that time / the measured duration (the chip holds a lower clock under fp64-dense load, so
                             1.0 is not reachable; the GRBM_GUI_ACTIVE clock estimate is unusable on dispatches < 0.3 ms)
  active_inst_valu_over_wave_cycles   SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (both in quad-cycles, summed over waves):
                             times the resident waves per SIMD = the share of time a SIMD's VALU is busy
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def main():
    dirs = sys.argv[1:] or ["gpurun_out/pmc_valu"]
    vals = defaultdict(lambda: defaultdict(list))          # kernel -> counter -> [values per dispatch]
    durs = defaultdict(list)
    waves_hint = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            seen = set()
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"]
                vals[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                key = (f, row["Dispatch_Id"])
                if key not in seen:
                    seen.add(key)
                    durs[k].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
                    waves_hint[k] = int(row["Grid_Size"]) // 64
    out = {}
    for k, cs in vals.items():
        # a counter spread over the 8 XCD instances appears as several rows per dispatch: rocprofv3 already sums them
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        # skip the first dispatch of each kernel (cold caches / clock ramp) when there are several
        dur = sorted(durs[k])[len(durs[k]) // 2]
        mm = re.search(r"(\w+_kernel\w*(<[^>]*>)?|__amd_\w+|\w+Kernel\w*)", k)
        short = mm.group(1) if mm else k[:90]
        rec = {"dispatches": len(durs[k]), "median_duration_us": round(dur / 1e3, 2), "counters": {c: round(v, 1) for c, v in m.items()}}
        if "SQ_INSTS_VALU" in m:
            waves = m.get("SQ_WAVES", waves_hint.get(k, 0))
            if waves:
                rec["valu_insts_per_wave"] = round(m["SQ_INSTS_VALU"] / waves, 1)
            issue_us = m["SQ_INSTS_VALU"] * 4.0 / 1024.0 / 2.4e3
            rec["valu_issue_us_at_2.4GHz"] = round(issue_us, 2)
            if dur > 0:
                rec["valu_issue_frac_at_2.4GHz"] = round(issue_us / (dur / 1e3), 4)
        if "SQ_ACTIVE_INST_VALU" in m and "SQ_WAVE_CYCLES" in m and m["SQ_WAVE_CYCLES"] > 0:
            rec["active_inst_valu_over_wave_cycles"] = round(m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"], 4)
        if "SQ_INSTS_VALU_MFMA_F64" in m or "SQ_INSTS_MFMA" in m:
            rec["mfma_insts"] = m.get("SQ_INSTS_MFMA")
        out[short] = rec
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()

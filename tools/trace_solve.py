#!/usr/bin/env python3
"""Launch-by-launch timeline of the LAST banded solve in a rocprofv3 kernel trace (the launches between the last damping
kernel and the retraction): python tools/trace_solve.py <..._kernel_trace.csv>.  Used for profiles/r02/11_*."""
import csv, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "sparse_damp" in r["Kernel_Name"]]
i0 = idx[-1]
t0 = int(rows[i0]["Start_Timestamp"])
tot = {}
for r in rows[i0:i0 + 400]:
    nm = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).split("::")[-1]
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(nm.ljust(24), "start %7.1f us  dur %6.1f us  grid %s x %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, dur,
                                                                   r.get("Grid_Size_X", ""), r.get("Grid_Size_Y", "")))
    tot[nm] = tot.get(nm, 0.0) + dur
    if "retract" in nm:
        print("total %.1f us" % ((int(r["End_Timestamp"]) - t0) / 1e3))
        break
print({k: round(v, 1) for k, v in tot.items()})

#!/usr/bin/env python3
"""
profiles/pmc_traffic.json from a PMC summary (tools/pmc_summary.py over the FETCH_SIZE and WRITE_SIZE passes of
tools/profile_round.sh): HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- FETCH_SIZE reports half of a wide
coalesced read on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section) -- for the kernels of the bench step at 1e6 x 4, stamped
with the digest of the sources they were measured on (tools/evidence_stamp.py).

    python tools/emit_pmc_traffic.py profiles/r03/03_pmc_hbm_traffic_summary.json [profiles/r03/04_pmc_valu_summary.json]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import evidence_stamp  # noqa: E402


def find(summary, *needles):
    for k, v in summary.items():
        if all(n in k for n in needles):
            return v
    return None


def traffic(rec):
    if not rec or "FETCH_SIZE" not in rec["counters"] or "WRITE_SIZE" not in rec["counters"]:
        return None
    return int(round((2.0 * rec["counters"]["FETCH_SIZE"] + rec["counters"]["WRITE_SIZE"]) * 1024))


def main():
    hbm = json.load(open(sys.argv[1]))
    valu = json.load(open(sys.argv[2])) if len(sys.argv) > 2 else {}
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    out = {"landmarks": 1000000, "cams": 4,
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/bench_ba.py and tools/bench_tri.py "
                     "(tools/profile_round.sh; summary %s); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE reads 1/2 of a 16 B/lane "
                     "coalesced stream on gfx950 (MI355X_MICROARCH.md, HBM)" % os.path.relpath(os.path.abspath(sys.argv[1]), ROOT)}
    tri = {"linear_ls": find(hbm, "tri_kernel<4", "0, false"), "iterative_ls": find(hbm, "tri_kernel<4", "1, false"),
           "linear_eigen": find(hbm, "tri_kernel<4", "2, false")}
    for name, rec in tri.items():
        out["%s_hbm_bytes_per_launch" % name] = traffic(rec)
    out["ba_linearize_hbm_bytes_per_launch"] = traffic(find(hbm, "ba_linearize_wave_kernel<4, true>"))
    out["ba_backsub_hbm_bytes_per_launch"] = traffic(find(hbm, "ba_backsub_kernel<4>"))
    out["ba_tail_hbm_bytes_per_launch"] = traffic(find(hbm, "ba_tail_kernel<4>"))
    lin = find(valu, "ba_linearize_wave_kernel<4, true>")
    if lin:
        out["valu"] = {"ba_linearize_schur": {"valu_instructions_per_launch": lin["counters"].get("SQ_INSTS_VALU"),
                                              "valu_instructions_per_wave": lin.get("valu_insts_per_wave"),
                                              "valu_issue_us_at_2.4GHz": lin.get("valu_issue_us_at_2.4GHz"),
                                              "launch_us_under_profiler": lin.get("median_duration_us"), "waves_per_simd": 1,
                                              "active_inst_valu_over_wave_cycles": lin.get("active_inst_valu_over_wave_cycles")}}
    def valu_of(rec, waves_per_simd):
        c = rec["counters"]
        return {"valu_instructions_per_launch": c.get("SQ_INSTS_VALU"), "valu_instructions_per_landmark_lane": round(c.get("SQ_INSTS_VALU", 0) * 64 / 1e6, 1),
                "valu_issue_us_at_2.4GHz": rec.get("valu_issue_us_at_2.4GHz"), "launch_us_under_profiler": rec.get("median_duration_us"),
                "waves_per_simd": waves_per_simd, "active_inst_valu_over_wave_cycles": rec.get("active_inst_valu_over_wave_cycles")}
    for key, needles, w in (("iterative_ls", ("tri_kernel<4", "1, false"), 3), ("tri_ls_and_iterative_fused", ("tri_kernel<4", "3, false"), 3),
                            ("ba_tail", ("ba_tail_kernel<4>",), 4)):
        rec = find(valu, *needles)
        if rec and "SQ_INSTS_VALU" in rec["counters"]:
            out.setdefault("valu", {})[key] = valu_of(rec, w)
    out["sources"] = {"ba": evidence_stamp.source_record("ba"), "tri": evidence_stamp.source_record("tri")}
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k.endswith("per_launch")}))


if __name__ == "__main__":
    main()

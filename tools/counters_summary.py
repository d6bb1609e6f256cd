#!/usr/bin/env python3
"""Summarise what tools/profile_counters.sh left under gpurun_out/<tag>/ :

    python tools/counters_summary.py gpurun_out/<tag>

* <tag>/ldati_sq_counters.txt : per LDATI kernel of the stress chunk (and the e2e regime when collected) the SQ counters,
  averaged per launch, with the derived shares DESIGN 4.2 quotes (VALU lane-instructions per event, issue share).
* <tag>/e2e_mfma_busy.txt / .json : MFMA-busy share of the SIMD cycles per conv instantiation,
  SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)  (bench.py reads the .json copied to profiles/).
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_BUSY_CYCLES count quad-cycles (MI355X_MICROARCH.md), SQ_INSTS_* wave
instructions, SQ_THREAD_CYCLES_VALU active-lane cycles (64 lanes x 4 cycles per full VALU wave instruction)."""
import collections
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _prov():
    """{lib_version, source_hash} of the tree the counters were collected from (bench.py ignores a summary of other sources)."""
    try:
        from v2ce_toolbox_amd import hip
        return hip.provenance()
    except Exception as e:                      # noqa: BLE001 -- a summary without provenance is simply never used by bench.py
        return {"error": repr(e)}


def kname(s):
    return re.sub(r"\(.*", "", s.replace("void ", "").replace("v2ce::(anonymous namespace)::", "")).replace(" ", "")


def load(dirs):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                per[kname(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return per


def durations(dirs):
    per = collections.defaultdict(list)
    for d in dirs:
        for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                per[kname(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    return per


def avg(v):
    return sum(v) / len(v) if v else 0.0


def ldati_table(out, wl, events_per_launch):
    dirs = sorted(glob.glob(os.path.join(out, f"sq_{wl}_s*")))
    dirs = [d for d in dirs if os.path.isdir(d)]
    if not dirs:
        return []
    per, dur = load(dirs), durations(dirs[:1])
    lines = [f"workload {wl}: SQ counters per launch (rocprofv3 --pmc, 4 counters per pass), events per LDATI call = {events_per_launch}"]
    tot_valu = 0.0
    # kernels that ran on the first call only (a fused pass whose hint missed / its two-pass repeat) are listed but not
    # summed into the per-call totals: they have at most half the launches of the steady-state kernels
    nl = {k: len(next(iter(c.values()))) for k, c in per.items() if k.startswith("ldati")}
    nmax = max(nl.values()) if nl else 0
    first_only = {k for k, n in nl.items() if 2 * n <= nmax}
    for k in sorted(per, key=lambda k: -avg(dur.get(k, [0]))):
        if not (k.startswith("ldati") or k.startswith("events")):
            continue
        c = {n: avg(v) for n, v in per[k].items()}
        n = len(next(iter(per[k].values())))
        lines.append(f"  {k}  (launches {n}, avg {avg(dur.get(k, [0])):.1f} us)")
        for name in sorted(c):
            lines.append(f"      {name:26s} {c[name]:18.0f}")
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        if wc:
            lines.append("      shares of SQ_WAVE_CYCLES: active-inst %.1f %% (VALU %.1f %%, LDS %.1f %%, VMEM %.1f %%), issue-stall %.1f %% "
                         "(LDS-issue %.1f %%), parked %.1f %%" % tuple(100 * c.get(x, 0.0) / wc for x in (
                             "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM",
                             "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY")))
        once = any(x in k for x in ("probe", "check", "slope_tab", "commit"))     # one-time device checks: not part of a call
        if c.get("SQ_INSTS_VALU") and events_per_launch:
            lane = 64.0 * c["SQ_INSTS_VALU"] / events_per_launch
            tot_valu += 0.0 if (once or k in first_only) else lane
            extra = ""
            if c.get("SQ_THREAD_CYCLES_VALU"):
                extra = ", active lanes per VALU instruction %.1f of 64" % (c["SQ_THREAD_CYCLES_VALU"] / 4.0 / c["SQ_INSTS_VALU"])
            lines.append(f"      VALU wave-instructions x 64 / event = {lane:.1f} lane-slots per event{extra}")
        if c.get("SQ_LDS_IDX_ACTIVE"):
            lines.append("      LDS bank conflicts: %.1f %% of the LDS-array cycles" % (100 * c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]))
    lines.append(f"  all LDATI kernels of a steady-state call (excluded: one-time device checks; kernels of the first call only: {sorted(first_only)}): {tot_valu:.1f} VALU lane-slots per event")
    # VALU-issue roofline of the whole call: executed VALU wave-instructions x 64 lanes against 1024 SIMDs x 16 lanes per clock
    # over the kernels' own busy clocks (GRBM_GUI_ACTIVE / 8 XCDs)
    js = {"events_per_call": events_per_launch, "valu_lane_slots_per_event": tot_valu, "kernels": {}}
    gui_tot, inst_tot = 0.0, 0.0
    for k, c in per.items():
        if not k.startswith("ldati") or any(x in k for x in ("probe", "check", "slope_tab", "commit")) or k in first_only:
            continue
        iv, gui = avg(c.get("SQ_INSTS_VALU", [])), avg(c.get("GRBM_GUI_ACTIVE", []))
        js["kernels"][k] = {"avg_us": avg(dur.get(k, [0])), "insts_valu": iv, "gui_active": gui,
                            "valu_frac": (iv * 64.0) / (16.0 * 1024.0 * gui / 8.0) if gui else None,
                            "active_lane_share": (avg(c.get("SQ_THREAD_CYCLES_VALU", [])) / 4.0 / iv / 16.0) if iv else None}
        gui_tot += gui
        inst_tot += iv
    js["valu_frac"] = (inst_tot * 64.0) / (16.0 * 1024.0 * gui_tot / 8.0) if gui_tot else None
    lines.append(f"  VALU-issue roofline of the call (VALU wave-instructions x 64 / (1024 SIMDs x 16 lanes/clk x busy clocks)): {js['valu_frac']}")
    js["_provenance"] = _prov()
    json.dump(js, open(os.path.join(out, f"{wl}_sq_counters.json"), "w"), indent=1)
    return lines


def mfma_table(out):
    d = os.path.join(out, "sq_e2e_mfma")
    if not os.path.isdir(d):
        return [], {}
    per, dur = load([d]), durations([d])
    lines = ["kernel, launches, avg us, MFMA busy share of SIMD cycles = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs), clock = GRBM_GUI_ACTIVE / 8 / time"]
    js = {}
    rows = []
    for k, c in per.items():
        if not k.startswith("conv3d"):
            continue
        busy, gui = avg(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [])), avg(c.get("GRBM_GUI_ACTIVE", []))
        if not gui:
            continue
        share = busy / 1024.0 / (gui / 8.0)
        us = avg(dur.get(k, [0]))
        rows.append((us * len(dur.get(k, [])), k, len(c["GRBM_GUI_ACTIVE"]), us, share, gui / 8.0 / us / 1e3 if us else 0.0))
        js[k] = {"launches": len(c["GRBM_GUI_ACTIVE"]), "avg_us": us, "mfma_busy": share, "clock_ghz": gui / 8.0 / us / 1e3 if us else None}
    for _, k, n, us, share, clk in sorted(rows, reverse=True):
        lines.append(f"{k:52s} {n:3d}  {us:8.1f}  {100 * share:5.1f} %   {clk:.2f} GHz")
    return lines, js


def main():
    out = sys.argv[1]
    ev = {}
    for wl in ("ldati_stress", "e2e", "ldati_sparse"):
        # events per call from the bench line of the same run when present
        for f in glob.glob(os.path.join(out, f"sq_{wl}_s1.log")):
            for line in open(f):
                if line.startswith("{"):
                    try:
                        j = json.loads(line)
                        ev[wl] = int(round(j["events_per_pair"] * j["config"]["frame_pairs_per_step_per_gpu"]))
                    except Exception:
                        pass
    text = []
    for wl in ("ldati_stress", "e2e", "ldati_sparse"):
        text += ldati_table(out, wl, ev.get(wl, 0))
    if text:
        open(os.path.join(out, "ldati_sq_counters.txt"), "w").write("\n".join(text) + "\n")
        print("\n".join(text))
    lines, js = mfma_table(out)
    if lines:
        open(os.path.join(out, "e2e_mfma_busy.txt"), "w").write("\n".join(lines) + "\n")
        js["_provenance"] = _prov()
        json.dump(js, open(os.path.join(out, "e2e_mfma_busy.json"), "w"), indent=1)
        print("\n".join(lines))


if __name__ == "__main__":
    main()

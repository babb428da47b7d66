#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (written by tools/gpu_profile.sh on the GPU box) into the tracked summaries
under profiles/:  <round>_<tag>_kernel_stats.csv, <round>_<tag>_pmc.json, and the pmc_traffic.json entry
bench.py reads for roofline.traffic.

usage: tools/summarize_prof.py <tag> <round> <traffic-key> [kernel-name-substring [output-suffix]]
       tools/summarize_prof.py --refill <traffic-key> <profiles/xx_pmc.json>   (add the pipe figures of a committed summary to its pmc_traffic.json entry)
(by default the kernel with the largest total duration is summarised; a substring picks another, e.g. wf_shade -> <round>_<tag>_<suffix>_pmc.json,
 and then no pmc_traffic.json entry is written)
"""
import collections
import csv
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag, rnd, key = sys.argv[1], sys.argv[2], sys.argv[3]
src = ROOT / "gpurun_out" / f"prof_{tag}"
dst = ROOT / "profiles"
dst.mkdir(exist_ok=True)

def pipe_figures(pmc, d, avg_ns):
    """The figures that say which pipe binds a kernel, from one profile (counter means per launch + the clean launch duration): replayed by bench.py
    next to `traffic`.  Also used to back-fill entries from a committed <round>_<tag>_pmc.json (tools/summarize_prof.py --refill)."""
    g = lambda n: pmc[n]["mean_per_dispatch"] if n in pmc else None
    o = {}
    if d.get("valu_lane_utilisation") is not None:
        o["lane_utilisation"] = round(d["valu_lane_utilisation"], 4)
    if d.get("wave_time_split"):
        o["wave_time_split"] = {k: round(v, 4) for k, v in d["wave_time_split"].items()}
    if g("SQ_INSTS_SALU") and g("SQ_INSTS_VALU"):
        o["salu_per_valu"] = round(g("SQ_INSTS_SALU") / g("SQ_INSTS_VALU"), 4)
    if g("GRBM_GUI_ACTIVE") and avg_ns:  # summed over the 8 XCDs: / 8 = busy clocks of the launch; / duration = the shader clock it ran at
        o["profiled_clock_mhz"] = round(g("GRBM_GUI_ACTIVE") / 8.0 / avg_ns * 1e3, 1)
    if g("SQ_LDS_IDX_ACTIVE") and g("GRBM_GUI_ACTIVE"):
        o["lds_busy"] = round(g("SQ_LDS_IDX_ACTIVE") / (g("GRBM_GUI_ACTIVE") / 8.0 * 256), 4)  # per-CU pipe cycles / (clocks x 256 CUs)
        if g("SQ_LDS_BANK_CONFLICT") is not None:
            o["lds_bank_conflict_share"] = round(g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"), 4)
    if d.get("l2_hit_rate") is not None:
        o["l2_hit_rate"] = round(d["l2_hit_rate"], 4)
    if d.get("write_bytes") is not None:
        o["write_bytes_per_launch"] = int(d["write_bytes"])
    return o


if tag == "--refill":  # tools/summarize_prof.py --refill <key> <profiles/xx_pmc.json>: add the pipe figures to an existing pmc_traffic.json entry
    tf = dst / "pmc_traffic.json"
    rec = json.loads(tf.read_text())
    prof = json.loads((ROOT / sys.argv[3]).read_text())
    rec[sys.argv[2]].update(pipe_figures(prof["pmc"], prof["derived"], prof["avg_ns"]))
    tf.write_text(json.dumps(rec, indent=1))
    print(json.dumps(rec[sys.argv[2]], indent=1))
    sys.exit(0)

stats = list(csv.DictReader(open(src / "trace_kernel_stats.csv")))
pick = sys.argv[4] if len(sys.argv) > 4 else None
suffix = ("_" + sys.argv[5]) if len(sys.argv) > 5 else ("_" + pick if pick else "")
main = max((r for r in stats if pick is None or pick in r["Name"]), key=lambda r: float(r["TotalDurationNs"]))


def clean_durations(trace_csv, name):
    """Durations (ns) of the launches of kernel `name` that did NOT overlap another frame kernel (trace_*) in time — rocprofv3's own
    kernel_stats averages every launch, and bench.py's buffer-pre-grow launches run three at a time (17.1 ms 'average'
    for a 9.06 ms kernel).  Returns (clean, dropped)."""
    rows = [r for r in csv.DictReader(open(trace_csv)) if "trace_" in r["Kernel_Name"]]
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
    clean, dropped = [], 0
    for i, (a, b, n) in enumerate(iv):
        if n != name:
            continue
        overlap = 0
        for j, (c, d, _) in enumerate(iv):
            if j != i:
                overlap = max(overlap, min(b, d) - max(a, c))
        if overlap > 0.02 * (b - a):
            dropped += 1
        else:
            clean.append(b - a)
    return clean, dropped


# rocprofv3's stats go in as they are, under a name that says so; the durations this repo quotes come from the clean launches only
shutil.copy(src / "trace_kernel_stats.csv", dst / f"{rnd}_{tag}_rocprof_kernel_stats_all_launches.csv")
trace_csv = src / "trace_kernel_trace.csv"
clean, dropped = clean_durations(trace_csv, main["Name"]) if trace_csv.exists() else ([], 0)
# the timed region of the trace pass: the last len(config.launches) launches (tools/gpu_profile.sh: REPS launches of STEPS frames)
timed = None
try:
    line = [l for l in open(src / "trace.bench.log").read().splitlines() if l.startswith("{") and "frames_per_launch_timed" in l][-1]
    cfg = json.loads(line)["config"]
    n_timed = len(cfg["launches"])
    if len(clean) >= n_timed and dropped + len(clean) >= n_timed:
        timed = clean[-n_timed:]
except Exception:
    cfg = None
durs = timed or clean
if durs:
    import statistics
    with open(dst / f"{rnd}_{tag}_trace_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "StdDev", "FramesPerLaunch", "DroppedOverlappingLaunches", "Source"])
        w.writerow([main["Name"], len(durs), sum(durs), sum(durs) / len(durs), min(durs), max(durs), statistics.pstdev(durs),
                    (cfg["launches"][0] if cfg and timed else ""), dropped,
                    "rocprofv3 --kernel-trace: launches of the timed region that overlap no other frame kernel (tools/summarize_prof.py)"])

pmc = {}
for f in sorted(src.glob("pmc_*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"] == main["Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")}
    for k, v in agg.items():
        pmc[k] = {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)}

out = {"kernel": main["Name"], "calls": len(durs), "avg_ns": (sum(durs) / len(durs) if durs else None), "min_ns": (min(durs) if durs else None),
       "max_ns": (max(durs) if durs else None),
       "duration_source": f"{len(durs)} launches of the trace pass's timed region that overlap no other frame kernel ({dropped} overlapping launches dropped); "
                          "the PMC passes serialise kernels and their per-dispatch counter means are unaffected",
       "frames_per_launch_of_durations": (cfg["launches"][0] if cfg and timed else None), "launch": meta,
       "launch_note": "as rocprofv3 prints it: VGPR_Count is in its own units (half the architectural count of these wave64 kernels) and "
                      "LDS_Block_Size is the STATIC group segment only (the kernels use dynamic LDS); the architectural register counts are in "
                      f"profiles/{rnd}_kernel_resources.json (tools/kernel_resources.py), the dynamic LDS bytes in bench.py's lds_bytes_per_block",
       "pmc": pmc}
# secondary kernels of the frame pipeline (blend_accumulate): duration + HBM bytes
others = {}
for r in stats:
    if r["Name"] != main["Name"] and "blend_accumulate" in r["Name"]:
        o = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"])}
        for f in sorted(src.glob("pmc_*_counter_collection.csv")):
            vals = collections.defaultdict(list)
            for row in csv.DictReader(open(f)):
                if row["Kernel_Name"] == r["Name"] and row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                    vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
            for k2, v2 in vals.items():
                o[k2 + "_KiB_mean"] = sum(v2) / len(v2)
        others["blend_accumulate"] = o
out["other_kernels"] = others
d = {}
g = lambda n: pmc[n]["mean_per_dispatch"] if n in pmc else None
if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
    # MI355X_MICROARCH.md §HBM: rocprofv3 units are KiB; on gfx950 FETCH_SIZE tallies 128-B requests of a wide
    # coalesced read at 64 B -> double it.  WRITE_SIZE is taken as reported.
    d["fetch_bytes_corrected"] = g("FETCH_SIZE") * 1024 * 2
    d["write_bytes"] = g("WRITE_SIZE") * 1024
    d["hbm_bytes_per_launch"] = d["fetch_bytes_corrected"] + d["write_bytes"]
if g("SQ_INSTS_VALU"):
    d["valu_wave_insts"] = g("SQ_INSTS_VALU")
    if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
        d["valu_lane_utilisation"] = g("SQ_THREAD_CYCLES_VALU") / (g("SQ_ACTIVE_INST_VALU") * 64)
if g("SQ_WAVE_CYCLES") and g("SQ_ACTIVE_INST_ANY"):
    d["wave_time_split"] = {"issuing": g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"),
                            "issue_stalled": g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"),
                            "waiting_on_counters": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")}
if g("TCC_HIT_sum") is not None:
    d["l2_hit_rate"] = g("TCC_HIT_sum") / max(1.0, g("TCC_HIT_sum") + g("TCC_MISS_sum"))
out["derived"] = d
(dst / f"{rnd}_{tag}{suffix}_pmc.json").write_text(json.dumps(out, indent=1))

# frames per launch of the profiled run: bench.py's own JSON line in the pass's log
frames_per_launch = None
for log in sorted(src.glob("pmc_*.bench.log")) + sorted(src.glob("trace.bench.log")):
    try:
        line = [l for l in open(log).read().splitlines() if l.startswith("{") and "frames_per_launch_timed" in l][-1]
        frames_per_launch = json.loads(line)["config"]["frames_per_launch_timed"]
        break
    except Exception:
        continue
tf = dst / "pmc_traffic.json"
rec = json.loads(tf.read_text()) if tf.exists() else {}
if "hbm_bytes_per_launch" in d and pick is None:
    sys.path.insert(0, str(ROOT))
    from rvpt_amd import build as rv_build
    rec[key] = {"hbm_bytes_per_launch": int(d["hbm_bytes_per_launch"]), "source": f"profiles/{rnd}_{tag}_pmc.json",
                "kernel_avg_ns": out["avg_ns"], "valu_wave_insts_per_launch": d.get("valu_wave_insts"),
                "frames_per_launch": frames_per_launch,  # bench.py scales the per-launch figures to its own launches
                **pipe_figures(pmc, d, out["avg_ns"]),   # what bench.py prints as the BVH lines' bound (roofline.bound = "valu_issue")
                # bench.py replays the figure only while the kernel sources still hash to this (the profile must be summarised on the
                # tree it was taken on)
                "kernel_sha": rv_build.kernel_sha()}
    tf.write_text(json.dumps(rec, indent=1))
print(json.dumps({"kernel": out["kernel"], "avg_us": (out["avg_ns"] / 1e3 if out["avg_ns"] else None), "clean_launches": len(durs), "dropped": dropped, **d}, indent=1))

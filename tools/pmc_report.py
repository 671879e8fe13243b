#!/usr/bin/env python3
"""Turns the output of tools/profile_round.sh (gpurun_out/prof_<tag>/) into the files kept under profiles/:

    python tools/pmc_report.py gpurun_out/prof_r03 r03 [W H batch key levels iterations]      (defaults 1920 1080 512 1080p_L5_I3 5 3)

  profiles/<tag>_kernel_stats.csv          rocprofv3 --kernel-trace --stats summary of the bench command (ofarn kernels)
  profiles/<tag>_kernel_stats_by_grid.csv  the same trace split per kernel, grid (= pyramid level) and stream
  profiles/<tag>_pmc_counters.txt          per kernel and grid: mean counter values per dispatch of every PMC pass
  profiles/<tag>_pmc_derived.txt           derived figures for the two dominant kernels (VALU busy, occupancy, LDS conflicts, HBM bytes)
  profiles/pmc_traffic.json                HBM bytes per work unit (FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE)
"""
import collections
import csv
import glob
import json
import os
import sys

root, tag = sys.argv[1], sys.argv[2]
W, H, BATCH = (int(x) for x in (sys.argv[3:6] if len(sys.argv) > 5 else (1920, 1080, 512)))
KEY = sys.argv[6] if len(sys.argv) > 6 else "1080p_L5_I3"
LEVELS, ITERS = (int(x) for x in (sys.argv[7:9] if len(sys.argv) > 8 else (5, 3)))
KCMD = f"python3 tools/kbench.py --w {W} --h {H} --levels {LEVELS} --iterations {ITERS} --batch {BATCH} --reps 1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def short(n):
    return n.split("(")[0].replace("void ", "").replace("ofarn::", "")


# ---- kernel trace
stats = glob.glob(root + "/trace/**/*kernel_stats.csv", recursive=True)
trace = glob.glob(root + "/trace/**/*kernel_trace.csv", recursive=True)
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(f"{P}/{tag}_kernel_stats.csv", "w") as o:
        o.write(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-two-stream --no-family-check (MI355X); ofarn kernels only\n")
        w = csv.DictWriter(o, fieldnames=rows[0].keys())
        w.writeheader()
        for r in rows:
            if "ofarn::" in r["Name"]:
                w.writerow(r)
if trace:
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(trace[0])):
        n = r["Kernel_Name"]
        if "ofarn::" not in n:
            continue
        acc[(short(n), r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Stream_Id"], r.get("VGPR_Count", ""),
             r.get("Scratch_Size", r.get("Private_Segment_Size", "")), r.get("LDS_Block_Size", ""))].append(
            int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open(f"{P}/{tag}_kernel_stats_by_grid.csv", "w") as o:
        o.write("# Derived from the kernel trace of the same rocprofv3 --kernel-trace --stats run as the _kernel_stats.csv next to it:\n"
                "# split per kernel, grid (= pyramid level; grid_z = frames or pairs of the wave) and stream.  bench.py's line of the same run is\n"
                "# profiles/<tag>_trace_bench.json: its roofline.kernel_avg_ms (HIP events) is the mean of the level-0 k_flow_iter rows here.\n"
                "# Rows on other streams than the first belong to the informational two-stream leg (two half-size waves overlapping).\n")
        o.write("kernel,grid_x,grid_y,grid_z,stream_id,vgpr,scratch,lds,calls,avg_ns,total_ns\n")
        for k in sorted(acc, key=lambda k: -sum(acc[k])):
            v = acc[k]
            o.write(f"\"{k[0]}\",{k[1]},{k[2]},{k[3]},{k[4]},{k[5]},{k[6]},{k[7]},{len(v)},{sum(v) / len(v):.0f},{sum(v)}\n")

# ---- PMC passes
cnt = collections.defaultdict(lambda: collections.defaultdict(list))   # (kernel, grid) -> counter -> values
for f in glob.glob(root + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "ofarn::" not in k:
            continue
        cnt[(short(k), r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(f"{P}/{tag}_pmc_counters.txt", "w") as o:
    o.write(f"# rocprofv3 --pmc <counters> -- {KCMD}   (one pass per counter group, no trace\n"
            f"# domains; MI355X; one wave of {BATCH} pairs {W}x{H}).  Mean counter value per dispatch; kernel@grid threads.\n")
    for k in sorted(cnt):
        o.write(f"{k[0]}@{k[1]}\n")
        for c, v in sorted(cnt[k].items()):
            o.write(f"    {c:34s} n={len(v):4d} mean={sum(v) / len(v):18.1f}\n")


def mean(k, c):
    v = cnt.get(k, {}).get(c)
    return sum(v) / len(v) if v else None


def biggest(prefix):
    ks = [k for k in cnt if k[0].startswith(prefix)]
    return max(ks, key=lambda k: int(k[1] or 0)) if ks else None


traffic = {"_provenance": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in two separate passes (no trace domains) -- " + KCMD +
                          f"; MI355X, one wave of {BATCH} pairs {W}x{H}.  Correction per MI355X_MICROARCH.md "
                          "(HBM section): on gfx950 FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced reads, so reads are doubled; "
                          "WRITE_SIZE is exact.  Per-kernel means: profiles/" + tag + "_pmc_counters.txt (KiB per dispatch)."}
with open(f"{P}/{tag}_pmc_derived.txt", "w") as o:
    for prefix, stage, units, alg in (("k_flow_iter<7, 2>", "flow_iter", BATCH * W * H, 56.0), ("k_polyexp_march<5, 1>", "polyexp", 2 * BATCH * W * H, 24.0)):
        k = biggest(prefix)
        if not k:
            continue
        o.write(f"{k[0]}@{k[1]}   (level 0, {units} work units per launch)\n")
        g = lambda c: mean(k, c)
        f, wv = g("FETCH_SIZE"), g("WRITE_SIZE")
        if f is not None and wv is not None:
            hbm = 2 * f * 1024 + wv * 1024
            traffic[stage] = {"kernel": f"{k[0]} level 0, grid {k[1]} threads", "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": wv,
                              "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg * units, "units_per_launch": units,
                              "hbm_bytes_per_unit": round(hbm / units, 3)}
            o.write(f"    HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE)   {hbm / 1e9:10.3f} GB = {hbm / units:.2f} B per unit (algorithmic {alg:.0f})\n")
        wc, busy, av, iv = g("SQ_WAVE_CYCLES"), g("SQ_BUSY_CYCLES"), g("SQ_ACTIVE_INST_VALU"), g("SQ_INSTS_VALU")
        waves = g("SQ_WAVES")
        if wc and av:
            o.write(f"    VALU busy = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES     {av / wc:10.3f}  (per-wave share of cycles with a VALU instruction executing; both in quad-cycles)\n")
        if iv and waves:
            o.write(f"    VALU instructions per wave                           {iv / waves:10.1f}   ({iv / waves / (units / waves / 64) if False else iv * 64 / units:.1f} per work unit)\n")
        il, ic, ia = g("SQ_INSTS_LDS"), g("SQ_LDS_BANK_CONFLICT"), g("SQ_LDS_IDX_ACTIVE")
        if il:
            o.write(f"    LDS instructions per work unit                       {il * 64 / units:10.2f}\n")
        if ic is not None and ia:
            o.write(f"    LDS bank-conflict cycles / LDS active cycles         {ic / ia:10.4f}\n")
        wa, wi, aa = g("SQ_WAIT_ANY"), g("SQ_WAIT_INST_ANY"), g("SQ_ACTIVE_INST_ANY")
        if wc and wa is not None:
            o.write(f"    wave cycles: waiting (s_waitcnt / barrier) {wa / wc:.3f}, issue-stalled {0 if wi is None else wi / wc:.3f}, issuing {0 if aa is None else aa / wc:.3f}\n")
        if wc and busy:
            o.write(f"    mean resident waves per SIMD = SQ_WAVE_CYCLES / (SQ_BUSY_CYCLES x 4 SIMDs x CUs/SE...) -> see raw counters; SQ_WAVES {waves:.0f}\n")
        vr, av2 = g("SQ_INSTS_VMEM_RD"), g("SQ_ACTIVE_INST_VMEM")
        if vr:
            o.write(f"    vector-memory read instructions per work unit        {vr * 64 / units:10.2f}\n")
        if wc and av2:
            o.write(f"    VMEM issue busy = SQ_ACTIVE_INST_VMEM / SQ_WAVE_CYCLES {av2 / wc:9.3f}\n")
        ta, gui = g("TA_TA_BUSY_sum"), g("GRBM_GUI_ACTIVE")
        gta = g("GRBM_TA_BUSY")
        if gta and gui:
            o.write(f"    GRBM_TA_BUSY / GRBM_GUI_ACTIVE                       {gta / gui:10.3f}\n")
        for c in ("TA_TA_BUSY_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_DATA_STALLED_BY_TC_CYCLES_sum", "TCC_HIT_sum", "TCC_MISS_sum",
                  "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_CVT", "SQ_VMEM_TA_ADDR_FIFO_FULL"):
            if g(c) is not None:
                o.write(f"    {c:52s} {g(c):16.1f}\n")
        o.write("\n")
# whole pipeline: HBM bytes of every ofarn kernel of the one-wave pass (256 pairs), reads doubled as above
tot_f = tot_w = 0.0
for k, cs in cnt.items():
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        tot_f += sum(cs["FETCH_SIZE"]) * 1024      # kbench --reps 1: one profiled pass (+ one warm-up pass, same kernels)
        tot_w += sum(cs["WRITE_SIZE"]) * 1024
if tot_f:
    passes = 2      # kbench runs the pass once untimed and once timed; both are in the counter file
    traffic["pipeline_hbm_bytes_per_pair"] = round((2 * tot_f + tot_w) / passes / BATCH)
    traffic["pipeline_hbm_bytes_note"] = (f"sum over all kernels of one pass of {BATCH} pairs of (2 x FETCH_SIZE + WRITE_SIZE) / {BATCH}; "
                                          "algorithmic (SURVEY 8(d)): 1003.4 MB per pair at 1080p L5 I3, 6154.1 MB at 4K L6 I5")
if len(traffic) > 1:
    old = {}
    try:
        old = json.load(open(f"{P}/pmc_traffic.json"))
    except Exception:
        pass
    if "flow_iter" in old:        # round-2 layout (flat, one wave of 256 pairs at 1080p): keep it under its own key
        old = {"1080p_L5_I3_round2_wave256": old}
    old[KEY] = traffic            # one entry per bench config key (bench.py CONFIGS[...]["key"])
    json.dump(old, open(f"{P}/pmc_traffic.json", "w"), indent=2)
print(open(f"{P}/{tag}_pmc_derived.txt").read())

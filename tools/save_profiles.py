#!/usr/bin/env python3
"""Copies what a round measured on the GPU box from gpurun_out/ (scratch) into profiles/ (tracked; what the judge reads).

    python tools/save_profiles.py r03

  gpurun_out/evidence_<tag>/*          (tools/collect_evidence.sh: every number DESIGN.md section 7 quotes besides the headline, each
                                        file starting with its command line)                      -> profiles/<tag>_<name>
  gpurun_out/prof_<tag>/, prof_<tag>_4k (tools/profile_round.sh: kernel trace + PMC passes)         -> via tools/pmc_report.py:
                                        profiles/<tag>_kernel_stats*.csv, <tag>_pmc_*.txt, pmc_traffic.json, <tag>_trace_bench.json
  gpurun_out/<tag>_bench_final.json/.err (python bench.py --steps 20 --warmup 5 --prof-table)      -> profiles/<tag>_bench_final.json,
                                        profiles/<tag>_bench_final_stage_table.txt
"""
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
n = 0
for f in sorted(glob.glob(os.path.join(G, f"evidence_{tag}", "*"))):
    shutil.copy(f, os.path.join(P, f"{tag}_{os.path.basename(f)}"))
    n += 1
for src, dst in ((f"{tag}_bench_final.json", f"{tag}_bench_final.json"),):
    if os.path.exists(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(P, dst))
        n += 1
err = os.path.join(G, f"{tag}_bench_final.err")
if os.path.exists(err):
    open(os.path.join(P, f"{tag}_bench_final_stage_table.txt"), "w").write(
        "".join(line for line in open(err) if "amdgpu.ids" not in line))
    n += 1
if os.path.isdir(os.path.join(G, f"prof_{tag}")):
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_report.py"), os.path.join(G, f"prof_{tag}"), tag], check=True,
                   stdout=subprocess.DEVNULL)
    tb = os.path.join(G, f"prof_{tag}", "trace_bench.json")
    if os.path.exists(tb):
        shutil.copy(tb, os.path.join(P, f"{tag}_trace_bench.json"))
    n += 1
if os.path.isdir(os.path.join(G, f"prof_{tag}_4k")):
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_report.py"), os.path.join(G, f"prof_{tag}_4k"), f"{tag}_4k", "3840", "2160", "64",
                    "4k_L6_I5", "6", "5"], check=True, stdout=subprocess.DEVNULL)
    n += 1
print(f"{n} item(s) copied into profiles/ for {tag}")

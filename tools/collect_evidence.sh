#!/bin/bash
# Runs on the GPU box (via gpurun): every measurement DESIGN.md section 7 quotes besides the headline bench line, each with its
# command line in front of its output, into gpurun_out/evidence_<tag>/ (copy what is to be kept into profiles/<tag>_*.txt|json).
#   bash tools/collect_evidence.sh r03
set -u
TAG=${1:-r03}
OUT=gpurun_out/evidence_$TAG
mkdir -p $OUT
run() {  # file, command...
    local f=$1; shift
    { echo "\$ $*"; timeout -k 10 400 "$@" 2>&1 | grep -v amdgpu.ids; echo "[exit ${PIPESTATUS[0]}]"; } > $OUT/$f
    tail -2 $OUT/$f
}
run consec_bench.txt python3 tools/consec_bench.py
run hostbench.txt python3 tools/hostbench.py 64
run lkbench.txt python3 tools/lkbench.py --batch 128
run kbench_gaussian_flag.txt python3 tools/kbench.py --flags 256 --batch 256 --reps 3
run kbench_box_same_run.txt python3 tools/kbench.py --batch 256 --reps 3
run kbench_winsize21.txt python3 tools/kbench.py --winsize 21 --batch 128 --reps 3
run kbench_640x480_L3_batch1024.txt python3 tools/kbench.py --w 640 --h 480 --levels 3 --batch 1024 --reps 3
run streamprof.txt python3 tools/streamprof.py
run bench_config2.json python3 bench.py --config 2 --steps 40 --warmup 5
run bench_config2_stream.json python3 bench.py --config 2 --stream --steps 40 --warmup 5
run bench_config4_1gpu.json python3 bench.py --config 4 --steps 5 --warmup 2 --cpu-sample 0 --no-family-check
run bench_config5_1gpu.json python3 bench.py --config 5 --steps 5 --warmup 2 --cpu-sample 0 --no-family-check
run bench_config4_rccl_ws1.json python3 bench.py --config 4 --gpus 1 --backend nccl --force-dist --steps 5 --warmup 2 --cpu-sample 0 --no-family-check
run bench_config5_rccl_ws1.json python3 bench.py --config 5 --gpus 1 --backend nccl --force-dist --steps 5 --warmup 2 --cpu-sample 0 --no-family-check
run bench_inproc_config3.json python3 bench.py --multi inproc --gpus 1 --config 3 --steps 5 --warmup 2 --no-family-check
run bench_inproc_config4.json python3 bench.py --multi inproc --gpus 1 --config 4 --steps 5 --warmup 2 --cpu-sample 0
run bench_inproc_config5.json python3 bench.py --multi inproc --gpus 1 --config 5 --steps 5 --warmup 2 --cpu-sample 0
run bench_ranks_config3_same_box.json python3 bench.py --config 3 --steps 5 --warmup 2 --cpu-sample 0 --no-family-check --no-two-stream
(cd tools/microbench && for b in hbm_rw march_layout valu_rates; do [ -x ./$b ] && { echo "\$ ./$b"; timeout -k 10 200 ./$b; } > ../../$OUT/$b.txt 2>&1; done)
ls -la $OUT

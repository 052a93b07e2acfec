#!/bin/bash
# run on the GPU box from the repo root: kernel stats + three PMC passes over the dominant kernel + a bench line
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
TAG=$1
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 > "$OUT/bench.json" 2> "$OUT/bench.err" || exit 2
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-predict > "$OUT/stats.log" 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --kernel-only --steps 10 > "$OUT/pmc1.log" 2>&1 || exit 4
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 bench.py --kernel-only --steps 10 > "$OUT/pmc2.log" 2>&1 || exit 5
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq" -- python3 bench.py --kernel-only --steps 10 > "$OUT/pmc3.log" 2>&1 || exit 6
python3 tools/pmc_summary.py "$OUT" > "$OUT/pmc_vocab_stream.json" || exit 7
find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
# keep what is merged back small
find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*kernel_trace.csv" -delete
cat "$OUT/bench.json"

#!/bin/bash
# run on the GPU box from the repo root:  bash tools/collect_profiles.sh <tag> [c2|c4]
# kernel stats of the bench step + three separate PMC passes over the dominant kernel + (c2 only) the default bench line
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
TAG=$1
CFG=${2:-c2}
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
if [ "$CFG" = c2 ]; then
  timeout -k 10 400 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || exit 2
else
  timeout -k 10 300 python3 bench.py --config "$CFG" --no-extras --no-predict --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err" || exit 2
fi
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --config "$CFG" --steps 50 --warmup 5 --no-cpu-baseline --no-predict --no-extras > "$OUT/stats.log" 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --config "$CFG" --kernel-only --steps 10 > "$OUT/pmc1.log" 2>&1 || exit 4
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 bench.py --config "$CFG" --kernel-only --steps 10 > "$OUT/pmc2.log" 2>&1 || exit 5
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq" -- python3 bench.py --config "$CFG" --kernel-only --steps 10 > "$OUT/pmc3.log" 2>&1 || exit 6
python3 tools/pmc_summary.py "$OUT" "$CFG" bf16 > "$OUT/pmc_vocab_stream.json" || exit 7
find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
# keep what is merged back small
find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*kernel_trace.csv" -delete
cat "$OUT/bench.json"

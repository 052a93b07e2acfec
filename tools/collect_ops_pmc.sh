#!/bin/bash
# run on the GPU box from the repo root:  bash tools/collect_ops_pmc.sh <tag> [c2|c4]
# one kernel-trace pass and three separate PMC passes (FETCH_SIZE | WRITE_SIZE | SQ_* + GRBM) over every op of a step, launched once each
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
TAG=$1
CFG=${2:-c4}
OUT=gpurun_out/ops_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -s KILL 150 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 tools/pmc_ops.py run "$CFG" "$OUT/order.json" > "$OUT/run0.log" 2>&1 || exit 3
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 tools/pmc_ops.py run "$CFG" "$OUT/order.json" > "$OUT/run1.log" 2>&1 || exit 4
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 tools/pmc_ops.py run "$CFG" "$OUT/order.json" > "$OUT/run2.log" 2>&1 || exit 5
timeout -s KILL 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq" -- python3 tools/pmc_ops.py run "$CFG" "$OUT/order.json" > "$OUT/run3.log" 2>&1 || exit 7
python3 tools/pmc_ops.py parse "$OUT/order.json" "$OUT/trace" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_sq" > "$OUT/ops_pmc.json" || exit 6
find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete

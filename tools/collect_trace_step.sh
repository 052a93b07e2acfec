cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/trace_c2; mkdir -p gpurun_out/trace_c2
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_c2 -- python3 bench.py --config c2 --steps 100 --warmup 5 --no-cpu-baseline --no-predict --no-extras > gpurun_out/trace_c2/run.log 2>&1 || exit 3
f=$(find gpurun_out/trace_c2 -name "*kernel_trace.csv" | head -1)
python3 tools/trace_step.py "$f" > gpurun_out/trace_c2/trace_step.txt || exit 4
find gpurun_out/trace_c2 -name "*kernel_trace.csv" -delete
cat gpurun_out/trace_c2/trace_step.txt

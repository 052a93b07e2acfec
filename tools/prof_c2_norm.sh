cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for m in 1 0; do
  export C1V_FUSE_NORM=$m
  timeout -k 10 150 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c2_n$m --output-format csv -- python3 bench.py --config c2 --steps 200 --no-cpu > gpurun_out/prof_c2_n$m.log 2>&1 || exit 1
done

#!/bin/bash
# bash tools/c4_env_scan.sh "VAR=a VAR=b ..."   -> ms/step of bench.py --config c4 under each setting (one box, back to back)
run() {
  env $1 timeout -k 10 200 python bench.py --config c4 --no-extras --no-predict --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'ms/step %.4f' % d['ms_per_step'], 'gemm us %.0f' % sum(r['us'] for r in d.get('gemm_rooflines', [])))" || exit 1
}
for s in $1; do run "$s"; done

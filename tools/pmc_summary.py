import csv, glob, json, sys
out = sys.argv[1]
cfg = sys.argv[2] if len(sys.argv) > 2 else 'c2'
dtype = sys.argv[3] if len(sys.argv) > 3 else 'bf16'
acc = {}
for d in ('pmc_fetch', 'pmc_write', 'pmc_sq'):
    for f in glob.glob('%s/%s/**/*counter_collection.csv' % (out, d), recursive=True):
        for r in csv.DictReader(open(f)):
            if 'vocab_stream_kernel' not in r['Kernel_Name']:
                continue
            key = (r['Counter_Name'], r['Dispatch_Id'])
            acc[key] = acc.get(key, 0.0) + float(r['Counter_Value'])
per = {}
for (name, disp), v in acc.items():
    per.setdefault(name, []).append(v)
res = {'kernel': 'vocab_stream_kernel<10,ACCUM>', 'config': cfg, 'dtype': dtype, 'per_launch_avg': {k: sum(v) / len(v) for k, v in sorted(per.items())},
       'launches': {k: len(v) for k, v in sorted(per.items())},
       'note': 'separate rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_*+GRBM) over `bench.py --kernel-only --steps 10`; '
               'values summed over the dimensions (XCDs / SEs) of a dispatch, averaged over the dispatches'}
print(json.dumps(res, indent=1))

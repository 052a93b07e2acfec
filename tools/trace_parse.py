import csv, glob, sys
f=glob.glob('gpurun_out/trace/**/*kernel_trace.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last step: find last occurrence of philox_multi
idx=[i for i,r in enumerate(rows) if 'step_prologue' in r['Kernel_Name']]
i0=idx[-1]
t0=int(rows[i0]['Start_Timestamp'])
prev_end=t0
for r in rows[i0:]:
    s=int(r['Start_Timestamp'])-t0; e=int(r['End_Timestamp'])-t0
    print('%8.1f %8.1f %6.1f gap %6.1f q%s %s'%(s/1e3,e/1e3,(e-s)/1e3,(s-prev_end)/1e3,r.get('Queue_Id','?'),r['Kernel_Name'][:60]))
    prev_end=max(prev_end,e)

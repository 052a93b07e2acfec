"""One steady-state training step out of a rocprofv3 --kernel-trace CSV: every launch in order with its median duration and
the median gap to the launch before it (the steps of the timed region are replays of one graph, so position k of every step
is the same kernel).  usage: python tools/trace_step.py <kernel_trace.csv> [prologue grid size]"""
import csv, sys, collections
import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
gridk = 'Grid_Size_X' if 'Grid_Size_X' in rows[0] else 'Grid_Size'
# steps start at the prologue kernel; keep the most common (grid, step length) = the timed region of the headline config
starts = [i for i, n in enumerate(names) if n.startswith('step_prologue_kernel')]
want = sys.argv[2] if len(sys.argv) > 2 else collections.Counter(rows[i][gridk] for i in starts).most_common(1)[0][0]
starts = [i for i in starts if rows[i][gridk] == want]
lens = collections.Counter(b - a for a, b in zip(starts, starts[1:]))
L = lens.most_common(1)[0][0]
steps = [a for a, b in zip(starts, starts[1:]) if b - a == L and [names[a + k] for k in range(L)] == [names[starts[1] + k] for k in range(L)]]
dur = np.array([[int(rows[a + k]['End_Timestamp']) - int(rows[a + k]['Start_Timestamp']) for k in range(L)] for a in steps], dtype=np.float64)
gap = np.array([[int(rows[a + k]['Start_Timestamp']) - int(rows[a + k - 1]['End_Timestamp']) for k in range(L)] for a in steps], dtype=np.float64)
print('%d steps of %d launches (prologue grid %s); medians in us' % (len(steps), L, want))
print('%3s %-78s %9s %8s %8s' % ('#', 'kernel', 'grid', 'dur', 'gap'))
for k in range(L):
    r = rows[steps[0] + k]
    print('%3d %-78s %9s %8.2f %8.2f' % (k, r['Kernel_Name'][:78], r[gridk], np.median(dur[:, k]) / 1e3, np.median(gap[:, k]) / 1e3))
print('sum of durations %.2f us, sum of gaps (incl. step-to-step) %.2f us, step period %.2f us' % (
    np.median(dur.sum(1)) / 1e3, np.median(gap.sum(1)) / 1e3, np.median(np.diff([int(rows[a]['Start_Timestamp']) for a in steps])) / 1e3))

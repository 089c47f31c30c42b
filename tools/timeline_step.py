"""Kernel timeline of one replayed train step from a rocprofv3 kernel trace CSV: start offset, duration, queue, kernel, grid.
Usage: python tools/timeline_step.py <kernel_trace.csv> [out.txt]
Caveat: under rocprofv3 the nodes of a hipGraph are submitted one after the other (~14 us apiece), so a parallel branch appears to
start 0.5-1.2 ms after its fork point; without the profiler the issue order of the branches makes no measurable difference."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_dev' in r['Kernel_Name']]
k = len(idx) // 2; k -= k % 2
seg = rows[idx[k] + 1: idx[k + 2] + 1]
t0 = int(seg[0]['Start_Timestamp'])
out = open(sys.argv[2], 'w') if len(sys.argv) > 2 else sys.stdout
for r in seg:
    n = r['Kernel_Name'].replace('void ', '')
    s = (int(r['Start_Timestamp']) - t0) / 1e3; d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    out.write("%8.1f %6.1f q%-3s %-44s %s\n" % (s, d, r['Queue_Id'], n[:44], r.get('Grid_Size_X', r.get('Grid_Size', ''))))

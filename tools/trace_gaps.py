"""Idle-gap report for one replayed train step from a rocprofv3 kernel trace CSV.
Usage: python tools/trace_gaps.py <kernel_trace.csv> [min_gap_us]"""
import csv, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 3000
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_dev' in r['Kernel_Name']]
# one step = two adam_dev launches (G, D); take the step of median duration among the replayed ones
steps = [(int(rows[idx[i + 2]]['End_Timestamp']) - int(rows[idx[i]]['End_Timestamp']), idx[i], idx[i + 2]) for i in range(0, len(idx) - 2, 2)]
steps.sort()
_, a, b = steps[len(steps) // 2]
seg = rows[a + 1:b + 1]
t0 = int(seg[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in seg)
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
ce = int(seg[0]['End_Timestamp']); prev = seg[0]; idle = 0; out = []
for r in seg[1:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s > ce:
        idle += s - ce
        if s - ce > thr: out.append("%9.1fus gap %6.1fus  after %-40s before %-40s" % ((ce - t0) / 1e3, (s - ce) / 1e3, prev['Kernel_Name'][:40], r['Kernel_Name'][:40]))
    if e > ce: ce = e; prev = r
print("kernels %d  wall %.3f ms  sum-of-kernels %.3f ms  idle %.3f ms" % (len(seg), (t1 - t0) / 1e6, busy / 1e6, idle / 1e6))
print("\n".join(out))

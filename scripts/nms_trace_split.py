"""Per-kernel durations of one ccnet_nms call from a rocprofv3 kernel trace:  python scripts/nms_trace_split.py trace.csv [call ...]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'nms_' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
calls, cur = [], None
for r in rows:
    name = r['Kernel_Name'].split('(anonymous namespace)::')[-1].split('(')[0]
    if 'bucket' in name:
        cur = []
        calls.append(cur)
    if cur is not None:
        cur.append((name, int(r['Start_Timestamp']), int(r['End_Timestamp'])))
for ci in [int(a) for a in sys.argv[2:]] or range(len(calls)):
    c = calls[ci]
    print('call %d: span %.1f us' % (ci, (c[-1][2] - c[0][1]) / 1e3), '  '.join('%s %.1f' % (n.replace('nms_', '').replace('_kernel', ''), (e - s) / 1e3) for n, s, e in c))

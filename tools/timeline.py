"""Print one steady-state cycle of a rocprofv3 --kernel-trace CSV as a timeline (start, end, duration, name)."""
import csv, re, sys, glob
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    n = re.sub(r'paac::', '', n)
    m = re.match(r'void dmm_kernel<Geom<([0-9, ]+)>, (true|false), (\d), (\d), (\d), (\d), (\d), (\d), (\d), (\d+), (\d), (true|false), (\d)>', n)
    if m:
        return 'dmm G%s ap%s bp%s T%sx%s W%s,%s,%s epi%s pf%s' % (m.group(1).replace(' ', ''), m.group(3), m.group(4), m.group(5), m.group(6), m.group(7), m.group(8), m.group(9), m.group(11), m.group(13))
    return n[:48]
idx = [i for i, r in enumerate(rows) if 'copyBuffer' in r['Kernel_Name']]
i0, i1 = idx[-3], idx[-2]
t0 = int(rows[i0]['Start_Timestamp'])
print('kernels per cycle:', i1 - i0, ' cycle wall us: %.1f' % ((int(rows[i1]['Start_Timestamp']) - t0) / 1000))
busy = 0
for r in rows[i0:i1]:
    s = int(r['Start_Timestamp']) - t0; e = int(r['End_Timestamp']) - t0
    busy += e - s
    print('%8.1f -> %8.1f  dur %6.1f  grid %6d x%4d  %s' % (s / 1000, e / 1000, (e - s) / 1000, int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) // int(r['Workgroup_Size_X']), int(r['Workgroup_Size_X']), short(r['Kernel_Name'])))
print('sum of kernel durations us: %.1f' % (busy / 1000))

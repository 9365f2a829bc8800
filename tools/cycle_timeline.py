"""One steady-state cycle out of a rocprofv3 --kernel-trace csv, with the idle gap before every kernel.
usage: python tools/cycle_timeline.py <dir with *_kernel_trace.csv> [out.txt]"""
import csv, glob, os, re, sys

src = sys.argv[1]
trace = glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'rmsprop_kernel' in r['Kernel_Name']]
k0 = len(marks) // 3


def wall(k):
    return int(rows[marks[k + 1] + 1]['Start_Timestamp']) - int(rows[marks[k] + 1]['Start_Timestamp'])


k = min(range(k0, min(k0 + 8, len(marks) - 2)), key=wall)
i0, i1 = marks[k] + 1, marks[k + 1] + 1
t0 = int(rows[i0]['Start_Timestamp'])
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
out.write("# start_us dur_us gap_before_us  workgroups x threads  kernel\n")
prev_end = int(rows[i0 - 1]['End_Timestamp'])
gaps = 0.0
for r in rows[i0:i1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    wg = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) // int(r['Workgroup_Size_X'])
    name = re.sub(r'paac::', '', r['Kernel_Name']).split('(')[0][:60]
    gap = (s - prev_end) / 1000
    gaps += max(gap, 0.0)
    out.write("%8.1f %6.1f %6.1f  %6d x %3d  %s\n" % ((s - t0) / 1000, (e - s) / 1000, gap, wg, int(r['Workgroup_Size_X']), name))
    prev_end = max(prev_end, e)
out.write("# cycle wall %.1f us, %d kernels, idle gaps %.1f us\n" % ((int(rows[i1]['Start_Timestamp']) - t0) / 1000, i1 - i0, gaps))

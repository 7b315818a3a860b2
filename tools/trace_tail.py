"""development aid: the kernels of one step of the headline bench from the end of Dither on (Reconstruct, Reindex), in launch order, runs of
the same kernel folded:   python tools/trace_tail.py gpurun_out/prof_<tag> [step index]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ev = [(r['Kernel_Name'].split('(')[0][:60], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
loads = [i for i, e in enumerate(ev) if 'k_load_tiles' in e[0]]
starts = [loads[0]]
for a, b in zip(loads, loads[1:]):
    if ev[b][1] - ev[a][2] > 5e6:
        starts.append(b)
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
step = ev[starts[k]:starts[k + 1]]
di = [i for i, e in enumerate(step) if 'k_dd_lookup' in e[0] or 'k_dither_tk' in e[0]]
i0 = di[-1] + 1
t0 = step[i0][1]
out = []
for n, a, b in step[i0:]:
    if out and out[-1][0] == n:
        out[-1][1] += (b - a) / 1e3
        out[-1][2] += 1
    else:
        out.append([n, (b - a) / 1e3, 1, a])
for n, d, c, a in out:
    print('%8.2f  %-60s %8.1f us x%d' % ((a - t0) / 1e6, n, d, c))

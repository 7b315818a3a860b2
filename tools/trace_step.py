"""One step of the headline bench out of a rocprofv3 kernel trace: per-kernel time, launch counts and idle gaps.
    python tools/trace_step.py gpurun_out/prof_<tag> [step index]"""
import csv
import glob
import sys
from collections import defaultdict

f = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ev = [(r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
loads = [i for i, e in enumerate(ev) if 'k_load_tiles' in e[0]]
starts = [loads[0]]
for a, b in zip(loads, loads[1:]):
    if ev[b][1] - ev[a][2] > 5e6:
        starts.append(b)
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
step = ev[starts[k]:starts[k + 1]]
span = (step[-1][2] - step[0][1]) / 1e6
busy = sum(e[2] - e[1] for e in step) / 1e6
# union of busy intervals (two streams overlap)
iv = sorted((a, b) for _, a, b in step)
u, cur_a, cur_b = 0, iv[0][0], iv[0][1]
for a, b in iv[1:]:
    if a > cur_b:
        u += cur_b - cur_a
        cur_a, cur_b = a, b
    else:
        cur_b = max(cur_b, b)
u += cur_b - cur_a
print('steps found %d; step %d: span %.2f ms, sum of kernels %.2f ms, GPU busy (union) %.2f ms, launches %d' % (len(starts), k, span, busy, u / 1e6, len(step)))
d = defaultdict(lambda: [0, 0])
for n, a, b in step:
    key = n.split('(')[0][:64]
    d[key][0] += (b - a) / 1e6
    d[key][1] += 1
for key, v in sorted(d.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print('%-66s %7.3f ms %4d' % (key, v[0], v[1]))
gaps = []
end = step[0][2]
for n, a, b in step[1:]:
    if a - end > 50e3:
        gaps.append(((a - end) / 1e6, n.split('(')[0][:50]))
    end = max(end, b)
print('idle gaps > 50 us: %.2f ms in %d gaps' % (sum(g[0] for g in gaps), len(gaps)))
for g in sorted(gaps, reverse=True)[:12]:
    print('  %.3f ms before %s' % g)

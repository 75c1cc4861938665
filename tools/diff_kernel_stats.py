"""Per-kernel difference of two rocprofv3 kernel_stats.csv files (ms per training step):
python tools/diff_kernel_stats.py a.csv b.csv [steps]"""
import csv
import sys


def load(f):
    return {r['Name']: (int(r['Calls']), float(r['TotalDurationNs']) / 1e6) for r in csv.DictReader(open(f))}


a, b = load(sys.argv[1]), load(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
names = sorted(set(a) | set(b), key=lambda n: -max(a.get(n, (0, 0))[1], b.get(n, (0, 0))[1]))
print("%-72s %14s %14s" % ("kernel (ms per step, calls)", "first", "second"))
ta = tb = 0
for n in names:
    ca, ma = a.get(n, (0, 0))
    cb, mb = b.get(n, (0, 0))
    ta += ma
    tb += mb
    if abs(ma - mb) / steps > 0.01:
        print("%-72s %6.3f (%5d) %6.3f (%5d)  %+.3f" % (n[:72], ma / steps, ca, mb / steps, cb, (ma - mb) / steps))
print("total %.3f %.3f" % (ta / steps, tb / steps))

"""Sums the [amg phase] / [amg chase] lines that ORC_AMG_TRACE=1 writes to stderr: wall ms per phase and level."""
import collections
import re
import sys

t = collections.defaultdict(float)
ev = collections.defaultdict(list)
for line in open(sys.argv[1]):
    m = re.match(r"\[amg phase n=(\d+)\] (.*) ([\d.]+) ms", line)
    if m:
        for k in ((m.group(1), m.group(2)), ("all", m.group(2)), ("all", "all")):
            t[k] += float(m.group(3))
    m = re.match(r"\[amg chase n=(\d+)\] launches (\d+) evaluations (\d+)", line)
    if m:
        ev[m.group(1)].append(int(m.group(3)))
for k in sorted(t):
    print("%-10s %-24s %9.1f ms" % (k[0], k[1], t[k]))
for k in ev:
    print("n=%s: %d aggregations, %d evaluations each" % (k, len(ev[k]), sum(ev[k]) // len(ev[k])))

"""Summarise `[rvm tune]` lines (RVM_VERBOSE=1): per tunable key the best candidate and every candidate's time."""
import collections
import re
import sys

d = collections.OrderedDict()
for l in open(sys.argv[1]):
    m = re.match(r'\[rvm tune\] (\S+)\s+cand (\d+)\s+([\d.]+) us', l)
    if m:
        d.setdefault(m.group(1), {})[int(m.group(2))] = float(m.group(3))
pat = sys.argv[2] if len(sys.argv) > 2 else ''
for k, v in d.items():
    if re.search(pat, k):
        best = min(v, key=v.get)
        print('%-18s best %2d %6.2f | ' % (k, best, v[best]) + ' '.join('%d:%.1f' % (c, t) for c, t in v.items()))

import json, sys
k = json.load(open(sys.argv[1]))
base = json.load(open(sys.argv[2])) if len(sys.argv) > 2 else None
tot = sum(x['ms'] for x in k)
print('total %.3f ms, %d launches' % (tot, len(k)))
groups = {}
for i, x in enumerate(k):
    b = ''
    if base and i < len(base) and base[i]['name'] == x['name']:
        b = '  (was %6.1f)' % (base[i]['ms'] * 1e3)
    print("  %-22s %7.1f us%s  %8.2f MFLOP %8.3f MB  %8.1f GFLOP/s %7.1f GB/s" % (
        x['name'], x['ms'] * 1e3, b, x['flops'] / 1e6, x['bytes'] / 1e6, x['flops'] / x['ms'] / 1e6, x['bytes'] / x['ms'] / 1e6))
    key = x['name'].split('.')[-1] if x['name'].startswith('mb') else x['name'].split('.')[0]
    groups[key] = groups.get(key, 0) + x['ms']
print({k_: round(v * 1e3, 1) for k_, v in sorted(groups.items(), key=lambda t: -t[1])})

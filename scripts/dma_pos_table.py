import re, collections, sys
d = collections.defaultdict(lambda: collections.defaultdict(list))
lib = None
for l in open(sys.argv[1]):
    if l.startswith('=='):
        lib = l.split()[1].replace('libdm_amd', '').replace('.so', '') or 'base'; continue
    m = re.match(r'\s*(.*?3x3)\s+.*?([\d.]+)\s*us', l)
    if m: d[m.group(1)][lib].append(float(m.group(2)))
libs = sorted({k for v in d.values() for k in v})
print(f"{'shape':22s}" + "".join(f"{k:>10s}" for k in libs))
for shape, v in d.items():
    print(f"{shape:22s}" + "".join(f"{min(v[k]):10.1f}" for k in libs))

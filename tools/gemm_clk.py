import re,collections,sys
for f in sys.argv[1:]:
    d=collections.defaultdict(list)
    for l in open(f):
        m=re.search(r"zout (\d) K (\d+) block \d+: tiles (\d+), cycles per tile: total (\d+), barriers (\d+), epilogue (\d+) len (\d+) st (\d+)",l)
        if m: d[(m.group(1),m.group(2))].append(tuple(int(x) for x in m.groups()[2:]))
    print(f)
    for k,v in sorted(d.items()):
        n=len(v); print("  zout %s K %s: %d samples, tiles %.1f, per tile total %.0f barriers %.0f epilogue %.0f (len %.0f st %.0f)"%((k[0],k[1],n)+tuple(sum(x[i] for x in v)/n for i in range(6))))

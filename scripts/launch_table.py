"""Group a bench.py --dump-launches file by (entry point, algorithmic work) and print time / rate per group."""
import collections, sys
g = collections.defaultdict(lambda: [0, 0.0])
for l in open(sys.argv[1]):
    n, f, b, ms, *sh = l.split()
    k = (n + (" " + sh[0] if sh and sh[0] != "-" else ""), float(f), float(b))
    g[k][0] += 1
    g[k][1] += float(ms)
rows = sorted(g.items(), key=lambda kv: -kv[1][1])
print("total %.2f ms" % sum(v[1] for v in g.values()))
for (n, f, b), (c, ms) in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    rate = f / (ms / c) * 1e-9 if f else b / (ms / c) * 1e-6
    print(f"{n:58s} x{c:4d} {ms:8.2f} ms  avg {ms / c * 1e3:8.1f} us  {f * 1e-9:8.1f} GF {b * 1e-6:8.1f} MB  -> {rate:8.1f} {'TF/s' if f else 'GB/s'}")

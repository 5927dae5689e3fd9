import sys, json
a = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); b = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
ka, kb = a["kernels"], b["kernels"]
rows = []
for k in sorted(set(ka) | set(kb)):
    ta, tb = ka.get(k, {}).get("total_ms", 0), kb.get(k, {}).get("total_ms", 0)
    rows.append((ta - tb, k, ta, tb))
for d, k, ta, tb in sorted(rows, reverse=True)[:10] + sorted(rows)[:5]:
    print(f"{k[:60]:60s} noflag {ta:8.3f}  flag {tb:8.3f}  diff {d:+.3f}")
print(a["kernels_from"])

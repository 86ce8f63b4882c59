"""Markdown table of a parity record (profiles/rNN_parity.jsonl: one JSON object per parity test, written by tests/parity.py on the GPU box):
eps_x (in h) / eps_v (relative to max |v|) of the HIP path against the f32 oracle (the reference's own arithmetic), against the f64 oracle (the
gate's arbiter), and of the f32 oracle against the f64 one.      python tools/parity_table.py profiles/r05_parity.jsonl [substring ...]"""
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip()]
want = sys.argv[2:]
f = lambda d, k: "%.1e" % d[k] if d and k in d else "—"
print("| case | HIP vs f32 oracle ε_x / ε_v | HIP vs f64 oracle ε_x / ε_v | f32 oracle vs f64 oracle ε_x / ε_v |")
print("|---|---|---|---|")
for r in rows:
    if "vs_f64" not in r or (want and not any(w in r["test"] for w in want)): continue
    a, b, c = r.get("vs_f32"), r.get("vs_f64"), r.get("oracle_f32_vs_f64")
    print(f"| `{r['test']}` | {f(a, 'eps_x')} / {f(a, 'eps_v')} | {f(b, 'eps_x')} / {f(b, 'eps_v')} | {f(c, 'eps_x')} / {f(c, 'eps_v')} |")

"""Markdown table of a parity record (profiles/rNN_parity.jsonl: one JSON object per parity test, written by tests/parity.py on the GPU box):
eps_x (in h) / eps_v (relative to max |v|) of the HIP path against the f32 oracle (the reference's own arithmetic), against the f64 oracle (the
gate's arbiter), and of the f32 oracle against the f64 one; with --all also the quantities the gate asserts since round 6 (F, C, plastic hardening,
log_vol_gain, det Fp, psi_pos; HIP vs the f64 oracle).      python tools/parity_table.py profiles/r06_parity.jsonl [--all] [substring ...]"""
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip()]
args = sys.argv[2:]
full = "--all" in args
want = [a for a in args if a != "--all"]
f = lambda d, k: "%.1e" % d[k] if d and k in d else "—"
extra = ("eps_F", "eps_C", "eps_hard", "eps_lvg", "eps_Jp", "eps_psi")
print("| case | HIP vs f32 oracle ε_x / ε_v | HIP vs f64 oracle ε_x / ε_v | f32 oracle vs f64 oracle ε_x / ε_v |" + (" HIP vs f64: ε_F / ε_C / hardening / log_vol_gain / det Fp / psi_pos |" if full else ""))
print("|---|---|---|---|" + ("---|" if full else ""))
for r in rows:
    if "vs_f64" not in r or (want and not any(w in r["test"] for w in want)): continue
    a, b, c = r.get("vs_f32"), r.get("vs_f64"), r.get("oracle_f32_vs_f64")
    print(f"| `{r['test']}` | {f(a, 'eps_x')} / {f(a, 'eps_v')} | {f(b, 'eps_x')} / {f(b, 'eps_v')} | {f(c, 'eps_x')} / {f(c, 'eps_v')} |" + (" " + " / ".join(f(b, k) for k in extra) + " |" if full else ""))

"""profiles/<tag>_pmc_summary.txt -> profiles/pmc_traffic.json: HBM bytes per launch of each kernel, corrected as
MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE x2 for wide coalesced reads; KB -> bytes), for bench.py's
`roofline.traffic` (PMC passes cannot run inside the timed bench: they are separate rocprofv3 --pmc runs)."""
import ast
import json
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01f"
fetch, write = {}, {}
for line in open(f"profiles/{tag}_pmc_summary.txt"):
    m = re.match(r"(?:void )?spk::(\w+)(?:<[^>]*>)? (\{.*\}) n=", line)
    if not m:
        continue
    vals = ast.literal_eval(m.group(2))
    if "FETCH_SIZE" in vals:
        fetch[m.group(1)] = float(vals["FETCH_SIZE"])
    if "WRITE_SIZE" in vals:
        write[m.group(1)] = float(vals["WRITE_SIZE"])
out = {"source": f"profiles/{tag}_pmc_summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py, 10M particles)",
       "particles_per_gpu": 10000000, "correction": "bytes = 1024 * (2 * FETCH_SIZE + WRITE_SIZE)  [gfx950: FETCH_SIZE counts 64 B per 128 B request]",
       # k_p2g_pair reads 32 B per lane at a 32 B stride: the x2 calibration (16 B per lane streaming reads) does not cover it
       "bytes_per_launch": {k: 1024.0 * (2.0 * fetch[k] + write.get(k, 0.0)) for k in fetch if k != "k_p2g_pair"}}
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))

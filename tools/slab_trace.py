"""slab_trace.py <dir-with-rocprofv3-csv> [--tail-fraction 0.5]: the slab group loop's launches from a rocprofv3 --kernel-trace run of
tools/probe.py slabs: per kernel the mean duration and the mean GAP in front of it (start minus the previous dispatch's end on the same
queue), over the last part of the trace (the steady state behind the first re-bin), and what a slab-substep adds up to.
Used for profiles/r06_slab8_*.txt (where the fixed cost of a slab-substep goes)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("spk::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_]+)(<.*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name


def main():
    d = sys.argv[1]
    frac = float(sys.argv[sys.argv.index("--tail-fraction") + 1]) if "--tail-fraction" in sys.argv else 0.5
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit(f"no *kernel_trace.csv under {d}")
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "0")))
    rows.sort()
    t_lo = rows[0][0] + (rows[-1][1] - rows[0][0]) * (1.0 - frac)
    rows = [r for r in rows if r[0] >= t_lo]
    dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
    last_end = {}
    for st, en, name, q in rows:
        dur[name] += (en - st) * 1e-3
        if q in last_end:
            gap[name] += max(0, st - last_end[q]) * 1e-3
        last_end[q] = max(last_end.get(q, 0), en)
        cnt[name] += 1
    span_us = (rows[-1][1] - rows[0][0]) * 1e-3
    busy = sum(dur.values())
    print(f"{len(rows)} dispatches over {span_us / 1e3:.2f} ms (last {frac:.0%} of the trace); kernels busy {busy / span_us:.1%} of it")
    print(f"{'kernel':58s} {'launches':>8s} {'mean us':>9s} {'gap in front us':>16s} {'share of span':>14s}")
    for name in sorted(dur, key=lambda k: -dur[k] - gap[k]):
        print(f"{name[:58]:58s} {cnt[name]:8d} {dur[name] / cnt[name]:9.2f} {gap[name] / cnt[name]:16.2f} {(dur[name] + gap[name]) / span_us:14.1%}")
    fused = max((n for n in cnt if n.startswith("k_g2p2g")), key=lambda n: cnt[n], default=None)
    if fused:
        n = cnt[fused]
        print(f"per slab-substep (= per {fused.split('<')[0]} launch, {n} of them): {span_us / n:.2f} us of wall clock, of which {dur[fused] / n:.2f} us in that kernel, "
              f"{(busy - dur[fused]) / n:.2f} us in the other kernels, {(span_us - busy) / n:.2f} us idle between launches")


if __name__ == "__main__":
    main()

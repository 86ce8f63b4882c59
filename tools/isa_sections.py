"""Static instruction counts between the SPK_MARK section markers of one kernel (analysis build with -DSPK_MARKS):
python tools/isa_sections.py [kernel-substring]   (default: the fused k_g2p2g<-1, false, true>)
Counts are per wave pass over the straight-line code that follows each marker up to the next one (both sides of branches are
counted: cold paths inflate a section, loops count once)."""
import collections, os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
hot = "--hot" in sys.argv   # -DSPK_HOT_ONLY: cold sides of the branches assumed away (device_math.h, SPK_HOT): the count of the hot path
args = [a for a in sys.argv[1:] if a != "--hot"]
sub = args[0] if args else "7k_g2p2gILin1ELb0ELb1ELi1"
src = os.path.join(root, "sparkl_amd", "csrc", "sparkl_hip.hip")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-fno-slp-vectorize", "-DSPK_MARKS", *(["-DSPK_HOT_ONLY"] if hot else []), *os.environ.get("EXTRA", "").split(), "-S", "--cuda-device-only",
                       "-o", "/tmp/marks.s", src], stderr=subprocess.DEVNULL)
lines = open("/tmp/marks.s").read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN3spk\S*" + re.escape(sub) + r"\S*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
sec, counts, order = "prologue", collections.defaultdict(lambda: collections.Counter()), ["prologue"]
for l in lines[start:end]:
    m = re.search(r"; SPKMARK (\S+)", l)
    if m:
        sec = m.group(1)
        if sec not in order: order.append(sec)
        continue
    t = l.strip().split()
    if not t or t[0].startswith((";", ".")) or t[0].endswith(":"): continue
    op = t[0]
    cls = "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "scratch_", "flat_")) else "salu" if op.startswith("s_") else "other"
    counts[sec][cls] += 1
    if cls == "valu":   # issue cost in units of one v_fma_f32 (tools/microbench/instrate.hip, cndmask.hip on gfx950)
        w = 3.3 if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")) else 1.78 if op.startswith("v_pk_") else 1.7 if "_dpp" in op else \
            1.6 if op.startswith(("v_cvt_f64", "v_mul_f64", "v_fma_f64", "v_add_f64")) else 1.55 if op.startswith(("v_cmp", "v_cndmask", "v_max", "v_min", "v_med3")) else 1.0
        counts[sec]["cost"] += w
    if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")): counts[sec]["trans"] += 1
    if op.startswith("v_pk_"): counts[sec]["pk"] += 1
    if op.startswith("v_mov"): counts[sec]["mov"] += 1
tot = collections.Counter()
for s in order:
    c = counts[s]; tot.update(c)
    print(f"{s:16s} valu {c['valu']:5d} (pk {c['pk']:4d} mov {c['mov']:4d} trans {c['trans']:3d}) cost {c['cost']:7.0f}  lds {c['lds']:4d}  vmem {c['vmem']:3d}  salu {c['salu']:4d}")
print(f"{'total':16s} valu {tot['valu']:5d} (pk {tot['pk']:4d} mov {tot['mov']:4d} trans {tot['trans']:3d}) cost {tot['cost']:7.0f}  lds {tot['lds']:4d}  vmem {tot['vmem']:3d}  salu {tot['salu']:4d}")

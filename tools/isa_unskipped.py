"""Exec-masked blocks of a kernel that the compiler laid out WITHOUT an s_cbranch_execz skip (short divergent sides: they are issued even when no
lane takes them).  python tools/isa_unskipped.py [file.s]   (default /tmp/kern.s as written by tools/isa_hist.sh)"""
import re, sys
L = open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/kern.s").read().splitlines()
def real(l):
    t = l.strip().split()
    return t if t and not t[0].startswith((";", ".")) and not t[0].endswith(":") else None
for i, l in enumerate(L):
    t = real(l)
    if not t or t[0] not in ("s_and_saveexec_b64", "s_andn2_saveexec_b64", "s_or_saveexec_b64"): continue
    j, k, br = i + 1, 0, False
    while j < len(L) and k < 3:
        tt = real(L[j])
        if tt:
            k += 1
            if tt[0].startswith("s_cbranch_exec"): br = True; break
        j += 1
    if br: continue
    j, ops = i + 1, []
    while j < len(L):
        tt = real(L[j])
        if tt:
            if "exec" in L[j] and tt[0].startswith(("s_or_b64", "s_andn2_saveexec", "s_mov_b64", "s_and_saveexec", "s_or_saveexec")): break
            if tt[0] == "s_xor_b64" and "exec" in L[j] and ops: break
            if tt[0].startswith(("v_", "ds_", "global_", "scratch_")): ops.append(tt[0])
        j += 1
    if ops: print(f"{i + 1:6d} {t[0]:22s} {len(ops):3d}  " + " ".join(ops[:10]))

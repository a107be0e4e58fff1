#!/usr/bin/env python3
"""Per-function summary of a device assembly listing (hipcc --cuda-device-only -S):
VGPRs, private segment, scratch stores/loads (callee-saved prologue saves vs the rest), instruction counts.

    python scripts/asm_summary.py /tmp/chomp_kernel.s [filter]"""
import re
import subprocess
import sys

path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
name = None
funcs = {}
order = []
for line in open(path):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        name = m.group(1)
        funcs[name] = dict(n=0, valu=0, salu=0, lds=0, st=0, ld=0, pro=0, vmem=0, dpp=0, first_branch=False)
        order.append(name)
        continue
    if name is None:
        continue
    f = funcs[name]
    s = line.strip()
    m = re.match(r"\.set \.L(_Z\w+)\.(num_vgpr|private_seg_size|numbered_sgpr), (\d+)", s)
    if m and m.group(1) in funcs:
        funcs[m.group(1)][m.group(2)] = int(m.group(3))
        continue
    if not s or s.startswith((".", ";")) or s.endswith(":"):
        continue
    op = s.split()[0]
    f["n"] += 1
    if op.startswith("v_"):
        f["valu"] += 1
        if "dpp" in s or "row_" in s or "quad_perm" in s or "wave_" in s:
            f["dpp"] += 1
    elif op.startswith("s_"):
        f["salu"] += 1
        if op.startswith("s_cbranch") or op.startswith("s_branch"):
            f["first_branch"] = True
    elif op.startswith("ds_"):
        f["lds"] += 1
    elif op.startswith("scratch_store"):
        f["st"] += 1
        if not f["first_branch"]:
            f["pro"] += 1
    elif op.startswith("scratch_load"):
        f["ld"] += 1
    elif op.startswith(("global_", "flat_", "buffer_")):
        f["vmem"] += 1
names = {}
try:
    out = subprocess.run(["c++filt"], input="\n".join(order), capture_output=True, text=True).stdout.split("\n")
    names = dict(zip(order, out))
except Exception:
    pass
print("%-110s %5s %5s %6s %6s %5s %5s %5s %5s %4s %4s(pro)" % ("function", "vgpr", "priv", "insts", "valu", "dpp", "salu", "lds", "vmem", "st", "ld"))
for k in order:
    f = funcs[k]
    nm = names.get(k, k).replace("(anonymous namespace)::", "")
    if flt and flt not in nm:
        continue
    print("%-110s %5s %5s %6d %6d %5d %5d %5d %5d %4d %4d(%d)" % (nm[:110], f.get("num_vgpr", "?"), f.get("private_seg_size", "?"), f["n"], f["valu"],
                                                                f["dpp"], f["salu"], f["lds"], f["vmem"], f["st"], f["ld"], f["pro"]))

#!/usr/bin/env python3
"""Opcode histogram of one function of a device assembly listing, per basic block when asked:

    python scripts/asm_ops.py /tmp/chomp_kernel.s 'phase_cost<double, false, true, 256, 11, true, 4>' [out.s]"""
import collections
import re
import subprocess
import sys

path, want = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
names = [re.match(r"^(_Z\w+):", l).group(1) for l in lines if re.match(r"^_Z\w+:", l)]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
dem = [d.replace("(anonymous namespace)::", "") for d in dem]
hits = [n for n, d in zip(names, dem) if want in d]
if not hits:
    raise SystemExit("no function matches; e.g.\n" + "\n".join(d for d in dem if "phase_cost" in d)[:2000])
target = hits[0]
start = next(i for i, l in enumerate(lines) if l.startswith(target + ":"))
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write("\n".join(body))
ops = collections.Counter()
for l in body:
    s = l.strip()
    if not s or s.startswith((".", ";")) or s.endswith(":"):
        continue
    op = s.split()[0]
    if "dpp" in s or "row_" in s or "quad_perm" in s:
        op += " (dpp)"
    ops[op] += 1
tot = sum(ops.values())
valu = sum(v for k, v in ops.items() if k.startswith("v_"))
print("%s: %d instructions, %d vector" % (want, tot, valu))
for k, v in ops.most_common():
    print("  %-30s %d" % (k, v))

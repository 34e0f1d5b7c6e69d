"""Instruction mix of the loops of a kernel in the assembly dump written by tools/isa_stats.sh (/tmp/isa_kernel.s):
prints every loop (backward branch) that contains `n_mfma` MFMAs (default 12: the XDL attention block)."""
import re
import sys
from collections import Counter

n_mfma = int(sys.argv[1]) if len(sys.argv) > 1 else 12
K = open("/tmp/isa_kernel.s").read().split("\n")
lab = {}
for i, l in enumerate(K):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        lab[m.group(1)] = i
for i, l in enumerate(K):
    m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in lab and lab[m.group(1)] < i:
        a = lab[m.group(1)]
        seg = [t.strip() for t in K[a + 1:i + 1] if t.strip() and not t.strip().startswith(";") and not t.strip().startswith(".")]
        nm = sum(1 for t in seg if t.startswith("v_mfma"))
        if nm == n_mfma and len(seg) < 600:
            c = Counter(t.split()[0].replace("_e32", "").replace("_e64", "") for t in seg)
            print(f"lines {a}-{i}: {len(seg)} instructions;", ", ".join(f"{k} {v}" for k, v in c.most_common(14)))

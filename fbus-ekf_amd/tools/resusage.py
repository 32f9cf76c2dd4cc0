#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage remarks (read from stdin)."""
import re, sys, subprocess
rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"remark: (?:[^:]+:\d+:\d+: )?\s*(.*?) \[-Rpass", line)
    if not m:
        if "error" in line: print(line.rstrip())
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        try: name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
        except Exception: pass
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = re.sub(r"^void ", "", name).split("(")[0]
        cur = {"name": name}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
for r in rows:
    print(f"{r['name']:<48} V={r.get('VGPRs','?'):>3} A={r.get('AGPRs','?'):>3} S={r.get('TotalSGPRs','?'):>3} "
          f"scratch={r.get('ScratchSize [bytes/lane]','?'):>5} spill={r.get('VGPRs Spill','?'):>4} occ={r.get('Occupancy [waves/SIMD]','?')}")

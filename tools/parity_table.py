#!/usr/bin/env python3
"""Every `[parity]` / `[emulation]` / `[team vs one-wave]` line the GPU suite printed (`pytest tests -m gpu -s`), grouped by test, as one table:
    python tools/parity_table.py gpurun_out/r05/pytest_gpu.log > profiles/r05_parity_table.txt
The figures are device-vs-oracle (fp32 kernels against the fp64 C oracle on identical inputs; fp64 kernels where the line says so) or
kernel-form-vs-kernel-form; the gates they are held to are in tests/util.py (assert_parity: single steps re-seeded from the oracle;
assert_window_parity: free-running windows)."""
import re
import sys

log = open(sys.argv[1], errors="replace").read().splitlines()
out, cur = [], None
for line in log:
    m = re.search(r"(tests/test_[a-z0-9_]+\.py)", line)
    if m and ("::" in line or line.strip().startswith("tests/")):
        cur = m.group(1)
    for tag in ("[parity]", "[emulation]", "[team vs one-wave]"):
        k = line.find(tag)
        if k >= 0:
            out.append(line[k:].rstrip())
tail = [l for l in log if re.search(r"\d+ passed", l)]
print(f"# {len(out)} parity lines of `pytest tests -m gpu -s` ({sys.argv[1]}); suite result: {tail[-1].strip() if tail else 'n/a'}")
print("# gates: tests/util.py -- single step: literal <= 1e-5, sigma-aware <= 1e-5, plain <= 2e-4, cov <= 1e-4, cov block-wise <= 1e-5; fp64 kernels 1e-9;")
print("#        free-running windows: literal <= 1e-5 (C++ dialect, N = 18: 5e-5, see util.py), sigma-aware <= 1e-4, plain <= 5e-3, cov <= 1e-4, cov block-wise <= 1e-4")
for l in out:
    print(l)

#!/usr/bin/env python3
"""Instruction mix of the kernels inside one object (llvm-objdump of the embedded gfx950 code object):
VALU by class (fma / pk_fma / accvgpr moves / mov / other), VMEM, DS, SALU, s_waitcnt, scratch.
  python tools/isa_stats.py fbus-ekf_amd/lib/obj/f32_18_frame.o [name-filter]"""
import collections, os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
path, filt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
with tempfile.TemporaryDirectory() as td:
    r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--list", "--type=o", f"--input={path}"], capture_output=True, text=True)
    if not [t for t in r.stdout.split() if "gfx950" in t]:      # host object / .so: the fat binary sits in .hip_fatbin
        fb = os.path.join(td, "fb")
        subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fb], check=True)
        path = fb
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--list", "--type=o", f"--input={path}"], capture_output=True, text=True)
    tgt = [t for t in r.stdout.split() if "gfx950" in t][0]
    co = os.path.join(td, "co")
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={path}", f"--targets={tgt}", f"--output={co}"], check=True)
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(dis)
cur, stats = None, {}
for line in dis.splitlines():
    m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
    if m:
        cur = m.group(1); stats[cur] = collections.Counter(); continue
    if cur is None: continue
    t = line.split()
    if len(t) < 1 or not re.match(r"^[a-z_0-9]+$", t[0]): continue
    op = t[0]
    c = stats[cur]
    c["total"] += 1
    if op.startswith("v_accvgpr"): c["v_accvgpr"] += 1
    elif op.startswith("v_pk_fma"): c["v_pk_fma"] += 1
    elif op.startswith("v_pk_"): c["v_pk_other"] += 1
    elif op.startswith("v_fma") or op.startswith("v_fmac"): c["v_fma"] += 1
    elif op.startswith("v_mov") : c["v_mov"] += 1
    elif op.startswith("v_"): c["v_other"] += 1
    elif op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_"): c["vmem"] += 1
    elif op.startswith("scratch_"): c["scratch"] += 1
    elif op.startswith("ds_"): c["ds"] += 1
    elif op.startswith("s_waitcnt"): c["s_waitcnt"] += 1
    elif op.startswith("s_nop"): c["s_nop"] += 1
    elif op.startswith("s_"): c["salu"] += 1
for k, c in stats.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    if "kernel" not in name or filt not in name: continue
    print(name)
    print("   " + "  ".join(f"{a}={c[a]}" for a in ("total", "v_fma", "v_pk_fma", "v_pk_other", "v_mov", "v_accvgpr", "v_other", "vmem", "scratch", "ds", "salu", "s_waitcnt", "s_nop")))

#!/usr/bin/env python3
"""Instruction mix of the LOOPS of a kernel (what a resident kernel issues per ImuUpdate, not what it holds in total):
every backward branch of the disassembly closes a loop; per loop the VALU instructions by class and -- the number that prices a
VALU-issue-bound kernel -- the issue slots, with the slow paths of libm calls (blocks skipped by a forward s_cbranch_execz whose
body holds v_mad_u64_u32 / v_alignbit: the Payne-Hanek reduction of sincosf) listed apart.
  python tools/loop_isa.py fbus-ekf_amd/lib/obj/f32_18_frames.o 'frames_kernel<float, 18, 0, 0, true>'
"""
import collections, os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
path, want = sys.argv[1], sys.argv[2]
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


def klass(op):
    if op.startswith("v_accvgpr"): return "accvgpr"
    if op.startswith("v_pk_"): return "pk"
    if op.startswith("v_mov"): return "mov"
    if op.startswith("v_fma") or op.startswith("v_fmac") or op.startswith("v_mul_f") or op.startswith("v_add_f") or op.startswith("v_sub_f"): return "fp"
    if op.startswith("v_"): return "v_other"
    if op.startswith("ds_"): return "ds"
    if op.startswith("buffer_") or op.startswith("global_"): return "vmem"
    if op.startswith("s_"): return "s"
    return "other"


cur, ins = None, {}
for line in dis.splitlines():
    m = re.match(r"^([0-9a-f]+) <(.+)>:$", line)
    if m:
        name = subprocess.run(["c++filt", m.group(2)], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")
        cur = name.split("(")[0].replace("void ", "")
        ins[cur] = []
        continue
    if cur is None: continue
    t = line.split()
    if not t or not re.match(r"^[a-z_0-9]+$", t[0]): continue
    m = re.search(r"// ([0-9A-F]+):", line)
    if not m: continue
    addr = int(m.group(1), 16)
    tgt_addr = None
    if t[0].startswith("s_cbranch") or t[0] == "s_branch":
        mm = re.search(r"\+0x([0-9a-f]+)>", line)
        if mm: tgt_addr = ("off", int(mm.group(1), 16))
    ins[cur].append((addr, t[0], tgt_addr))
for name, L in ins.items():
    if want not in name: continue
    base = L[0][0]
    addr_idx = {a - base: i for i, (a, _, _) in enumerate(L)}
    print(name, f"({len(L)} instructions)")
    loops = []
    for i, (a, op, tg) in enumerate(L):
        if tg and tg[1] in addr_idx and addr_idx[tg[1]] <= i:
            loops.append((addr_idx[tg[1]], i))
    for (lo, hi) in sorted(loops, key=lambda x: x[0] - x[1])[:6]:
        body = L[lo:hi + 1]
        # forward-skipped slow blocks: s_cbranch_execz ... whose span holds v_mad_u64_u32 (Payne-Hanek)
        slow = set()
        for i in range(lo, hi + 1):
            a, op, tg = L[i]
            if op.startswith("s_cbranch_exec") and tg and tg[1] in addr_idx and i < addr_idx[tg[1]] <= hi:
                span = range(i + 1, addr_idx[tg[1]])
                if any(L[j][1].startswith("v_mad_u64_u32") or L[j][1].startswith("v_alignbit") for j in span):
                    slow.update(span)
        c, cs = collections.Counter(), collections.Counter()
        for i in range(lo, hi + 1):
            (cs if i in slow else c)[klass(L[i][1])] += 1
        valu = sum(c[k] for k in ("fp", "pk", "mov", "accvgpr", "v_other"))
        print(f"  loop [{lo}, {hi}] {hi - lo + 1} instructions: VALU slots on the fast path {valu}"
              f"  (fp {c['fp']}  packed {c['pk']}  mov {c['mov']}  accvgpr {c['accvgpr']}  other {c['v_other']})"
              f"  ds {c['ds']} vmem {c['vmem']} salu {c['s']};  slow blocks {sum(cs.values())}")

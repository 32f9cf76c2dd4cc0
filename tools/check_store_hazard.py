#!/usr/bin/env python3
"""Static check of the gfx950 store-data hazard in every built object: a buffer_store_dwordx3/x4 (or b64 x2 etc.,
anything wider than 64 bits) whose data registers are written by the instruction in the very next issue slot.
Measured on MI355X (round 2): that write reaches memory on lanes 12-15 of every 16.  LLVM pads the case only when the
store's soffset is not a register; csrc/ekf_kernels.hpp::store_chunks is written accordingly.  Exit code 1 on a hit."""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"

def disasm(path):
    with tempfile.TemporaryDirectory() as td:
        r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--list", "--type=o", f"--input={path}"], capture_output=True, text=True)
        if not [t for t in r.stdout.split() if "gfx950" in t]:
            fb = os.path.join(td, "fb")
            subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fb], check=True)
            path = fb
            r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--list", "--type=o", f"--input={path}"], capture_output=True, text=True)
        out = ""
        for i, t in enumerate(t for t in r.stdout.split() if "gfx950" in t):
            co = os.path.join(td, f"co{i}")
            subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={path}", f"--targets={t}", f"--output={co}"], check=True)
            out += subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
        return out

def regs(tok):
    m = re.match(r"^([va])\[(\d+):(\d+)\]", tok)
    if m: return {(m.group(1), k) for k in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r"^([va])(\d+)\b", tok)
    return {(m.group(1), int(m.group(2)))} if m else set()

def check(path):
    hits, stores, kernel = [], 0, None
    lines = [l for l in disasm(path).splitlines()]
    ins = []
    for l in lines:
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", l)
        if m: kernel = m.group(1); continue
        t = l.split("//")[0].replace(",", " ").split()
        if t and re.match(r"^[a-z_0-9]+$", t[0]): ins.append((kernel, t))
    for i, (k, t) in enumerate(ins[:-1]):
        if re.match(r"^(buffer|global|flat|scratch)_store_dwordx[34]$", t[0]):
            stores += 1
            data = regs(t[1] if t[0].startswith("buffer_") else t[2])      # global/flat/scratch: vaddr first, then vdata
            k2, n = ins[i + 1]
            if k2 != k: continue
            if n[0].startswith("v_") and not n[0].startswith("v_cmp") and regs(n[1]) & data:
                hits.append((k, " ".join(t[:3]), " ".join(n[:3])))
            if n[0].startswith(("buffer_load", "global_load", "ds_read")) and regs(n[1]) & data:
                pass    # a load's write-back comes hundreds of cycles later
    return stores, hits

if __name__ == "__main__":
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "fbus-ekf_amd", "lib", "obj", "*.o")))
    bad = 0
    for f in files:
        n, hits = check(f)
        print(f"{os.path.basename(f):<24} {n:5d} wide stores, {len(hits)} with a data register overwritten in the next slot")
        for h in hits[:5]:
            print("    ", subprocess.run(["c++filt", h[0]], capture_output=True, text=True).stdout.strip()[:80], "|", h[1], "->", h[2])
        bad += len(hits)
    sys.exit(1 if bad else 0)

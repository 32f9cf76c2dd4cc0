#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS usage of the built code objects (llvm-readelf --notes on the gfx950 code
objects embedded in fbus-ekf_amd/lib/obj/*.o or in the .so): vgpr (arch VGPRs), agpr, total allocated, sgpr, scratch
bytes per lane, LDS bytes, and the waves per SIMD the register file then allows.
  python tools/kernel_resources.py [file ...]        default: every object under fbus-ekf_amd/lib/obj"""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"

def code_objects(path):
    """extract the gfx950 code object(s) of a host object / shared library"""
    out = []
    with tempfile.TemporaryDirectory() as td:
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--list", "--type=o", f"--input={path}"], capture_output=True, text=True)
        tgts = [t for t in r.stdout.split() if "gfx950" in t]
        if not tgts:                      # a .so / .o with the fat binary in a section: dump .hip_fatbin first
            fb = os.path.join(td, "fb")
            subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", path, fb], check=True)
            r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--list", "--type=o", f"--input={fb}"], capture_output=True, text=True)
            tgts = [t for t in r.stdout.split() if "gfx950" in t]
            path = fb
        for i, t in enumerate(tgts):
            co = os.path.join(td, f"co{i}")
            subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={path}", f"--targets={t}", f"--output={co}"], check=True)
            out.append(subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout)
    return out

def kernels(notes):
    for blk in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
        blk = ".agpr_count:" + blk
        g = lambda k: (re.search(rf"\.{k}:\s*(\S+)", blk) or [None, "?"])[1]
        name = g("name")
        yield name, int(g("vgpr_count")), int(g("agpr_count")), int(g("sgpr_count")), int(g("private_segment_fixed_size")), int(g("group_segment_fixed_size"))

def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "fbus-ekf_amd", "lib", "obj", "*.o")))
    print(f"{'kernel':<110} {'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'scratch':>8} {'lds':>6} {'waves/SIMD':>10}")
    for f in files:
        for notes in code_objects(f):
            for name, v, a, s, p, l in kernels(notes):
                dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                dem = re.sub(r"\(anonymous namespace\)::", "", dem).split("(")[0].replace("void ", "")
                alloc = (v + 7) // 8 * 8            # .vgpr_count already includes the AGPRs (unified file)
                print(f"{dem[:110]:<110} {v - a:>5} {a:>5} {s:>5} {p:>8} {l:>6} {min(8, 512 // max(alloc, 1)):>10}")

if __name__ == "__main__":
    main()

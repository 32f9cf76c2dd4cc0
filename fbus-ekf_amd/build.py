#!/usr/bin/env python3
"""Builds fbus-ekf_amd/lib/libfbus_ekf.so (HIP, gfx950) in-tree with hipcc.

The library is 25 translation units compiled in parallel and linked into one shared object:
  fbus_ekf.hip                          handle, C ABI, the small kernels (pack/unpack, init, EMA, marker pose)
  kernels_tu.hip x 24                   one kernel family for one (float|double, N = 18|15), both dialects (-DFBUS_TU_T/N/FAMILY):
                                        float: predict / correct / frame / frames / team / meas / fmeas / msplit (8 x 2 = 16),
                                        double: predict / correct / frame / meas (4 x 2 = 8; len(units()) == 25 with fbus_ekf.hip)
Objects live in fbus-ekf_amd/lib/obj/ (git-ignored) and are rebuilt when a source they include is newer.
  python build.py [--force] [--only f32_18_correct,...] [--jobs N]
FBUS_OUT / FBUS_EXTRA_FLAGS: experiment builds (A/B of differently built kernels via FBUS_EKF_LIB).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HEADERS = [os.path.join(CSRC, h) for h in ("ekf_kernels.hpp", "ekf_device.hpp", "vision_device.hpp", "ekf_launch.hpp", "ekf_team.hpp", "ekf_meas.hpp", "ekf_meas_split.hpp")] + \
          [os.path.join(HERE, "..", "include", "fbus_ekf.h")]
OUT = os.environ.get("FBUS_OUT") or os.path.join(HERE, "lib", "libfbus_ekf.so")   # FBUS_OUT / FBUS_EXTRA_FLAGS: experiment builds
OBJDIR = os.environ.get("FBUS_OBJDIR") or os.path.join(os.path.dirname(OUT), "obj" if not os.environ.get("FBUS_OUT") else
                                                       "obj_" + os.path.splitext(os.path.basename(OUT))[0])
FAMILIES = {"predict": 1, "correct": 2, "frame": 3, "frames": 5, "team": 6, "meas": 7, "fmeas": 8, "msplit": 9}
# Per-family scheduler choice (measured in one run, B = 65 536, tools/ab_bench.sh, profiles/logs/r02_ab2.log): the
# max-ILP strategy of the AMDGPU machine scheduler shortens the per-call kernels, where one wave per SIMD has nothing
# but its own independent instructions to cover dependent-issue stalls (predict 13.4 -> 13.05 us, stacked correct
# 23.0 -> 21.6 us, headline +3.5 %), and lengthens the fused frame kernel (-2.8 %: more live registers, more
# v_accvgpr traffic), which therefore keeps the default strategy.
FAMILY_FLAGS = {"predict": os.environ.get("FBUS_PREDICT_FLAGS", "-mllvm -amdgpu-sched-strategy=max-ilp").split(),
                "correct": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
                "frame": os.environ.get("FBUS_FRAME_FLAGS", "").split(), "frames": os.environ.get("FBUS_FRAMES_FLAGS", "").split(),
                "team": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"], "meas": os.environ.get("FBUS_MEAS_FLAGS", "").split(),
                "fmeas": os.environ.get("FBUS_FMEAS_FLAGS", "").split(), "msplit": os.environ.get("FBUS_MSPLIT_FLAGS", "").split()}
# fp64 units.  meas (correct_pixels2 / correct_corners2 <double>, 512 registers + scratch): the max-memory-clause strategy leaves them
# 28-136 bytes of scratch instead of 136-340 and is 4-11 % faster (profiles/r05_f64_sched.txt); FBUS_F64_FLAGS_<FAMILY> overrides
F64_FAMILY_FLAGS = {fam: os.environ.get("FBUS_F64_FLAGS_" + fam.upper(),
                                        "-mllvm -amdgpu-sched-strategy=max-memory-clause" if fam == "meas" else "").split()
                    for fam in ("predict", "correct", "frame", "meas")}
TYPES = {"f32": "float", "f64": "double"}


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def units():
    """(name, source, defines)"""
    out = [("main", os.path.join(CSRC, "fbus_ekf.hip"), [])]
    for tn, t in TYPES.items():
        for n in (18, 15):
            for fam, code in FAMILIES.items():
                if fam in ("frames", "team", "fmeas", "msplit") and tn == "f64":
                    continue                    # fp64: one fused frame kernel (family "frame": frame2_kernel), windows frame by frame
                out.append((f"{tn}_{n}_{fam}", os.path.join(CSRC, "kernels_tu.hip"),
                            [f"-DFBUS_TU_T={t}", f"-DFBUS_TU_N={n}", f"-DFBUS_TU_FAMILY={code}"] +
                            # fp32 only: the fp64 kernels sit at the 512-register limit and spill more under max-ILP
                            # FBUS_NO_FAMILY_FLAGS: the flag-free A/B baseline -- for BOTH record types (advisor, round 5)
                            ([] if (os.environ.get("FBUS_NO_FAMILY_FLAGS") or tn != "f32") else FAMILY_FLAGS[fam]) +
                            (F64_FAMILY_FLAGS.get(fam, []) if (tn == "f64" and not os.environ.get("FBUS_NO_FAMILY_FLAGS")) else [])))
    return out


def _base_flags():
    # -fno-slp-vectorize: the SLP pass packs the unrolled scalar FMAs into v_pk_fma_f32 and pays for it with
    # ~1.9x more instructions (v_mov / v_accvgpr shuffles to form register pairs) -- measured on the .s
    return ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fPIC"] + os.environ.get("FBUS_EXTRA_FLAGS", "").split()


def _flags_of(defs):
    """the flag string an object was / would be compiled with: written next to the object (<obj>.flags) so that a flag change made
    through the environment (FBUS_EXTRA_FLAGS, FBUS_*_FLAGS, FBUS_NO_FAMILY_FLAGS) invalidates it -- mtimes alone do not"""
    return " ".join(_base_flags() + list(defs))


def _stale(obj, src, defs=None):
    if not os.path.exists(obj):
        return True
    if defs is not None:
        try:
            if open(obj + ".flags").read() != _flags_of(defs):
                return True
        except OSError:
            return True
    t = os.path.getmtime(obj)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in [src, os.path.abspath(__file__)] + HEADERS)


def needs_build():
    """the shared object is older than a source (the objects are a build cache: they do not travel with the library)"""
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    srcs = {src for _, src, _ in units()} | set(HEADERS) | {os.path.abspath(__file__)}
    if any(os.path.exists(d) and os.path.getmtime(d) > t for d in srcs):
        return True
    # a partial rebuild (--only) links a library that is newer than every source while other objects are still stale
    return os.path.isdir(OBJDIR) and any(_stale(os.path.join(OBJDIR, name + ".o"), src, defs) for name, src, defs in units()
                                         if os.path.exists(os.path.join(OBJDIR, name + ".o")))


def build(force=False, verbose=False, only=None, jobs=None):
    if not force and not needs_build():
        return OUT
    os.makedirs(OBJDIR, exist_ok=True)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    flags = _base_flags()
    todo = []
    for name, src, defs in units():
        obj = os.path.join(OBJDIR, name + ".o")
        if only and not any(o in name for o in only) and os.path.exists(obj):
            continue
        if force or _stale(obj, src, defs) or (only and any(o in name for o in only)):
            todo.append((name, [hipcc()] + flags + defs + ["-c", src, "-o", obj], obj, _flags_of(defs)))
    # the fp64 / fused-frame units take longest: start them first
    todo.sort(key=lambda u: (("f64" in u[0]) * 2 + ("frame" in u[0]) + ("correct" in u[0])), reverse=True)  # "frame" matches "frames" too
    jobs = jobs or int(os.environ.get("FBUS_JOBS", "0")) or min(8, os.cpu_count() or 1)

    def run(u):
        if verbose:
            print("  hipcc", u[0], flush=True)
        r = subprocess.run(u[1], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"{u[0]}: {' '.join(u[1])}\n{r.stderr[-4000:]}")
        if r.stderr.strip() and verbose:
            print(r.stderr[-2000:])
        with open(u[2] + ".flags", "w") as f:
            f.write(u[3])
        return u[0]

    with ThreadPoolExecutor(max_workers=jobs) as ex:
        list(ex.map(run, todo))
    objs = [os.path.join(OBJDIR, name + ".o") for name, _, _ in units()]
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print("  link", os.path.relpath(OUT, HERE), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


def build_host_asan(out=None):
    """The library with the HOST half of fbus_ekf.hip (handle, argument checks, staging, parameter tables, RCCL binding) under
    AddressSanitizer + UBSan: `-fsanitize=address,undefined -fno-gpu-sanitize` instruments host code only, the device code
    objects are the normal ones (GPU ASan is not available on this pool).  Linked against the already built kernel objects.
    Load it with LD_PRELOAD of clang's ASan runtime (tests/test_sanitizers_cpu.py)."""
    build()
    out = out or os.path.join(os.path.dirname(OUT), "libfbus_ekf_asan.so")
    obj = os.path.join(OBJDIR, "main_asan.o")
    san = ["-fsanitize=address,undefined", "-fno-gpu-sanitize", "-shared-libsan"]
    subprocess.run([hipcc(), "--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-fno-slp-vectorize", "-fPIC"] + san +
                   ["-c", os.path.join(CSRC, "fbus_ekf.hip"), "-o", obj], check=True, capture_output=True)
    objs = [os.path.join(OBJDIR, name + ".o") for name, _, _ in units() if name != "main"]
    subprocess.run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + san + ["-o", out, obj] + objs, check=True, capture_output=True)
    return out


def asan_runtime():
    """clang's shared ASan runtime (what LD_PRELOAD needs for build_host_asan's library)"""
    import glob
    c = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    return c[-1] if c else None


if __name__ == "__main__":
    only = None
    jobs = None
    for i, a in enumerate(sys.argv):
        if a == "--only":
            only = sys.argv[i + 1].split(",")
        if a == "--jobs":
            jobs = int(sys.argv[i + 1])
    if "--host-asan" in sys.argv:
        print(build_host_asan())
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose=True, only=only, jobs=jobs))

#!/usr/bin/env python3
"""Builds fbus-ekf_amd/lib/libfbus_ekf.so (HIP, gfx950) in-tree with hipcc."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(HERE, "csrc", "fbus_ekf.hip")]
DEPS = SRC + [os.path.join(HERE, "csrc", "ekf_kernels.hpp"), os.path.join(HERE, "csrc", "ekf_device.hpp"), os.path.join(HERE, "csrc", "vision_device.hpp"),
              os.path.join(HERE, "..", "include", "fbus_ekf.h")]
OUT = os.environ.get("FBUS_OUT") or os.path.join(HERE, "lib", "libfbus_ekf.so")   # FBUS_OUT / FBUS_EXTRA_FLAGS: experiment builds


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    srcs = [s for s in SRC if os.path.exists(s)]
    # -fno-slp-vectorize: the SLP pass packs the unrolled scalar FMAs into v_pk_fma_f32 and pays for it with
    # ~1.9x more instructions (v_mov / v_accvgpr shuffles to form register pairs) -- measured on the .s
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-shared", "-fPIC", "-o", OUT] + os.environ.get("FBUS_EXTRA_FLAGS", "").split() + srcs
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

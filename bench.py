#!/usr/bin/env python3
"""bench.py -- EKF steps/s of the batched predict/correct hot path on N MI355X.

Workload (BASELINE.json / BASELINE.md section 2.1): 200 Hz IMU + 30 Hz stereo, 4 markers
per frame, fp32, N = 18 parity layout, Matlab dialect.  One bench "step" = 1 s of
simulated time for the whole batch = 30 camera frames = ten times the 7/7/6 pattern =
200 predict launches + 30 correct launches = 230 EKF steps per filter, every launch
through the per-call C ABI (the state makes a full HBM round trip per EKF step).  (Rounds
1 and 2a used 0.1 s per step; with `--steps 20` the timed region was then 6 ms, short
enough for the bracketing synchronisations to cost 3 % and for the driver's GPU-busy
sampler to miss it.  The metric is per EKF step and does not depend on the choice.)
Inputs are generated on the host with the seeded synthetic generator and are resident
in HBM before the timed region starts.

  python bench.py [--gpus N] [--steps K] [--warmup W]            weak scaling: 65 536 filters per GPU
  python bench.py --gpus N --total-batch 262144                  strong scaling: BASELINE config 4, the same
                                                                 262 144 filters over 1 / 2 / 4 / 8 GPUs
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...      (what the driver runs)

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N rank processes itself
(`python -m torch.distributed.run ...` as a CHILD process, before this process has touched the GPU),
relays their output and exits with their return code; it fails loudly when the box has fewer than N GPUs.
Under a launcher, WORLD_SIZE must equal --gpus.

Prints ONE COMPACT JSON line (< 6 KB: `compact_line`) on rank 0 -- the contract's keys, `roofline`, `cpu_baseline`, one short block
each for `fp64` and `north_star` -- and writes the FULL result (every leg described below) to `bench_detail.json` beside this file
(`--detail-file`; `--detail-stderr` also prints it as one `bench_detail: {...}` line on stderr).  (Round 5 printed the full 20.7 KB object on stdout and the driver
could not parse it.)  In the full result `roofline` is for the dominant kernel (predict): bytes the kernel MOVES per
launch (1436 B per filter: 828 read, 608 written -- the predict-invariant covariance tail and ba/bg/g are not
written back) / its average duration measured with HIP events on the launch stream inside the timed region;
`achieved_api` prices SURVEY.md 8(d)'s full record round trip (1620 B) instead.  `roofline_hbm_resident` is the
same measurement at 1 048 576 filters per GPU (839 MB of records: three times the 256 MiB Infinity Cache);
`roofline_b262144` the round-2 leg (210 MB: partly cache-resident).  `fp64` is the same workload through the fp64 kernels
(per-call, one launch per frame as `fp64.fused_frame`, and past the cache at 524 288 filters as `fp64.roofline_hbm_resident`),
`north_star_rows` the same mixed schedule with the north star's own measurement updates (correct_pixels / correct_corners) in
place of the pose update.  `legs_skipped` names the legs a line does not carry.  `cpu_baseline` is the fp64 dense oracle port
(oracle/), timed on a bounded sample.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "fbus-ekf_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

PATTERN = (7, 7, 6)                 # predicts between corrects: 200 Hz IMU / 30 Hz stereo
STEPS_PER_PATTERN = sum(PATTERN) + len(PATTERN)         # 23 EKF steps per filter in 0.1 s
PATTERNS_PER_STEP = 10              # one bench step = 1 s of sensor time = 30 camera frames
STEPS_PER_BENCH_STEP = STEPS_PER_PATTERN * PATTERNS_PER_STEP      # 230 EKF steps per filter
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PIXELS_SQ = "r06_pixels_sq.json"
PROFILE_ROUND = 6                   # only profiles/r06_* digests are quoted as `traffic` (collected on this round's kernels)
# packed record round trip + inputs (what the per-call API implies)
PREDICT_BYTES_API = 2 * 796 + 28
CORRECT_BYTES_API = lambda M: 2 * 796 + 32 * M
# what the kernels move (fp32, N = 18): records are 50 chunks of 16 B
#   predict: reads 50 chunks + 28 B IMU sample, writes 5 (p q R v) + 33 (covariance elements ImuUpdate can change)
#   correct: reads 50 chunks + 32 B per marker slot, writes 2 (p q) + 3 (v ba bg g) + 43 (covariance + prev id) + 1 B flag
PREDICT_BYTES_MOVED = 50 * 16 + 28 + 38 * 16
CORRECT_BYTES_MOVED = lambda M: 50 * 16 + 32 * M + 48 * 16 + 1
POOL = 4                            # distinct 0.1 s input patterns (IMU samples + marker frames) resident in HBM, cycled
# The HBM-roofline claim needs records that do NOT fit the 256 MiB (268 MB) Infinity Cache: 1 048 576 filters = 839 MB of records.
# (Round 2 used 262 144 filters = 210 MB, which still fits: that leg is kept as `roofline_b262144`, labelled for what it is.)
HBM_LEG_BATCH = 1048576
F64_HBM_LEG_BATCH = 524288
MID_LEG_BATCH = 262144
HBM_COPY_CEILING_GBS = 6290.0       # MI355X_MICROARCH.md: what a float4 copy kernel reaches on this part (0.79 of the 8 TB/s spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=65536, help="filters per GPU (weak scaling)")
    ap.add_argument("--total-batch", type=int, default=0,
                    help="strong scaling: this many filters in total, cut into contiguous 64-aligned shards over the "
                         "ranks (BASELINE.json config 4: 262144 over 1/2/4/8 GPUs)")
    ap.add_argument("--tile", type=int, default=1,
                    help="generate the inputs for batch/tile filters and repeat them on the device (the 1 048 576-filter "
                         "leg as the main workload: --batch 1048576 --tile 16)")
    ap.add_argument("--only-pixels", action="store_true", help="run only the compute-bound correct_pixels leg (profiling)")
    ap.add_argument("--markers", type=int, default=4)
    ap.add_argument("--dtype", type=int, choices=[32, 64], default=32,
                    help="record / arithmetic type of the MAIN workload (64: the reference's own arithmetic as the main line -- what "
                         "tools/profile_gpu.sh profiles for the fp64 digests; the headline and the driver's run are 32)")
    ap.add_argument("--mode", choices=["stacked", "nearest"], default="stacked")
    ap.add_argument("--dialect", choices=["matlab", "cpp"], default="matlab")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--detail-stderr", action="store_true", help="also print the full result as one `bench_detail: {...}` line on stderr")
    ap.add_argument("--detail-file", default=None,
                    help="where the FULL result goes (default: bench_detail.json beside bench.py and under gpurun_out/ when that exists); "
                         "stdout carries the compact line only")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-hbm-leg", action="store_true", help="skip the 1 048 576-filter (HBM-resident) and 262 144-filter roofline legs")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the fp64 leg and the correct_pixels (compute-bound) leg")
    ap.add_argument("--graphs", action="store_true",
                    help="replay each 0.1 s pattern (23 launches) from a captured HIP graph (launch-bound small batches); "
                         "the per-kernel HIP-event timing then comes from a short eager pass after the timed region")
    ap.add_argument("--kernel-timing", choices=["on", "off"], default="on",
                    help="bracket runs of launches with HIP events inside the timed region (feeds `roofline`)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------
# multi-GPU front door
# ------------------------------------------------------------------------------------------------
def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks as a child `torch.distributed.run` and relay.
    Nothing in this process has touched the GPU (torch.cuda.device_count() does not initialise it), and the ranks
    are CHILD processes -- this process is never replaced by another program."""
    import torch
    shared = os.environ.get("FBUS_BENCH_DEBUG_SHARED_GPU") == "1"
    have = torch.cuda.device_count()
    if have < args.gpus and not shared:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} requested but this box has {have} GPU(s); refusing to report a "
                         f"{args.gpus}-GPU number from fewer devices (there is no CPU fallback)\n")
        return 3
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


# ------------------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------------------
def pmc_traffic(batch, args, world, kernel="predict"):
    """HBM-side bytes per launch of a kernel from THIS round's committed rocprofv3 PMC passes (profiles/r06_digest_b<batch>.json,
    written by tools/profile_digest.py: separate FETCH_SIZE / WRITE_SIZE passes, KiB units, FETCH_SIZE doubled on gfx950 as
    MI355X_MICROARCH.md prescribes).  bench.py cannot collect PMC counters itself.  The digest is only quoted when it was taken
    on THIS configuration (batch, dialect, markers, mode, eager launches, one GPU) and in THIS round (PROFILE_ROUND: a digest of an
    earlier round's kernels is never quoted -- `traffic` is null then and `legs_skipped` says so)."""
    if world != 1 or args.graphs:
        return None, None
    path = os.path.join(ROOT, "profiles", f"r{PROFILE_ROUND:02d}_digest_b{batch}.json")
    try:
        d = json.load(open(path))
        cfg = d.get("_config", {})
        if (cfg.get("batch"), cfg.get("dialect"), cfg.get("markers"), cfg.get("mode")) != \
                (batch, args.dialect, args.markers, args.mode):
            return None, None
        for name, v in d.items():
            if name.startswith(kernel + "_kernel<float, 18") and "fetch_bytes" in v:
                return v["fetch_bytes"] + v["write_bytes"], os.path.relpath(path, ROOT)
    except Exception:
        pass
    return None, None


def pmc_traffic_named(name, batch, args, kernel_prefix):
    """as pmc_traffic, from a named digest under profiles/ (the fp64 legs)"""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
        cfg = d.get("_config", {})
        if (cfg.get("batch"), cfg.get("dialect"), cfg.get("markers"), cfg.get("mode")) != (batch, args.dialect, args.markers, args.mode):
            return None, None
        for k, v in d.items():
            if k.startswith(kernel_prefix) and "fetch_bytes" in v:
                return v["fetch_bytes"] + v["write_bytes"], os.path.join("profiles", name)
    except Exception:
        pass
    return None, None


def usable_cores():
    """host cores this process may actually use: cgroup CPU quota, else the affinity mask, else cpu_count"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, int(quota / period + 0.5)))
            break
        except Exception:
            continue
    return n


def cpu_baseline(args, seconds):
    """The fp64 dense oracle port on the host cores, on a bounded sample, two legs:

    value          the reference's own CPU path: C++ dialect, nearest marker with hysteresis, ONE 7-row update per frame
                   solved by LDLT -- FILTER::UpdateCovariance + UpdateNominalState (filter.cpp:588-616,533-582) and
                   FILTER::ObservationUpdate (filter.cpp:622-741) restated operation for operation with dense 18 x 18 loops;
    same_workload  what the GPU number runs: the bench's dialect and correct mode (Matlab dialect, all 4 markers stacked
                   into one 28-row update with `inv`, MeasureUpdate.m:84 generalised) -- more work per correct than the
                   reference ever does.
    Both on all usable host cores (one thread team per schedule), `value_1thread` beside."""
    import oracle_capi as oc
    from fbus_ekf import capi, synth
    try:
        oc.build(native=True)
        native = True
    except Exception:
        native = False
    cores = usable_cores()

    def leg(dialect, mode, secs):
        prm = capi.default_params(dialect)

        def run(Bs, threads, reps):
            orc = oc.Oracle(dialect, 18, native=native, nthreads=threads)
            nom, rot, P, prev = synth.initial_state(0, Bs, list(prm.p0_diag), 18)
            acc, gyr = synth.imu_samples(0, Bs, 0, sum(PATTERN), nom)
            frames = [synth.marker_frame(0, Bs, f, args.markers, nom, prm) for f in range(len(PATTERN))]
            ids = np.stack([f[0] for f in frames]); pos = np.stack([f[1] for f in frames]); quat = np.stack([f[2] for f in frames])
            dt = np.full(sum(PATTERN), 0.005)
            t0 = time.perf_counter()                             # one thread team runs the whole schedule
            orc.schedule(nom, rot, P, prev, PATTERN, reps, acc, gyr, dt, ids, pos, quat, mode)
            return Bs * STEPS_PER_PATTERN * reps / (time.perf_counter() - t0)

        probe = run(256, 1, 1)                                   # steps/s of one thread, short probe
        Bs1 = int(min(8192, max(256, probe * min(secs, 4.0) * 0.3 / STEPS_PER_PATTERN)))
        one = run(Bs1, 1, 1)
        BsN = 256 * cores                                        # 256 filters per thread, reps sized for ~`secs`
        reps = int(max(1, min(2000, one * cores * 0.6 * secs / (BsN * STEPS_PER_PATTERN))))
        allc = run(BsN, cores, reps)
        return allc, one, f"{BsN} filters x {reps} patterns of 0.1 s (20 predict + 3 correct each, M={args.markers})"

    ref_v, ref_1, ref_s = leg(capi.DIALECT_CPP, capi.MODE_NEAREST, seconds * 0.5)
    d = capi.DIALECT_MATLAB if args.dialect == "matlab" else capi.DIALECT_CPP
    m = capi.MODE_STACKED if args.mode == "stacked" else capi.MODE_NEAREST
    same_v, same_1, same_s = leg(d, m, seconds * 0.5)
    flags = "-march=native" if native else "generic x86-64"
    # SURVEY.md 8(d) allowed an optional Eigen-typed twin "if Eigen3 is found on the box".  There is none and none is promised: Eigen is
    # in neither image (dev container, GPU box), so a twin could be neither compiled nor checked anywhere this repository runs; the
    # dense C port (same operations, same order, plain loops instead of Eigen's expression templates) is the CPU baseline.
    eigen = "no Eigen-typed twin (Eigen3 is not in the image; nothing here could compile or check one): the dense C port is the baseline"
    return {"value": ref_v, "unit": "EKF steps/s", "cores": cores, "kind": "port", "eigen": eigen,
            "path": "reference CPU path restated: C++ dialect, UpdateCovariance+UpdateNominalState (filter.cpp:533-616) per "
                    "IMU sample, ObservationUpdate (filter.cpp:622-741) per frame = nearest marker with hysteresis, 7 rows, LDLT",
            "sample": f"{ref_s}, fp64 dense oracle port, {cores} threads, {flags}",
            "value_1thread": ref_1,
            "same_workload": {"value": same_v, "value_1thread": same_1, "unit": "EKF steps/s",
                              "path": f"the GPU run's own configuration ({args.dialect} dialect, correct mode {args.mode}: "
                                      f"{7 * args.markers if args.mode == 'stacked' else 7}-row dense update), "
                                      "MeasureUpdate.m:37-103 / ImuUpdate.m:36-82 restated",
                              "sample": f"{same_s}, fp64 dense oracle port, {cores} threads, {flags}"}}


class Workload:
    """device-resident inputs of `pool` distinct 0.1 s patterns for the filters [lo, hi) + the filter handle.
    tile > 1: the inputs and the initial nominal state are generated for (hi - lo) / tile filters and repeated `tile` times on
    the device (the 1 048 576-filter leg: every filter still has its own record and its own copy of the inputs in HBM -- the
    traffic is that of distinct filters -- but the host generates 65 536 of them, not a million)."""

    def __init__(self, torch, dev, local_rank, lo, hi, args, pool, with_cov=True, dtype=32, tile=1):
        from fbus_ekf import BatchedFilter, capi, synth
        self.torch, self.capi = torch, capi
        dialect = capi.DIALECT_MATLAB if args.dialect == "matlab" else capi.DIALECT_CPP
        self.mode = capi.MODE_STACKED if args.mode == "stacked" else capi.MODE_NEAREST
        self.B, self.M, self.dtype = hi - lo, args.markers, dtype
        assert self.B % tile == 0
        gen_hi = lo + self.B // tile
        prm = capi.default_params(dialect)
        npdt, tdt = (np.float32, torch.float32) if dtype == 32 else (np.float64, torch.float64)
        rep = (lambda t, d: t if tile == 1 else torch.cat([t] * tile, dim=d))
        f32 = lambda a, d=0: rep(torch.from_numpy(np.ascontiguousarray(a, npdt)).to(dev), d)
        nom, rot, P, prev = synth.initial_state(lo, gen_hi, list(prm.p0_diag), 18, with_cov=with_cov)
        self.pool = []
        for s in range(pool):
            acc, gyr = synth.imu_samples(lo, gen_hi, s * sum(PATTERN), sum(PATTERN), nom)
            frames = []
            for f in range(len(PATTERN)):
                ids, pos, quat = synth.marker_frame(lo, gen_hi, s * len(PATTERN) + f, self.M, nom, prm)
                frames.append((rep(torch.from_numpy(ids).to(dev), 0), f32(pos), f32(quat)))
            self.pool.append((f32(acc, 1), f32(gyr, 1), frames))            # IMU samples are (K, B, 3): filters along dim 1
        self.d_dt = torch.from_numpy(np.full(max(PATTERN), 0.005, npdt)).to(dev)
        # The handle stays on its own (non-blocking) stream; the inputs above were uploaded on torch's stream, so
        # the device is synchronised before the first launch and on both sides of every timed region.
        self.flt = BatchedFilter(self.B, prm, device=local_rank, dtype=dtype, nstate=18, order_streams=False)   # inputs are uploaded and synchronised before the timed region
        if tile > 1:
            assert P is None
            nom, rot, prev = np.tile(nom, (tile, 1)), np.tile(rot, (tile, 1)), np.tile(prev, tile)
        self.state0 = (nom, rot, P, prev)
        self.reset_state()
        _, self.bpf, total = self.flt.records()
        self.rec = torch.empty(total, dtype=torch.uint8, device=dev)  # records live in a torch tensor -> RCCL can ship them
        self.flt.attach_records(self.rec)
        torch.cuda.synchronize()

    def reset_state(self):
        nom, rot, P, prev = self.state0
        self.flt.set_state(nom, rot, P, prev)
        if P is None:
            self.flt.reset_cov()                                  # P0 diagonal written by the device (no 680 MB host array)

    def pattern(self, j, fused=False):
        """0.1 s of sensor time: 7 / 7 / 6 IMU samples, a camera frame behind each run"""
        acc, gyr, frames = self.pool[j % len(self.pool)]
        k = 0
        for f, K in enumerate(PATTERN):
            ids, pos, quat = frames[f]
            self.flt.frame(acc[k:k + K], gyr[k:k + K], self.d_dt[:K], ids, pos, quat, self.mode, fused=fused)
            k += K

    def step(self, i, fused=False):
        """one bench step: 1 s of sensor time"""
        for r in range(PATTERNS_PER_STEP):
            self.pattern(i * PATTERNS_PER_STEP + r, fused)

    def build_windows(self):
        """inputs of a whole bench step as ONE window (30 frames, 200 IMU samples) for fbus_ekf_frames_fused_dev: the same
        patterns in the same order as step(i) runs them, concatenated on the device.  Steps start at pool offset
        (i * PATTERNS_PER_STEP) % pool: one window per distinct offset."""
        torch = self.torch
        self.windows = {}
        for off in sorted({(i * PATTERNS_PER_STEP) % len(self.pool) for i in range(len(self.pool))}):
            ent = [self.pool[(off + r) % len(self.pool)] for r in range(PATTERNS_PER_STEP)]
            self.windows[off] = (torch.cat([e[0] for e in ent]), torch.cat([e[1] for e in ent]),
                                 torch.stack([f[0] for e in ent for f in e[2]]), torch.stack([f[1] for e in ent for f in e[2]]),
                                 torch.stack([f[2] for e in ent for f in e[2]]))
        self.kcount = np.array(list(PATTERN) * PATTERNS_PER_STEP, np.int32)
        self.d_dt_window = torch.full((int(self.kcount.sum()),), 0.005, dtype=self.d_dt.dtype, device=self.rec.device)

    def step_window(self, i):
        """the same bench step as ONE launch (records resident in registers for the whole second of sensor time)"""
        acc, gyr, ids, pos, quat = self.windows[(i * PATTERNS_PER_STEP) % len(self.pool)]
        self.flt.frames(self.kcount, acc, gyr, self.d_dt_window, ids, pos, quat, self.mode)


def timed(torch, fn, steps, warmup, barrier=lambda: None, before_timing=lambda: None):
    """W warm-up steps, then exactly K steps between barrier + synchronize on both sides.  Python's cyclic garbage collector
    is switched off for the duration (as `timeit` does): with torch imported a full collection takes 40-60 ms, and it fired
    in the middle of a 62 ms timed region (always at the 16th step: FBUS_BENCH_DEBUG_HOST_TIMES=1 shows the submitting
    thread standing still while the GPU runs dry).  Nothing the steps allocate is cyclic."""
    import gc
    gc_was_on = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        return _timed(torch, fn, steps, warmup, barrier, before_timing)
    finally:
        if gc_was_on:
            gc.enable()


def _timed(torch, fn, steps, warmup, barrier, before_timing):
    for i in range(warmup):
        fn(i)
    torch.cuda.synchronize()
    before_timing()
    barrier()
    torch.cuda.synchronize()
    marks = [] if os.environ.get("FBUS_BENCH_DEBUG_HOST_TIMES") == "1" else None
    t0 = time.perf_counter()
    for i in range(steps):
        fn(warmup + i)
        if marks is not None:
            marks.append(time.perf_counter() - t0)
    if marks is not None:
        marks.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    barrier()
    el = time.perf_counter() - t0
    if marks is not None:       # host-side submission time of each step (ms) and when the GPU was done
        sys.stderr.write("host submit ms per step: " + " ".join(f"{(b - a) * 1e3:.2f}" for a, b in zip([0.0] + marks, marks[:-1])) +
                         f" | submitted at {marks[-1] * 1e3:.2f} ms, GPU done at {el * 1e3:.2f} ms\n")
    return el


def roofline_block(w, pred_ms, pred_n, corr_ms, corr_n, traffic, traffic_src):
    roof, corr = _roofline_block(w, pred_ms, pred_n, corr_ms, corr_n, traffic, traffic_src)
    # which kernels actually ran (fbus_ekf_launch_info): below a quarter of the chip the per-call predict is the 3-role team
    # kernel, whose roles reload the nominal chunks and overlapping covariance chunks -- the one-wave byte model (1436 B per
    # filter) does not describe it, so only the API-priced figures are kept there
    pol = w.flt.launch_policy(M=w.M, K=max(PATTERN))
    tname = "float" if w.dtype == 32 else "double"
    if pol["roles_predict"] > 1:
        roof.update({"kernel": f"predict_team_kernel<{tname},18,{pol['roles_predict']} roles>", "achieved": roof["achieved_api"],
                     "frac": roof["frac_api"], "bytes_moved_per_launch": None, "frac_of_copy_ceiling": None, "traffic": None,
                     "traffic_source": None, "traffic_GBs": None, "byte_model": "api (SURVEY 8(d)): the team kernel's roles re-read shared chunks"})
    roof["launch_policy"] = pol
    return roof, corr


def _roofline_block(w, pred_ms, pred_n, corr_ms, corr_n, traffic, traffic_src):
    B, M = w.B, w.M
    es = 1 if w.dtype == 32 else 2                                # fp64 records and inputs are twice the bytes
    tname = "float" if w.dtype == 32 else "double"
    pred_us = pred_ms / pred_n * 1e3 if pred_n else float("nan")
    corr_us = corr_ms / corr_n * 1e3 if corr_n else float("nan")
    moved = PREDICT_BYTES_MOVED * B * es
    achieved = moved / (pred_us * 1e-6) / 1e9
    api = PREDICT_BYTES_API * B * es / (pred_us * 1e-6) / 1e9
    roof = {"bound": "hbm", "kernel": f"predict_kernel<{tname},18>", "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
            "avg_launch_us": pred_us, "launches": pred_n, "bytes_moved_per_launch": moved,
            "achieved_api": api, "frac_api": api / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": PREDICT_BYTES_API * B * es,
            "frac_of_copy_ceiling": achieved / HBM_COPY_CEILING_GBS,
            "traffic_GBs": (traffic / (pred_us * 1e-6) / 1e9) if traffic else None,
            "note": "achieved = bytes the kernel moves (1436 B per filter: 828 read, 608 written; the PMC `traffic` of the "
                    "committed profile is the cross-check) / HIP-event launch time; achieved_api prices SURVEY 8(d)'s full "
                    "record round trip (1620 B) and can pass the peak. At 65 536 filters the 52 MB of records stay in the "
                    "256 MB Infinity Cache between launches: that rate is cache + HBM, see roofline_hbm_resident"}
    cmoved = (CORRECT_BYTES_MOVED(M) - 1) * es + 1
    corr = {"kernel": f"correct_kernel<{tname},18,{'stacked' if w.mode == w.capi.MODE_STACKED else 'nearest'}>",
            "avg_launch_us": corr_us, "launches": corr_n,
            "achieved_GBs": cmoved * B / (corr_us * 1e-6) / 1e9,
            "frac": cmoved * B / (corr_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "bytes_moved_per_launch": cmoved * B,
            "achieved_api_GBs": CORRECT_BYTES_API(M) * B * es / (corr_us * 1e-6) / 1e9}
    return roof, corr



def fp64_leg(torch, dev, local_rank, args, capi):
    """the same per-call workload through the fp64 kernels (the reference's own arithmetic: common.hpp:205-247) at the batch of
    the headline: value, ms per step and the roofline of the dominant kernel from the bytes the fp64 kernels move"""
    torch.cuda.empty_cache()
    w = Workload(torch, dev, local_rank, 0, args.batch, args, 2, with_cov=(args.batch <= 131072), dtype=64)
    steps, warm = max(2, min(args.steps, 4)), 1
    w.flt.timing_enable(True, stride=2)
    el = timed(torch, w.step, steps, warm, before_timing=lambda: (w.flt.timing_reset(), w.flt._keep.clear()))
    p_ms, p_n = w.flt.timing_read(capi.KERNEL_PREDICT)
    pn_ms, pn_n = w.flt.timing_read(capi.KERNEL_PREDICT_N)
    c_ms, c_n = w.flt.timing_read(capi.KERNEL_CORRECT)
    w.flt.timing_enable(False)
    tr, src = pmc_traffic_named(f"r{PROFILE_ROUND:02d}_digest_f64_b{w.B}.json", w.B, args, "predict_kernel<double, 18")
    roof, corr = roofline_block(w, p_ms, p_n, c_ms, c_n, tr, src)
    roof.pop("note")
    # (round 4) the same frames as ONE launch each: frame2_kernel<double> -- the parked K-step predict loop + the row-split passes
    # with the record resident in registers / LDS -- instead of K + 1 launches
    w.reset_state()
    elf = timed(torch, lambda i: w.step(i, fused=True), steps, warm, before_timing=lambda: w.flt._keep.clear())
    blk = {"value": w.B * STEPS_PER_BENCH_STEP * steps / el, "unit": "EKF steps/s", "ms_per_step": el / steps * 1e3, "steps": steps,
           "dtype": "f64", "batch": w.B, "records_MB": w.B * 1600 / 1e6, "roofline": roof, "correct_kernel": corr,
           "fused_frame": {"value": w.B * STEPS_PER_BENCH_STEP * steps / elf, "unit": "EKF steps/s", "ms_per_step": elf / steps * 1e3,
                           "vs_per_call": el / elf,
                           "note": "one launch per camera frame (K predicts + correct resident: frame2_kernel<double>, 512 registers + 39 KiB of "
                                   "LDS per wave = four workgroups per CU, 120-140 bytes of scratch at N = 18); rounds 1-3 ran an fp64 frame as "
                                   "K + 1 launches"},
           "note": "per-call API, fp64 kernels (the reference's own arithmetic: same device functions instantiated for double, 1600-byte records)"}
    w.flt.close()
    del w
    # (round 6) the north star's own update through the fp64 kernels: per call, and one ENTRY-POINT call per camera frame
    # (fbus_ekf_frame_meas_fused_dev with fp64 records = fbus_ekf_predict_n_dev + the per-call update: two launches per frame instead of
    # K + 1 -- the one-launch form does not exist for fp64 records)
    ns = north_star_rows_leg(torch, dev, local_rank, args, capi, B=args.batch, steps=2, warmup=1, dtype=64,
                             only=("pixels_m4", "fused_frame_pixels_m4"))
    for name in ("pixels_m4", "fused_frame_pixels_m4"):
        if name in ns:
            blk[name] = ns[name]
    if "fused_frame_pixels_m4" in blk:
        blk["fused_frame_pixels_m4"]["launch"] = ("fbus_ekf_frame_meas_fused_dev on fp64 records: predict_n (K resident steps) + the per-call update = TWO "
                                                  "launches per camera frame (no one-launch form for fp64: 86 KiB of LDS per wave would be needed)")
        blk["fused_frame_pixels_m4"]["vs_per_call"] = blk["fused_frame_pixels_m4"]["value"] / blk["pixels_m4"]["value"]
    # the fp64 kernels past the Infinity Cache: 524 288 filters = 839 MB of 1600-byte records, two input patterns of 168 MB
    # (the fp32 leg of the same name holds 1 048 576 filters: the same bytes)
    if not args.no_hbm_leg and args.batch < MID_LEG_BATCH:
        torch.cuda.empty_cache()
        w2 = Workload(torch, dev, local_rank, 0, F64_HBM_LEG_BATCH, args, 2, with_cov=False, dtype=64, tile=F64_HBM_LEG_BATCH // 65536)
        steps2 = 2
        w2.flt.timing_enable(True, stride=2)
        el2 = timed(torch, w2.step, steps2, 1, before_timing=lambda: (w2.flt.timing_reset(), w2.flt._keep.clear()))
        p_ms, p_n = w2.flt.timing_read(capi.KERNEL_PREDICT)
        c_ms, c_n = w2.flt.timing_read(capi.KERNEL_CORRECT)
        w2.flt.timing_enable(False)
        tr2, src2 = pmc_traffic_named(f"r{PROFILE_ROUND:02d}_digest_f64_b{w2.B}.json", w2.B, args, "predict_kernel<double, 18")
        roof2, corr2 = roofline_block(w2, p_ms, p_n, c_ms, c_n, tr2, src2)
        roof2.pop("note")
        roof2.update({"batch": w2.B, "records_MB": w2.B * 1600 / 1e6, "input_patterns": 2, "steps": steps2,
                      "value": w2.B * STEPS_PER_BENCH_STEP * steps2 / el2, "unit_value": "EKF steps/s", "correct_kernel": corr2,
                      "note": "the fp64 kernels at 524 288 filters: 839 MB of records (three times the 256 MiB Infinity Cache), nothing "
                              "survives in the cache between launches -- the HBM streaming rate of the fp64 path"})
        blk["roofline_hbm_resident"] = roof2
        w2.flt.close()
        del w2
    return blk


def _pixels_sq_profile():
    """this round's committed SQ-counter digest of the pixel-row kernel (tools/pixels_prof.sh -> profiles/r06_pixels_sq.json)"""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", PIXELS_SQ)))
    except Exception:
        return None


def north_star_rows_leg(torch, dev, local_rank, args, capi, B=65536, steps=3, warmup=1, hbm_resident=False, dtype=32, only=None):
    """The north star's OWN MeasureUpdate -- flat-port refractive stereo reprojection of the ArUco corners, per-corner 2 x N
    Jacobians (fbus_ekf_correct_pixels_dev, csrc/ekf_meas.hpp) -- in the headline's mixed workload: the same 200 Hz + 30 Hz
    schedule (7 / 7 / 6 per-call predicts, then a camera frame; one bench step = 1 s of sensor time = 230 EKF steps per filter)
    with the correct launch replaced by
        pixels_m4 / pixels_m4_stereo      4 marker slots, left camera (32 reprojection rows per filter and frame) / both cameras (64)
        pixels_m16 / pixels_m16_stereo    16 marker slots (BASELINE config 5's shape): 128 / 256 rows
        corners_m4     fbus_ekf_correct_corners_dev (stereo corners triangulated through the port on the device, 12 rows per marker)
        pixels_m4_n15  as pixels_m4 with north_star's literal 15-state filter (N = 18 without the gravity block: 608-byte records)
        fused_frame_*  the same frames through fbus_ekf_frame_meas_fused_dev: ONE launch per camera frame (K predicts + the update)
    Scene: a wall of 16 markers 1.2 - 1.8 m in front of the port, image points = flat-port projections of the true corners +
    noise (synth.pixel_wall_scene), filters at rest.  Reported per case: EKF steps/s, ms per bench step, launch time of the update
    (HIP events on the handle's stream), its algorithmic bytes (2 x 796 + 68 M) and their fraction of 8 TB/s, and --
    these kernels are VALU-bound, that fraction is not what limits them -- the VALU issue fraction = wave-level VALU instructions of
    one launch (SQ_INSTS_VALU, committed rocprofv3 pass profiles/r06_pixels_sq.json) / (launch time x SIMDs x clock / 4)."""
    from fbus_ekf import BatchedFilter, synth
    torch.cuda.empty_cache()
    out = {}
    prof = _pixels_sq_profile()
    npdt, tdt = (np.float32, torch.float32) if dtype == 32 else (np.float64, torch.float64)
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, npdt)).to(dev)          # (the record / input type of this leg)
    if hbm_resident:
        # past the Infinity Cache: 524 288 filters = 419 MB of records (+ 67 MB of image points per camera); what a VALU-bound update and
        # the memory-bound predicts between the frames do when nothing stays cache-resident from launch to launch
        B, steps = 524288, 1
    cases = (("pixels_m4", 4, "pixels", 18, False, False), ("pixels_m4_stereo", 4, "pixels", 18, True, False),
             ("pixels_m16", 16, "pixels", 18, False, False), ("pixels_m16_stereo", 16, "pixels", 18, True, False),
             ("corners_m4", 4, "corners", 18, True, False), ("pixels_m4_n15", 4, "pixels", 15, False, False),
             ("fused_frame_pixels_m4", 4, "pixels", 18, False, True), ("fused_frame_pixels_m4_stereo", 4, "pixels", 18, True, True),
             ("fused_frame_pixels_m16_stereo", 16, "pixels", 18, True, True), ("fused_frame_corners_m4", 4, "corners", 18, True, True),
             ("fused_window_pixels_m4", 4, "pixels", 18, False, "window"))
    if hbm_resident:
        cases = (("pixels_m4", 4, "pixels", 18, False, False), ("fused_frame_pixels_m4", 4, "pixels", 18, False, True))
    if only is not None:
        cases = tuple(c for c in cases if c[0] in only)
    scenes = {}
    for name, slots, kind, nstate, stereo, fused in cases:
        size = 0.15
        if slots not in scenes:
            # (pixel_wall_scene REPLACES the marker map inside the parameter set by its wall of 16 markers: the parameters belong to the scene)
            prm = capi.default_params(capi.DIALECT_MATLAB if args.dialect == "matlab" else capi.DIALECT_CPP)
            prm.marker_size = size
            scenes[slots] = (prm,) + tuple(synth.pixel_wall_scene(B, slots, prm, size, seed=9, stereo=True))
        prm, nom, rot, ids, left, right = scenes[slots]
        acc, gyr = synth.imu_samples(0, B, 0, sum(PATTERN), nom)
        d_acc, d_gyr = f32(acc), f32(gyr)
        d_dt = torch.full((max(PATTERN),), 0.005, dtype=tdt, device=dev)
        d_ids, d_left, d_right = torch.from_numpy(ids).to(dev), f32(left), f32(right)
        if fused == "window":
            # one launch per 0.1 s pattern x 10: a bench step as ONE window of 30 frames (200 IMU samples) -- the same frame's image points
            # 30 times over (every frame reads its own copy), the pattern's IMU samples 10 times over
            nfr = len(PATTERN) * PATTERNS_PER_STEP
            w_kc = list(PATTERN) * PATTERNS_PER_STEP
            w_acc, w_gyr = d_acc.repeat(PATTERNS_PER_STEP, 1, 1), d_gyr.repeat(PATTERNS_PER_STEP, 1, 1)
            w_dt = torch.full((sum(w_kc),), 0.005, dtype=tdt, device=dev)
            w_ids, w_left = d_ids.unsqueeze(0).repeat(nfr, 1, 1).contiguous(), d_left.unsqueeze(0).repeat(nfr, 1, 1, 1).contiguous()
            w_right = d_right.unsqueeze(0).repeat(nfr, 1, 1, 1).contiguous() if stereo else None
        nvis = float((ids >= 0).sum(axis=1).mean())
        prev0 = np.zeros(B, np.int32)
        with BatchedFilter(B, prm, device=local_rank, order_streams=False, nstate=nstate, dtype=dtype) as flt:
            flt.set_state(nom, rot, None, prev0)
            flt.reset_cov()
            torch.cuda.synchronize()

            def update():
                if kind == "pixels":
                    flt.correct_pixels(d_ids, d_left, d_right if stereo else None)
                else:
                    flt.correct_corners(d_ids, d_left, d_right, capi.VIS_REFRACTIVE, capi.MODE_STACKED)

            def bench_step(i):
                if fused == "window":
                    flt.frames_meas(w_kc, w_acc, w_gyr, w_dt, w_ids, w_left, w_right, capi.MEAS_PIXELS if kind == "pixels" else capi.MEAS_CORNERS,
                                    capi.VIS_REFRACTIVE, capi.MODE_STACKED)
                    return
                for r in range(PATTERNS_PER_STEP):
                    k = 0
                    for K in PATTERN:
                        if fused:                                    # one launch per camera frame: K predicts + the update, record resident
                            flt.frame_meas(d_acc[k:k + K], d_gyr[k:k + K], d_dt[:K], d_ids, d_left, d_right if stereo else None,
                                           capi.MEAS_PIXELS if kind == "pixels" else capi.MEAS_CORNERS, capi.VIS_REFRACTIVE, capi.MODE_STACKED)
                        else:
                            for j in range(K):
                                flt.predict(d_acc[k + j], d_gyr[k + j], d_dt[:1])
                            update()
                        k += K
            el = timed(torch, bench_step, steps, warmup, before_timing=lambda: flt._keep.clear())   # throughput: no event brackets inside the timed region
            flt.timing_enable(True, stride=1)                     # launch times: one more step with a bracket around every launch
            flt.timing_reset()
            bench_step(0)
            flt.sync()
            u_ms, u_n = flt.timing_read(capi.KERNEL_FRAME if fused else capi.KERNEL_CORRECT_CORNERS)
            p_ms, p_n = flt.timing_read(capi.KERNEL_PREDICT)
            applied = float(flt.applied().mean())
            finite = bool(np.isfinite(flt.get_state()[0]).all())
            us = u_ms / max(u_n, 1) * 1e3
            if fused and u_n == 0:                                # fp64 records: the entry point ran predict_n + the per-call update (two launches)
                pn_ms, pn_n = flt.timing_read(capi.KERNEL_PREDICT_N)
                c_ms, c_n = flt.timing_read(capi.KERNEL_CORRECT_CORNERS)
                u_n = c_n
                us = (pn_ms + c_ms) / max(c_n, 1) * 1e3
            rows = nvis * ((16 if stereo else 8) if kind == "pixels" else 12)
            rec_b = (dtype // 8) * (27 + nstate * (nstate + 1) // 2 + 1)
            bytes_api = 2 * rec_b + (4 + 8 * dtype // 4) * slots      # record round trip + the slot's image points (68 B per slot in fp32)
            blk = {"value": B * STEPS_PER_BENCH_STEP * steps / el, "unit": "EKF steps/s", "ms_per_step": el / steps * 1e3, "steps": steps,
                   "batch": B, "nstate": nstate, "marker_slots": slots, "markers_in_view_mean": nvis, "rows_per_filter_and_frame": rows,
                   "camera": ("stereo" if stereo else "left") if kind == "pixels" else "stereo (triangulated)",
                   "filters_updated_frac": applied, "state_finite": finite}
            if fused == "window":
                nfr = len(PATTERN) * PATTERNS_PER_STEP
                blk.update({"launch": "fbus_ekf_frames_meas_fused_dev: a bench step (30 camera frames, 200 ImuUpdates) in ONE launch, the record "
                                      "resident throughout (offline replay)",
                            "frame_avg_launch_us": us, "window_avg_launch_us": us * nfr, "window_launches": u_n / nfr,      # (the library counts a window as its frames)
                            "us_per_ekf_step": us * nfr / STEPS_PER_BENCH_STEP,
                            "note": "extra to the metric: one record round trip per 230 EKF steps"})
            elif fused:
                kavg = sum(PATTERN) / len(PATTERN)
                blk.update({"launch": "fbus_ekf_frame_meas_fused_dev: K = 7 / 7 / 6 ImuUpdates + the update in ONE launch per camera frame "
                                      "(frame_meas_kernel: record resident, covariance parked in LDS across the fold)",
                            "frame_avg_launch_us": us, "frame_launches": u_n, "us_per_ekf_step": us / (kavg + 1),
                            "frame_bytes_per_filter_actual": 2 * rec_b + 68 * slots + 28 * kavg,
                            "note": "extra to the metric: one record round trip per FRAME; priced per step by SURVEY 8(d) it would read > 1 of the roofline"})
            else:
                blk.update({"update": ("fbus_ekf_correct_pixels_dev (" + ("stereo" if stereo else "left camera") + ")") if kind == "pixels"
                                      else "fbus_ekf_correct_corners_dev (refractive, stacked)",
                            "update_avg_launch_us": us, "update_launches": u_n, "predict_avg_launch_us": p_ms / max(p_n, 1) * 1e3,
                            "update_bytes_per_filter_api": bytes_api, "update_algorithmic_GBs": bytes_api * B / (us * 1e-6) / 1e9,
                            "update_frac_of_hbm_peak": bytes_api * B / (us * 1e-6) / 1e9 / HBM_PEAK_GBS})
                if kind == "pixels":
                    blk["reprojection_rows_per_s"] = rows * B / (us * 1e-6)
                    if prof and int(prof.get("batch", 0)) == B and int(prof.get("marker_slots", 0)) == slots and \
                            prof.get("camera", "left") == ("stereo" if stereo else "left"):
                        insts, mhz = float(prof["counters_per_launch"]["SQ_INSTS_VALU"]), float(prof.get("clock_MHz") or 2400.0)
                        simds = int(prof.get("simds", 1024))
                        blk.update({"valu_insts_per_launch": insts, "clock_MHz": mhz, "source": "profiles/" + PIXELS_SQ,
                                    "valu_issue_frac": insts / (us * 1e-6 * simds * mhz * 1e6 / 4.0),
                                    "valu_issue_frac_profiled_run": prof.get("valu_issue_frac_kernel_trace"),
                                    "sq_active_inst_valu_over_wave_cycles": prof.get("SQ_ACTIVE_INST_VALU_over_SQ_WAVE_CYCLES")})
            flt.timing_enable(False)
        out[name] = blk
    if hbm_resident:
        out["records_MB"] = B * 800 / 1e6
    out["note"] = ("same 200 Hz + 30 Hz schedule as `value`, the camera frame applied as reprojection rows (pixels_*: left camera 2 rows per corner, "
                   "*_stereo 4 rows per corner) or as triangulated corner rows (corners_m4) instead of marker poses; per call (one launch per EKF "
                   "step) or fused_frame_* (one launch per camera frame); VALU-bound updates: read valu_issue_frac, not the byte fraction")
    return out


_REAL_STDOUT = None


def emit(line):
    """the one JSON line, on the process's original stdout (see main)"""
    sys.stdout.flush()
    os.write(_REAL_STDOUT if _REAL_STDOUT is not None else 1, (line + "\n").encode())


LINE_LIMIT = 6144                   # the driver's parser lost round 5's 20.7 KB line: the stdout line stays below this
DETAIL_FILE = "bench_detail.json"   # everything else, beside bench.py (and as one `bench_detail: ...` line on stderr)


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _r(x, digits=5):
    """floats to `digits` significant digits (the line carries numbers, not noise)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        return float(f"{x:.{digits}g}") if np.isfinite(x) else None
    if isinstance(x, dict):
        return {k: _r(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, digits) for v in x]
    return x


def compact_line(out):
    """The ONE stdout line the driver parses, from the full result `out`: the contract's keys, `roofline` (dominant kernel, with the
    HBM-resident leg folded in as `hbm_resident`), `cpu_baseline`, and one short block each for `fp64` and `north_star`.  Everything
    else -- the other legs, per-row notes, launch policy -- is `bench_detail.json`.  Blocks are dropped from the end of `optional`
    until the line fits LINE_LIMIT (never `roofline` / `cpu_baseline` / `config`)."""
    head = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                       "vs_baseline", "dtype", "data"))
    line = {}
    cfg = out.get("config") or {}
    line["config"] = _pick(cfg, ("workload", "batch_per_gpu", "total_filters", "markers", "ekf_steps_per_bench_step", "parallelism"))
    roof = out.get("roofline") or {}
    r = _pick(roof, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "avg_launch_us", "launches",
                     "bytes_moved_per_launch", "achieved_api", "frac_api"))
    hb = out.get("roofline_hbm_resident")
    if hb:
        r["hbm_resident"] = _pick(hb, ("batch", "frac", "achieved", "traffic", "avg_launch_us", "frac_of_copy_ceiling"))
    line["roofline"] = r
    cb = out.get("cpu_baseline")
    line["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "sample", "value_1thread", "path")) if cb else None
    optional = {}
    if out.get("correct_kernel"):
        optional["correct_kernel"] = _pick(out["correct_kernel"], ("kernel", "avg_launch_us", "frac"))
    for k in ("fused_frame", "fused_window"):
        if out.get(k):
            optional[k] = _pick(out[k], ("value", "ms_per_step"))
    f64 = out.get("fp64")
    if f64:
        optional["fp64"] = dict(_pick(f64, ("value", "ms_per_step")),
                                predict_us=(f64.get("roofline") or {}).get("avg_launch_us"), frac=(f64.get("roofline") or {}).get("frac"),
                                fused_frame=(f64.get("fused_frame") or {}).get("value"),
                                pixels_m4=(f64.get("pixels_m4") or {}).get("value"),
                                fused_frame_pixels_m4=(f64.get("fused_frame_pixels_m4") or {}).get("value"))
    ns = out.get("north_star_rows")
    if ns:
        blk = {}
        for name in ("pixels_m4", "pixels_m4_stereo", "fused_frame_pixels_m4"):
            c = ns.get(name)
            if c:
                blk[name] = dict(_pick(c, ("value", "ms_per_step")),
                                 launch_us=c.get("update_avg_launch_us", c.get("frame_avg_launch_us")))
        optional["north_star"] = blk
    for k in ("gather_ms", "state_finite"):
        if k in out:
            optional[k] = out[k]
    if out.get("legs_skipped"):
        optional["legs_skipped"] = [s.split(":")[0] for s in out["legs_skipped"]]
    optional["detail"] = DETAIL_FILE
    line.update(optional)
    line = dict(head, **_r(line))                     # value / ms_per_step at full precision (the driver cross-checks them)
    text = json.dumps(line, separators=(",", ":"))
    drop = [k for k in ("legs_skipped", "gather_ms", "correct_kernel", "fused_window", "fused_frame", "fp64", "north_star") if k in line]
    while len(text) >= LINE_LIMIT and drop:
        line.pop(drop.pop(0))
        text = json.dumps(line, separators=(",", ":"))
    if len(text) >= LINE_LIMIT:                       # free text is the only thing left that can be long
        for blk, key in (("cpu_baseline", "path"), ("cpu_baseline", "sample"), ("config", "workload")):
            if isinstance(line.get(blk), dict) and isinstance(line[blk].get(key), str):
                line[blk][key] = line[blk][key][:200]
        text = json.dumps(line, separators=(",", ":"))
    assert len(text) < LINE_LIMIT, len(text)
    return text


def emit_result(out, detail_path=None, detail_stderr=False):
    """full result -> bench_detail.json (or --detail-file) [+ stderr with --detail-stderr]; compact line -> stdout"""
    detail = json.dumps(out)
    written = []
    for path in ([detail_path] if detail_path else [os.path.join(ROOT, DETAIL_FILE), os.path.join(ROOT, "gpurun_out", DETAIL_FILE)]):
        try:
            if os.path.isdir(os.path.dirname(os.path.abspath(path))):
                with open(path, "w") as f:
                    f.write(detail + "\n")
                written.append(path)
        except OSError as e:
            sys.stderr.write(f"bench.py: could not write {path}: {e}\n")
    written = ", ".join(written)
    # stderr: where the full result went (the whole object only on request: a 20 KB line in front of the one the driver parses is a
    # risk not worth taking, and the committed copy is profiles/rNN_bench_detail.json)
    if detail_stderr:
        sys.stderr.write("bench_detail: " + detail + "\n")
    else:
        sys.stderr.write(f"bench.py: full result ({len(detail)} bytes) -> {written or 'nowhere (no writable path)'}\n")
    sys.stderr.flush()
    emit(compact_line(out))


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}: launch {args.gpus} ranks or pass --gpus {world}\n")
        sys.exit(2)

    # stdout carries exactly ONE line, the JSON: everything else that libraries print there (RCCL's version banner at
    # communicator creation sits in the C stdio buffer until exit, i.e. BEHIND the JSON line) is sent to stderr for the
    # whole run, and the JSON line is written to the original stdout at the end.
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)

    import torch
    from fbus_ekf import capi, shard
    dist = None
    # FBUS_BENCH_DEBUG_SHARED_GPU=1: rehearsal of the N > 1 code path on a 1-GPU box (all ranks on cuda:0,
    # gloo instead of RCCL for the control collectives); never used by the driver's multi-GPU runs.
    shared = world > 1 and os.environ.get("FBUS_BENCH_DEBUG_SHARED_GPU") == "1"
    if shared:
        local_rank = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    if not shared and local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants cuda:{local_rank} but the box has {torch.cuda.device_count()} GPU(s)")
    torch.cuda.set_device(local_rank)                        # before the process group: RCCL binds to the current device
    # FBUS_BENCH_FORCE_RCCL=1: create the RCCL process group even with ONE rank, so that a 1-GPU box runs every collective of
    # the N > 1 path (barrier, all-reduce, all-gather of the records) through RCCL itself (tests/test_bench_frontdoor.py)
    force_pg = os.environ.get("FBUS_BENCH_FORCE_RCCL") == "1"
    if world > 1 or force_pg:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if shared:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    ctl_dev = "cpu" if (dist is not None and dist.get_backend() == "gloo") else dev

    strong = args.total_batch > 0
    if strong:
        lo, hi = shard.shard_range(args.total_batch, rank, world)
    else:
        lo, hi = shard.weak_range(args.batch, rank)          # this rank's filters in the global index space
    total_filters = args.total_batch if strong else args.batch * world

    def barrier():
        if dist is not None:
            dist.barrier()

    if args.only_pixels:
        emit(json.dumps({"north_star_rows": north_star_rows_leg(torch, dev, local_rank, args, capi),
                         "north_star_rows_hbm_resident": None if args.no_hbm_leg else north_star_rows_leg(torch, dev, local_rank, args, capi, hbm_resident=True)}))
        return
    w = Workload(torch, dev, local_rank, lo, hi, args, POOL if args.tile == 1 else 2, with_cov=(hi - lo) <= 131072 and args.tile == 1,
                 tile=args.tile, dtype=args.dtype)
    flt = w.flt
    if strong:
        flt.set_policy_batch(args.total_batch)      # the same kernels whatever the shard layout (fbus_ekf_set_policy_batch)
    frames_timed = args.steps * len(PATTERN) * PATTERNS_PER_STEP
    # HIP-event brackets on every stride-th camera frame (a pair costs ~8 us of stream time: <= 1.5 % of the timed region)
    stride = max(6, min(16, frames_timed // 10))
    flt.timing_enable(args.kernel_timing == "on", stride=stride)
    bench_step = w.step
    if args.graphs:
        graph_ids = [flt.graph_capture(lambda j=j: w.pattern(j)) for j in range(POOL)]

        def bench_step(i):
            for r in range(PATTERNS_PER_STEP):
                flt.graph_launch(graph_ids[(i * PATTERNS_PER_STEP + r) % POOL])

    def clear():
        flt.timing_reset()
        flt._keep.clear()

    elapsed = timed(torch, bench_step, args.steps, args.warmup, barrier, clear)
    elapsed = shard.max_over_ranks(elapsed, dist, world, ctl_dev)

    if args.graphs and args.kernel_timing == "on":          # events cannot live inside a graph: short eager pass
        flt.timing_reset()
        for i in range(4):
            w.pattern(i)
        torch.cuda.synchronize()
    pred_ms, pred_n = flt.timing_read(capi.KERNEL_PREDICT)
    corr_ms, corr_n = flt.timing_read(capi.KERNEL_CORRECT)
    flt.timing_enable(False)

    # ---- extra (never `value`): the same frames through the fused one-launch-per-frame kernel ----
    w.reset_state()
    fused_elapsed = timed(torch, lambda i: w.step(i, fused=True), args.steps, args.warmup, barrier)
    fused_elapsed = shard.max_over_ranks(fused_elapsed, dist, world, ctl_dev)
    flt._keep.clear()

    # ---- extra (never `value`): the same steps, each as ONE launch of the frame-window kernel ----
    w.reset_state()
    w.build_windows()
    window_elapsed = timed(torch, w.step_window, args.steps, args.warmup, barrier)
    window_elapsed = shard.max_over_ranks(window_elapsed, dist, world, ctl_dev)
    flt._keep.clear()
    w.windows = None

    # ---- the single end-of-run collective: gather the packed records (timed separately) ----
    # Through the LIBRARY: fbus_ekf_gather issues ncclAllGather (equal shards) / a grouped ncclBroadcast per rank (ragged shards)
    # on the handle's stream with the library's own communicator -- also with one rank, so that a 1-GPU run exercises the same
    # call.  torch.distributed only carries the 128-byte unique id.  The gloo rehearsal on a shared GPU
    # (FBUS_BENCH_DEBUG_SHARED_GPU) cannot use RCCL (two ranks on one device) and keeps the torch path.
    gather_via = "torch.distributed"
    sizes = shard.record_bytes_of_ranks(args.total_batch, world, w.bpf) if strong else None
    if not shared and os.environ.get("FBUS_BENCH_TORCH_GATHER") != "1":
        failed = 0.0
        try:
            shard.native_comm_init(flt, rank, world, dist, ctl_dev)
        except Exception as e:                       # no RCCL on this box: say so, fall back
            failed = 1.0
            sys.stderr.write(f"bench.py: native RCCL communicator unavailable on rank {rank} ({e}); gathering through torch.distributed\n")
        # the choice is collective: one rank without the communicator and everybody takes the torch path
        if shard.max_over_ranks(failed, dist, world, ctl_dev) == 0.0:
            gather_via = "fbus_ekf_gather (RCCL inside the library)"
    torch.cuda.synchronize()
    barrier()
    tg = time.perf_counter()
    if gather_via.startswith("fbus_ekf_gather"):
        gathered = shard.gather_records_native(flt, w.rec, sizes, world)
    else:
        gathered = shard.gather_records(w.rec.cpu() if ctl_dev == "cpu" else w.rec, dist, world) if not strong else \
            shard.gather_records_ragged(w.rec.cpu() if ctl_dev == "cpu" else w.rec, dist, world, ctl_dev)
    torch.cuda.synchronize()
    gather_ms = (time.perf_counter() - tg) * 1e3
    if gather_via.startswith("fbus_ekf_gather"):
        assert torch.equal(gathered[rank], w.rec), "gathered records differ from this rank's own"
    nomf, _, Pf, _ = flt.get_state()
    finite = bool(np.isfinite(nomf).all() and np.isfinite(Pf).all())
    finite = shard.max_over_ranks(0.0 if finite else 1.0, dist, world, ctl_dev) == 0.0
    ranks_seen = int(shard.sum_over_ranks(1.0, dist, world, ctl_dev))

    if rank == 0:
        total_steps = total_filters * STEPS_PER_BENCH_STEP * args.steps
        value = total_steps / elapsed
        traffic, traffic_src = pmc_traffic(w.B, args, world)
        roof, corr = roofline_block(w, pred_ms, pred_n, corr_ms, corr_n, traffic, traffic_src)
        out = {
            "metric": f"EKF steps/s (ImuUpdate+MeasureUpdate), batch={total_filters if strong else w.B}, {w.M} markers",
            "value": value, "unit": "EKF steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f32" if args.dtype == 32 else "f64", "data": "synthetic",
            "dtype_note": "f32 is the arithmetic BASELINE.json's north_star quotes the metric on; the reference itself computes in "
                          "double (common.hpp:205-247, filter.cpp:533-741) -- the same workload through the fp64 kernels is the "
                          "`fp64` block of this line",
            "config": {"workload": (f"{args.total_batch} filters in total, sharded over {world} GPU(s) "
                                    f"({w.B} on rank 0)" if strong else f"batch {w.B} filters/GPU") +
                                   f", 200 Hz IMU + 30 Hz stereo (7/7/6 predicts per correct, 1 s of sensor time per bench step), "
                                   f"{w.M} markers/frame, N=18, {args.dialect} dialect, correct mode {args.mode}, "
                                   "per-call API (one launch per EKF step)" + (", replayed from HIP graphs" if args.graphs else ""),
                       "batch_per_gpu": w.B, "total_filters": total_filters, "markers": w.M,
                       "ekf_steps_per_bench_step": STEPS_PER_BENCH_STEP,
                       "parallelism": f"independent filter shards x{world}, one RCCL gather at the end",
                       "ranks_seen": ranks_seen,
                       "collective_backend": (dist.get_backend() if dist is not None else None)},
            "roofline": roof, "correct_kernel": corr,
            "fused_frame": {"value": total_steps / fused_elapsed, "unit": "EKF steps/s",
                            "ms_per_step": fused_elapsed / args.steps * 1e3,
                            "note": "same frames, one launch per camera frame (K predicts + correct, records "
                                    "resident in registers); moves 1/(K+1) of the per-call bytes, VALU-bound; "
                                    "reported beside, never instead of, the per-call number"},
            "fused_window": {"value": total_steps / window_elapsed, "unit": "EKF steps/s",
                             "ms_per_step": window_elapsed / args.steps * 1e3, "frames_per_launch": len(PATTERN) * PATTERNS_PER_STEP,
                             "note": "same steps, one launch per bench step (fbus_ekf_frames_fused_dev: 30 camera frames = 200 "
                                     "predicts + 30 corrects with the records resident in registers from the first load to the "
                                     "last store) -- the offline-replay form; beside, never instead of, the per-call number"},
            "gather_ms": gather_ms, "gathered_bytes": int(sum(g.numel() for g in gathered)), "gather_via": gather_via,
            "state_finite": finite,
        }
    flt.close()
    del w

    # ---- roofline legs at other batch sizes on this GPU (rank 0, N = 1 only) ----
    def batch_leg(batch, pool, tile, steps2, note, pmc_batch=None):
        torch.cuda.empty_cache()
        w2 = Workload(torch, dev, local_rank, 0, batch, args, pool, with_cov=False, tile=tile)
        warm2 = 1
        w2.flt.timing_enable(True, stride=2)
        el2 = timed(torch, w2.step, steps2, warm2, before_timing=lambda: (w2.flt.timing_reset(), w2.flt._keep.clear()))
        p_ms, p_n = w2.flt.timing_read(capi.KERNEL_PREDICT)
        c_ms, c_n = w2.flt.timing_read(capi.KERNEL_CORRECT)
        w2.flt.timing_enable(False)
        w2.reset_state()
        elf = timed(torch, lambda i: w2.step(i, fused=True), steps2, warm2, before_timing=lambda: w2.flt._keep.clear())
        tr2, src2 = pmc_traffic(batch, args, 1)
        roof2, corr2 = roofline_block(w2, p_ms, p_n, c_ms, c_n, tr2, src2)
        roof2.pop("note")
        roof2.update({"batch": batch, "records_MB": batch * 800 / 1e6, "input_patterns": pool, "steps": steps2,
                      "value": batch * STEPS_PER_BENCH_STEP * steps2 / el2, "unit_value": "EKF steps/s",
                      "correct_kernel": corr2,
                      "fused_frame_value": batch * STEPS_PER_BENCH_STEP * steps2 / elf, "note": note})
        w2.flt.close()
        del w2
        return roof2

    if rank == 0:
        out["roofline_hbm_resident"] = out["roofline_b262144"] = out["fp64"] = out["north_star_rows"] = None
        # legs this line does NOT carry, and why (an N > 1 line has none of the single-GPU side legs: say so instead of nulls)
        skipped = []
        if world > 1 or strong:
            why = f"{world} ranks" if world > 1 else "strong-scaling run (--total-batch)"
            skipped += [f"{leg}: single-GPU side leg, not run with {why}" for leg in
                        ("roofline_hbm_resident", "roofline_b262144", "fp64", "north_star_rows", "cpu_baseline")]
        else:
            if args.no_hbm_leg or args.batch >= MID_LEG_BATCH:
                skipped += [f"{leg}: " + ("--no-hbm-leg" if args.no_hbm_leg else f"--batch {args.batch} is itself a large-batch run")
                            for leg in ("roofline_hbm_resident", "roofline_b262144")]
            if args.no_extra_legs:
                skipped += ["fp64: --no-extra-legs", "north_star_rows: --no-extra-legs"]
            if args.no_cpu_baseline:
                skipped += ["cpu_baseline: --no-cpu-baseline"]
        if world == 1 and not strong and out["roofline"].get("traffic") is None:
            skipped += [f"roofline.traffic: no profiles/r{PROFILE_ROUND:02d}_digest_b{args.batch}.json taken on this configuration (digests of earlier rounds' kernels are not quoted)"]
        out["legs_skipped"] = skipped
        if world == 1 and not args.no_hbm_leg and not strong and args.batch < MID_LEG_BATCH:
            out["roofline_hbm_resident"] = batch_leg(
                HBM_LEG_BATCH, 2, 16, max(2, min(args.steps, 3)),
                "same bench pattern at 1 048 576 filters on one GPU: 839 MB of records (three times the 256 MiB Infinity Cache) + "
                "two input patterns of 168 MB each -- nothing survives in the cache from one launch to the next, so this is the "
                "HBM streaming rate.  `frac` is against the 8 TB/s spec, `frac_of_copy_ceiling` against the 6.29 TB/s a float4 "
                "copy reaches on this part (MI355X_MICROARCH.md)")
            out["roofline_b262144"] = batch_leg(
                MID_LEG_BATCH, 1, 1, max(2, min(args.steps, 4)),
                "262 144 filters: 210 MB of records -- PARTLY CACHE-RESIDENT (they fit the 268 MB Infinity Cache; round 2 reported "
                "this leg as the HBM figure): kept for continuity, the HBM claim is roofline_hbm_resident")
        if world == 1 and not args.no_extra_legs and not strong:
            out["fp64"] = fp64_leg(torch, dev, local_rank, args, capi)
            out["north_star_rows"] = north_star_rows_leg(torch, dev, local_rank, args, capi)
            if not args.no_hbm_leg:
                out["north_star_rows_hbm_resident"] = north_star_rows_leg(torch, dev, local_rank, args, capi, hbm_resident=True)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        emit_result(out, args.detail_file, args.detail_stderr)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

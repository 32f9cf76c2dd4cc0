"""bench.py's multi-GPU front door.

CPU part: `--gpus N` on a box with fewer GPUs fails loudly (no JSON line, non-zero exit), a launcher whose
WORLD_SIZE disagrees with --gpus is refused, and the ragged gather used by the strong-scaling mode works over gloo.
GPU part (one GPU is enough): the N > 1 code path rehearsed with all ranks on cuda:0 and gloo for the control
collectives (FBUS_BENCH_DEBUG_SHARED_GPU=1), weak and strong scaling, started the way the driver's `--gpus N` does.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(kw)
    return env


def test_gpus_flag_without_enough_devices_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("this box really has 8 GPUs")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "1", "--warmup", "0"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "--gpus 8 requested" in r.stderr and "refusing" in r.stderr
    assert "\"metric\"" not in r.stdout                      # no number is reported


def test_world_size_must_match_gpus_flag():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=_env(WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=4 but --gpus 2" in r.stderr
    r = subprocess.run([sys.executable, BENCH], env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2


def _ragged_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "fbus-ekf_amd"))
    from fbus_ekf import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard.shard_range(200, rank, world)            # 200 filters over 3 ranks: 64 / 64 / 72 (ragged)
    local = torch.arange(lo * 800, hi * 800, dtype=torch.int64).to(torch.uint8)
    out = shard.gather_records_ragged(local, dist, world, "cpu")
    n = shard.sum_over_ranks(1.0, dist, world, "cpu")
    if rank == 0:
        q.put(([o.numpy() for o in out], n))
    dist.barrier()
    dist.destroy_process_group()


def test_ragged_gather_over_gloo():
    import socket
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "fbus-ekf_amd"))
    from fbus_ekf import shard
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ragged_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out, n = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert n == 3.0
    covered = 0
    for r in range(world):
        lo, hi = shard.shard_range(200, r, world)
        assert lo % 64 == 0 and out[r].size == (hi - lo) * 800
        assert np.array_equal(out[r], (np.arange(lo * 800, hi * 800) % 256).astype(np.uint8))
        covered += hi - lo
    assert covered == 200


def _run_bench(extra, env):
    """runs bench.py; checks the compact stdout line (the driver's view) and returns the FULL result (--detail-file)"""
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        detail = os.path.join(td, "detail.json")
        r = subprocess.run([sys.executable, BENCH, "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-hbm-leg", "--no-extra-legs",
                            "--detail-file", detail] + extra, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{") and "\"metric\"" in l]
        assert len(lines) == 1 and r.stdout.strip().splitlines()[-1] == lines[0], r.stdout[-2000:]
        assert len(lines[0]) < 6144
        line = json.loads(lines[0])
        full = json.load(open(detail))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert line[k] == full[k], k
    assert line["config"]["workload"] == full["config"]["workload"] and "roofline" in line and "cpu_baseline" in line
    assert line["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-4)
    return full


@pytest.mark.gpu
def test_two_rank_rehearsal_weak_and_strong_scaling():
    """`python bench.py --gpus 2` starts two rank processes by itself (child torch.distributed.run); with
    FBUS_BENCH_DEBUG_SHARED_GPU=1 both sit on cuda:0 and use gloo, which exercises everything but RCCL itself."""
    env = _env(FBUS_BENCH_DEBUG_SHARED_GPU="1")
    one = _run_bench(["--batch", "4096"], _env())
    assert one["n_gpus"] == 1 and one["scaling"] == "weak" and one["config"]["ranks_seen"] == 1
    assert one["state_finite"] and one["roofline"]["frac"] > 0 and one["roofline"]["launches"] > 0
    # 4096 filters = 64 tiles: the per-call predict is the 3-role team kernel there, and the line says so instead of pricing it with
    # the one-wave kernel's byte model (round-3 advisor finding)
    pol = one["roofline"]["launch_policy"]
    assert pol["roles_predict"] == (3 if 64 <= pol["simds"] // 4 else 1) and pol["policy_batch"] == 4096
    if pol["roles_predict"] > 1:
        assert "predict_team_kernel" in one["roofline"]["kernel"] and one["roofline"]["bytes_moved_per_launch"] is None
        assert one["roofline"]["frac"] == one["roofline"]["frac_api"]
    else:
        assert one["roofline"]["bytes_moved_per_launch"] == 1436 * 4096
    assert any(l.startswith("roofline_hbm_resident: --no-hbm-leg") for l in one["legs_skipped"])
    # the metric names the run's own batch; the gather of a single rank already goes through the library's RCCL entry point
    assert "batch=4096" in one["metric"] and one["gather_via"].startswith("fbus_ekf_gather") and one["gathered_bytes"] == 4096 * 800
    weak = _run_bench(["--gpus", "2", "--batch", "4096"], env)
    assert weak["n_gpus"] == 2 and weak["scaling"] == "weak" and weak["config"]["ranks_seen"] == 2
    assert weak["config"]["total_filters"] == 8192 and weak["config"]["batch_per_gpu"] == 4096
    assert weak["gathered_bytes"] == 2 * 4096 * 800 and weak["state_finite"]
    assert weak["config"]["collective_backend"] == "gloo" and weak["cpu_baseline"] is None
    # an N > 1 line names the single-GPU side legs it does not carry
    assert {l.split(":")[0] for l in weak["legs_skipped"]} >= {"roofline_hbm_resident", "fp64", "north_star_rows", "cpu_baseline"}
    strong = _run_bench(["--gpus", "2", "--total-batch", "8192"], env)
    assert strong["n_gpus"] == 2 and strong["scaling"] == "strong" and strong["config"]["ranks_seen"] == 2
    assert strong["config"]["total_filters"] == 8192 and strong["config"]["batch_per_gpu"] == 4096
    assert strong["gathered_bytes"] == 8192 * 800 and strong["state_finite"]
    # strong scaling keys the kernel-family choice on the whole job (fbus_ekf_set_policy_batch): the same kernels on every shard layout
    assert strong["roofline"]["launch_policy"]["policy_batch"] == 8192
    # value counts the filters of ALL ranks
    assert strong["value"] == pytest.approx(8192 * 230 * 3 / (strong["ms_per_step"] * 3e-3), rel=1e-6)


@pytest.mark.gpu
def test_single_rank_through_rccl_itself():
    """one rank, but with the RCCL process group created (FBUS_BENCH_FORCE_RCCL=1): the barrier, the max / sum all-reduces and the
    all-gather of the packed records that the N > 1 path issues all run through RCCL on this GPU"""
    out = _run_bench(["--batch", "4096"], _env(FBUS_BENCH_FORCE_RCCL="1", MASTER_PORT="29547"))
    assert out["n_gpus"] == 1 and out["config"]["collective_backend"] == "nccl" and out["config"]["ranks_seen"] == 1
    assert out["gathered_bytes"] == 4096 * 800 and out["state_finite"] and out["gather_ms"] > 0
    assert out["gather_via"].startswith("fbus_ekf_gather")          # the records travel through the library's own communicator


@pytest.mark.gpu
def test_native_gather_entry_point_single_rank():
    """fbus_ekf_comm_unique_id / comm_init / gather with one rank (what a 1-GPU box can run of the N > 1 path): the equal-shard
    form (ncclAllGather) and the ragged form (grouped ncclBroadcast) both return this rank's records bit for bit; a wrong
    bytes_of_rank entry and a gather without a communicator are refused"""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "fbus-ekf_amd"))
    from fbus_ekf import BatchedFilter, capi, synth
    B = 1000                                           # ragged: 15.6 tiles -> records are 16 tiles long
    prm = capi.default_params(0)
    nom, rot, P, prev = synth.initial_state(0, B, list(prm.p0_diag), 18)
    with BatchedFilter(B, prm) as flt:
        flt.set_state(nom, rot, P, prev)
        ptr, bpf, total = flt.records()
        assert total == 1024 * bpf
        rec = torch.empty(total, dtype=torch.uint8, device="cuda:0")
        flt.attach_records(rec)
        out = torch.zeros(total, dtype=torch.uint8, device="cuda:0")
        with pytest.raises(capi.FbusError):
            flt.gather(out)                            # no communicator yet
        flt.comm_init(BatchedFilter.comm_unique_id(), 0, 1)
        flt.gather(out); flt.sync()
        assert torch.equal(out, rec)
        out.zero_()
        flt.gather(out, [total]); flt.sync()
        assert torch.equal(out, rec)
        with pytest.raises(capi.FbusError):
            flt.gather(out, [total - 64])


@pytest.mark.gpu
def test_cpp_sharded_filter_gathers_through_the_library(tmp_path):
    """include/fbus/sharded_filter.hpp (what INTEGRATION.md's 8-GPU driver is written against), compiled with plain g++ and run
    as ONE rank on this box: the shard arithmetic, the communicator from a unique id, a predict on the shard and the gather --
    the gathered bytes must be this rank's records (hipMemcpy'd back through the C ABI's device pointer)."""
    libdir = os.path.join(ROOT, "fbus-ekf_amd", "lib")
    src = tmp_path / "sharded.cpp"
    src.write_text(r'''
#include <fbus/sharded_filter.hpp>
#include <hip/hip_runtime_api.h>
#include <cstdio>
#include <cstring>
#include <vector>
int main() {
    using SF = fbus::ShardedFilter<float>;
    long lo, hi, covered = 0;
    for (int r = 0; r < 8; ++r) { SF::shard_range(262144, r, 8, lo, hi); if (hi - lo != 32768 || lo % 64) return 2; covered += hi - lo; }
    if (covered != 262144) return 3;
    SF::shard_range(200, 2, 3, lo, hi); if (lo != 128 || hi != 200) return 4;
    const long total = 1000;
    SF f(total, 0, 1, SF::unique_id(), fbus::BatchedFilter<float>::defaults(FBUS_DIALECT_MATLAB), 0);
    if (f.lo() != 0 || f.hi() != total || f.gathered_bytes() != 1024u * 800u || f.offset_of(0) != 0) return 5;
    f.filter().reset_covariance();
    std::vector<float> a(total * 3, 0.1f), w(total * 3, 0.01f);
    f.filter().predict(a.data(), w.data(), 0.005f);
    void *out = nullptr, *recs = nullptr; size_t tot = 0;
    if (hipMalloc(&out, f.gathered_bytes()) != hipSuccess) return 6;
    f.gather(out); f.filter().sync();
    fbus_ekf_records(f.filter().handle(), &recs, nullptr, &tot);
    std::vector<char> g(tot), mine(tot);
    hipMemcpy(g.data(), out, tot, hipMemcpyDeviceToHost); hipMemcpy(mine.data(), recs, tot, hipMemcpyDeviceToHost);
    std::printf("gathered %zu bytes equal %d\n", tot, int(std::memcmp(g.data(), mine.data(), tot) == 0));
    return std::memcmp(g.data(), mine.data(), tot) == 0 ? 0 : 7;
}
''')
    exe = tmp_path / "sharded"
    subprocess.run(["g++", "-std=c++14", "-O1", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                    str(src), "-o", str(exe), "-L", libdir, "-lfbus_ekf", "-L", "/opt/rocm/lib", "-lamdhip64",
                    f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "gathered 819200 bytes equal 1" in r.stdout


@pytest.mark.gpu
def test_gather_with_a_callers_own_communicator():
    """fbus_ekf_comm_attach: the caller creates the ncclComm_t itself (here through ctypes on the same librccl the library binds),
    the library only uses it -- and does not destroy it"""
    import ctypes as C
    import torch
    sys.path.insert(0, os.path.join(ROOT, "fbus-ekf_amd"))
    from fbus_ekf import BatchedFilter, capi, synth
    rccl = C.CDLL("librccl.so.1")

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    torch.cuda.set_device(0)
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    B = 640
    prm = capi.default_params(0)
    nom, rot, P, prev = synth.initial_state(0, B, list(prm.p0_diag), 18)
    with BatchedFilter(B, prm) as flt:
        flt.set_state(nom, rot, P, prev)
        _, bpf, total = flt.records()
        rec = torch.empty(total, dtype=torch.uint8, device="cuda:0")
        flt.attach_records(rec)
        lib = capi.load_library()
        assert lib.fbus_ekf_comm_attach(flt._h, comm, 0, 1) == 0
        out = torch.zeros(total, dtype=torch.uint8, device="cuda:0")
        flt.gather(out); flt.sync()
        assert torch.equal(out, rec)
        assert lib.fbus_ekf_comm_destroy(flt._h) == 0          # detaches; the communicator is still the caller's
        with pytest.raises(capi.FbusError):
            flt.gather(out)
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    assert rccl.ncclCommDestroy(comm) == 0

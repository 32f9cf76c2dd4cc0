"""CPU suite, part 3: host logic -- seeded streams are shard-invariant, shard ranges
tile the batch, and the N > 1 path (shard -> step -> gather) gives the single-process
result bit for bit (world_size-2 gloo, oracle standing in for the device engine)."""
import os
import socket

import numpy as np
import pytest

from fbus_ekf import capi, shard, synth


def test_shard_ranges_tile_the_batch():
    for total in (1, 63, 64, 65, 4096, 65536, 262144, 100001):
        for world in (1, 2, 3, 8):
            r = [shard.shard_range(total, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            for (a0, a1), (b0, b1) in zip(r, r[1:]):
                assert a1 == b0 and a0 <= a1
            assert all(lo % 64 == 0 for lo, _ in r)


def test_streams_are_shard_invariant():
    prm = capi.default_params(0)
    full = synth.initial_state(0, 3000, list(prm.p0_diag), 18, mixed_cov=True)
    for lo, hi in ((0, 1024), (1000, 2100), (2047, 3000)):
        part = synth.initial_state(lo, hi, list(prm.p0_diag), 18, mixed_cov=True)
        for f, p in zip(full, part):
            assert np.array_equal(f[lo:hi], p)
    acc, gyr = synth.imu_samples(0, 3000, 5, 3, full[0])
    a2, g2 = synth.imu_samples(1500, 2500, 5, 3, full[0][1500:2500])
    assert np.array_equal(acc[:, 1500:2500], a2) and np.array_equal(gyr[:, 1500:2500], g2)
    ids, pos, quat = synth.marker_frame(0, 3000, 2, 4, full[0], prm)
    i2, p2, q2 = synth.marker_frame(700, 1300, 2, 4, full[0][700:1300], prm)
    assert np.array_equal(ids[700:1300], i2) and np.array_equal(pos[700:1300], p2) and np.array_equal(quat[700:1300], q2)
    assert len({tuple(sorted(r)) for r in ids[:50].tolist()}) > 5           # markers vary across filters
    assert all(len(set(r)) == 4 for r in ids[:200].tolist())                # drawn without replacement
    assert np.abs(np.linalg.norm(quat, axis=-1) - 1).max() < 1e-12


def test_synthetic_measurement_is_consistent_with_the_state():
    """a correct() with the synthetic markers barely moves a filter sitting at x0 (h(x0) + 1e-3 noise)."""
    from replay_ref import OracleEngine
    prm = capi.default_params(0)
    nom, rot, P, prev = synth.initial_state(0, 32, list(prm.p0_diag), 18)
    ids, pos, quat = synth.marker_frame(0, 32, 0, 4, nom, prm)
    eng = OracleEngine(32, 0, 18)
    eng.set_state(nom, rot, P, prev)
    ok = eng.correct(ids, pos, quat, 1)
    assert ok.all()
    assert np.abs(eng.nominal[:, 0:3] - nom[:, 0:3]).max() < 2e-2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    import torch
    import torch.distributed as dist
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for sub in ("../fbus-ekf_amd", "../oracle", "."):
        sys.path.insert(0, os.path.join(here, sub))
    from replay_ref import OracleEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard.shard_range(total, rank, world)
    out = _run_shard(OracleEngine, lo, hi)
    local = torch.from_numpy(np.concatenate([out[0].ravel(), out[2].ravel()]))
    gathered = shard.gather_records(local, dist, world)
    worst = shard.max_over_ranks(float(rank + 1), dist, world)
    if rank == 0:
        q.put(([g.numpy() for g in gathered], worst))
    dist.barrier()
    dist.destroy_process_group()


def _run_shard(engine_cls, lo, hi):
    prm = capi.default_params(0)
    nom, rot, P, prev = synth.initial_state(lo, hi, list(prm.p0_diag), 18)
    eng = engine_cls(hi - lo, 0, 18)
    eng.set_state(nom, rot, P, prev)
    step = 0
    for frame in range(2):
        acc, gyr = synth.imu_samples(lo, hi, step, 3, nom)
        for k in range(3):
            eng.predict(acc[k], gyr[k], np.array([0.005]))
        step += 3
        ids, pos, quat = synth.marker_frame(lo, hi, frame, 4, nom, prm)
        eng.correct(ids, pos, quat, 1)
    return eng.get_state()


def test_two_rank_gloo_run_equals_single_process_bitwise():
    import torch.multiprocessing as mp
    total, world = 256, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    gathered, worst = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert worst == 2.0
    from replay_ref import OracleEngine
    ref = _run_shard(OracleEngine, 0, total)
    off = 0
    for r in range(world):
        lo, hi = shard.shard_range(total, r, world)
        n = hi - lo
        nom = gathered[r][:n * 19].reshape(n, 19)
        P = gathered[r][n * 19:].reshape(n, 18, 18)
        assert np.array_equal(nom, ref[0][lo:hi]) and np.array_equal(P, ref[2][lo:hi])
        off += n
    assert off == total


def test_recording_files_round_trip_and_fusion_trace_format(tmp_path):
    """file formats either side of the path: imu.txt / image.txt in, fusion.txt
    (`t p q(wxyz) v ba bg`, 17 columns, filter.cpp:238-248) out"""
    import os
    from fbus_ekf import replay
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "land_slice.npz"))
    np.savetxt(tmp_path / "imu.txt", g["imu"], fmt="%.9f")
    np.savetxt(tmp_path / "image.txt", g["image"], fmt="%.9f")
    imu, image = replay.load_recording(str(tmp_path))
    assert imu.shape == g["imu"].shape and image.shape == g["image"].shape
    assert np.abs(imu - g["imu"]).max() < 1e-8 and np.abs(image - g["image"]).max() < 1e-8
    st = g["states_matlab"]
    rows = replay.fusion_rows(st)
    assert rows.shape == (len(st), 17)
    nom = st[:, 1:20]
    assert np.array_equal(rows[:, 0], st[:, 0]) and np.array_equal(rows[:, 1:4], nom[:, 0:3])
    assert np.array_equal(rows[:, 4:8], nom[:, 6:10]) and np.abs(np.linalg.norm(rows[:, 4:8], axis=1) - 1).max() < 1e-6
    assert np.array_equal(rows[:, 8:11], nom[:, 3:6]) and np.array_equal(rows[:, 14:17], nom[:, 13:16])
    replay.save_fusion(tmp_path / "fusion.txt", st)
    back = np.loadtxt(tmp_path / "fusion.txt")
    assert back.shape == rows.shape and np.abs(back - rows).max() < 1e-8
    with pytest.raises(ValueError):
        np.savetxt(tmp_path / "image.txt", g["image"][:, :8], fmt="%.9f")
        replay.load_recording(str(tmp_path))


def test_cpp_shard_arithmetic_matches_python(tmp_path):
    """include/fbus/sharded_filter.hpp::shard_range (what a C++ multi-GPU driver cuts the batch with) against fbus_ekf.shard.shard_range
    (what bench.py uses) over many (total, world): the same contiguous 64-aligned ranges, and record_bytes_of_ranks = whole tiles."""
    import subprocess
    from fbus_ekf import shard
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    src = tmp_path / "sr.cpp"
    src.write_text(r'''
#include <fbus/sharded_filter.hpp>
#include <cstdio>
int main() {
    const long totals[] = {1, 63, 64, 65, 200, 1000, 4096, 65536, 262144, 262145, 1048576};
    for (long t : totals) for (int w = 1; w <= 8; ++w) for (int r = 0; r < w; ++r) {
        long lo, hi; fbus::ShardedFilter<float>::shard_range(t, r, w, lo, hi);
        std::printf("%ld %d %d %ld %ld\n", t, w, r, lo, hi);
    }
}
''')
    exe = tmp_path / "sr"
    subprocess.run(["g++", "-std=c++14", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    rows = np.array(out, dtype=np.int64).reshape(-1, 5)
    assert len(rows) == 11 * 36
    for t, w, r, lo, hi in rows:
        assert shard.shard_range(int(t), int(r), int(w)) == (lo, hi)
    for t in (200, 1000, 262145):
        for w in (1, 3, 8):
            b = shard.record_bytes_of_ranks(t, w, 800)
            assert sum(b) >= t * 800 and all(x % (64 * 800) == 0 for x in b)
            assert sum(x // 800 for x in b) - t < 64 * w

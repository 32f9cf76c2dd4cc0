"""fbus::NodeFilter (include/fbus/node_filter.hpp): one host object that owns the filters of a whole node -- one shard, one host
thread and one handle per device.  The reference's counterpart is its one stateful FILTER object driven by one thread
(C++/src/filter.cpp:190-250).  CPU: the shard arithmetic, compiled with plain g++ against the library's C ABI (no device).  GPU: two
shards on ONE device stepped concurrently through the class, gathered by peer copies, against the single-handle run -- bit-equal,
because every shard keys its kernel choice on the whole job."""
import os
import subprocess

import numpy as np
import pytest

from fbus_ekf import capi, shard

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _compile(tmp_path, name, text):
    src = tmp_path / f"{name}.cpp"
    src.write_text(text)
    exe = tmp_path / name
    libdir = os.path.dirname(capi.library_path())
    subprocess.run(["g++", "-std=c++14", "-O1", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L", libdir, "-lfbus_ekf",
                    f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-pthread"], check=True)
    return exe


def test_node_filter_shard_arithmetic_matches_the_python_and_sharded_filter_rules(tmp_path):
    exe = _compile(tmp_path, "ranges", r'''
#include <fbus/node_filter.hpp>
#include <fbus/sharded_filter.hpp>
#include <cstdio>
#include <cstdlib>
int main(int argc, char** argv) {
    const long total = std::atol(argv[1]); const int n = std::atoi(argv[2]);
    for (int k = 0; k < n; ++k) {
        long lo, hi, lo2, hi2;
        fbus::NodeFilter<float>::shard_range(total, k, n, lo, hi);
        fbus::ShardedFilter<float>::shard_range(total, k, n, lo2, hi2);
        if (lo != lo2 || hi != hi2) return 2;
        std::printf("%ld %ld\n", lo, hi);
    }
    return 0;
}
''')
    for total, n in ((262144, 8), (262144 + 100, 8), (200, 3), (65, 2), (64 * 7, 7), (1000003, 5)):
        r = subprocess.run([str(exe), str(total), str(n)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        rows = [tuple(int(x) for x in l.split()) for l in r.stdout.split("\n") if l]
        assert rows == [shard.shard_range(total, k, n) for k in range(n)]
        assert rows[0][0] == 0 and rows[-1][1] == total and all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
        assert all(lo % 64 == 0 for lo, _ in rows)


@pytest.mark.gpu
def test_node_filter_two_shards_on_one_device_equal_the_single_handle(tmp_path):
    from fbus_ekf import BatchedFilter, synth
    total, M, K = 24576 + 64, 4, 3
    prm = capi.default_params(0)
    r32 = lambda a: np.asarray(a, np.float64).astype(np.float32)
    nom, rot, _, prev = synth.initial_state(0, total, list(prm.p0_diag), 18, with_cov=False)
    acc, gyr = synth.imu_samples(0, total, 0, K, nom)
    ids, pos, quat = synth.marker_frame(0, total, 0, M, nom, prm)
    arrays = {"nom": r32(nom), "rot": r32(rot), "prev": prev.astype(np.int32), "acc": r32(acc), "gyr": r32(gyr), "ids": ids.astype(np.int32),
              "pos": r32(pos), "quat": r32(quat)}
    for k, v in arrays.items():
        v.tofile(tmp_path / f"{k}.bin")
    exe = _compile(tmp_path, "node", r'''
#include <fbus/node_filter.hpp>
#include <cstdio>
#include <string>
template <typename T> std::vector<T> rd(const std::string& p, size_t n) {
    std::vector<T> v(n); FILE* f = std::fopen(p.c_str(), "rb"); if (!f || std::fread(v.data(), sizeof(T), n, f) != n) std::abort();
    std::fclose(f); return v; }
int main(int argc, char** argv) {
    const std::string d = argv[1];
    const long total = std::atol(argv[2]); const int M = 4, K = 3;
    const fbus_params prm = fbus::BatchedFilter<float>::defaults(FBUS_DIALECT_MATLAB);
    auto nom = rd<float>(d + "/nom.bin", total * 19), rot = rd<float>(d + "/rot.bin", total * 9);
    auto prev = rd<int32_t>(d + "/prev.bin", total);
    auto acc = rd<float>(d + "/acc.bin", K * total * 3), gyr = rd<float>(d + "/gyr.bin", K * total * 3);
    auto ids = rd<int32_t>(d + "/ids.bin", total * M); auto pos = rd<float>(d + "/pos.bin", total * M * 3), quat = rd<float>(d + "/quat.bin", total * M * 4);
    fbus::NodeFilter<float> node(total, { 0, 0 }, prm);                 // two shards, both on device 0 (a one-GPU box)
    if (node.shards() != 2 || node.lo(1) != node.hi(0) || node.hi(1) != total) return 3;
    node.for_each_shard([&](int k, fbus::BatchedFilter<float>& f) {
        const long lo = node.lo(k);
        f.set_state(&nom[lo * 19], &rot[lo * 9], nullptr, &prev[lo]);
        f.reset_covariance();
        for (int s = 0; s < K; ++s) f.predict(&acc[(s * total + lo) * 3], &gyr[(s * total + lo) * 3], 0.005f);
        f.correct(M, &ids[lo * M], &pos[lo * M * 3], &quat[lo * M * 4], fbus::BatchedFilter<float>::Mode::Stacked);
        f.predict(&acc[lo * 3], &gyr[lo * 3], 0.005f);
    });
    // the gather: peer copies into the record buffer of a handle that holds the whole batch (its state is then read back unpacked)
    fbus::BatchedFilter<float> all(int(total), prm, 0);
    void* buf = nullptr; size_t bytes = 0;
    fbus_ekf_records(all.handle(), &buf, nullptr, &bytes);
    if (bytes != node.gathered_bytes() || node.offset_of(1) != size_t(node.hi(0)) * 800) return 4;
    node.gather_to(0, buf);
    node.sync();
    std::vector<float> n2(total * 19), r2(total * 9), P2(total * 324); std::vector<int32_t> p2(total);
    all.get_state(n2.data(), r2.data(), P2.data(), p2.data());
    FILE* f = std::fopen((d + "/out.bin").c_str(), "wb");
    std::fwrite(n2.data(), 4, n2.size(), f); std::fwrite(P2.data(), 4, P2.size(), f); std::fclose(f);
    return 0;
}
''')
    r = subprocess.run([str(exe), str(tmp_path), str(total)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    out = np.fromfile(tmp_path / "out.bin", np.float32)
    n_nom, n_P = out[:total * 19].reshape(total, 19), out[total * 19:].reshape(total, 18, 18)
    with BatchedFilter(total, prm) as flt:
        flt.set_state(arrays["nom"], arrays["rot"], None, arrays["prev"])
        flt.reset_cov()
        dt = np.array([np.float64(np.float32(0.005))])
        for s in range(K):
            flt.predict(arrays["acc"][s], arrays["gyr"][s], dt)
        flt.correct(arrays["ids"], arrays["pos"], arrays["quat"], capi.MODE_STACKED)
        flt.predict(arrays["acc"][0], arrays["gyr"][0], dt)
        one = flt.get_state()
    assert np.isfinite(n_nom).all() and np.array_equal(n_nom, one[0]) and np.array_equal(n_P, one[2])

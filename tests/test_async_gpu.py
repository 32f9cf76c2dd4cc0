"""GPU suite: the host-pointer entry points without the wait (round 6: fbus_ekf_predict_async / _predict_n_async / _correct_async /
_correct_pixels_async).  The reference's caller hands one IMU sample at a time to the filter and returns at once (FILTER::SetImuData,
C++/src/filter.cpp:24-55; BatchImuProcessing runs one predict per sample, :505-516; SetDetectionResult copies the detections, :61-65).
Asserted: the asynchronous sequence == the synchronous host-pointer sequence BIT FOR BIT (same kernels, same inputs); arrays are taken
by value (the caller overwrites its pageable buffer right behind the call); more calls in flight than the ring has slots; pinned memory
is transferred in place; argument checks are those of the synchronous calls; not capturable into a graph."""
import numpy as np
import pytest

from fbus_ekf import BatchedFilter, capi, synth

pytestmark = pytest.mark.gpu
DT = np.array([0.005], np.float32)
f32 = lambda a: np.ascontiguousarray(a, np.float32)


def _inputs(B, K, M, dialect=0, seed=0):
    prm = capi.default_params(dialect)
    nom, rot, P, prev = synth.initial_state(seed, seed + B, list(prm.p0_diag), 18, mixed_cov=True)
    acc, gyr = synth.imu_samples(seed, seed + B, 0, K, nom)
    frames = [synth.marker_frame(seed, seed + B, f, M, nom, prm) for f in range(3)]
    return prm, f32(nom), f32(rot), f32(P), prev, f32(acc), f32(gyr), frames


@pytest.mark.parametrize("B", [300, 4096])
@pytest.mark.parametrize("dialect", [0, 1])
def test_async_sequence_equals_the_synchronous_one_bit_for_bit(B, dialect):
    K, M = 20, 4
    prm, nom, rot, P, prev, acc, gyr, frames = _inputs(B, K, M, dialect)
    with BatchedFilter(B, prm) as a, BatchedFilter(B, prm) as s:
        for f in (a, s):
            f.set_state(nom, rot, P, prev)
        k = 0
        for fr, Kf in enumerate((7, 7, 6)):
            ids, pos, quat = frames[fr]
            for j in range(Kf):
                s.predict(acc[k + j], gyr[k + j], DT)
                buf_a, buf_g = acc[k + j].copy(), gyr[k + j].copy()
                a.predict_async(buf_a, buf_g, DT)
                buf_a[...] = np.nan; buf_g[...] = np.nan             # by value: the caller's (pageable) buffers are free on return
            s.correct(ids, f32(pos), f32(quat), capi.MODE_STACKED)
            bp, bq, bi = f32(pos).copy(), f32(quat).copy(), ids.copy()
            a.correct_async(bi, bp, bq, capi.MODE_STACKED)
            bp[...] = np.nan; bq[...] = np.nan; bi[...] = -1
            k += Kf
        # K samples in one asynchronous call, too
        s.predict_n(acc[:5], gyr[:5], np.full(5, DT[0], np.float32))
        a.predict_async(acc[:5], gyr[:5], np.full(5, DT[0], np.float32), K=5)
        st = a.async_stats()
        assert st["calls"] == 24 and st["direct_pieces"] == 0        # 24 calls through an 8-slot ring
        ga, gs = a.get_state(), s.get_state()                         # (get_state completes everything queued)
        assert np.array_equal(a.applied(), s.applied())
    for name, x, y in zip(("nominal", "rot", "P", "prev"), ga, gs):
        assert np.isfinite(np.asarray(x, np.float64)).all() and np.array_equal(x, y), name


def test_pinned_arrays_are_transferred_in_place():
    import torch
    B, K, M = 8192, 12, 4
    prm, nom, rot, P, prev, acc, gyr, frames = _inputs(B, K, M)
    ids, pos, quat = frames[0]
    pin = lambda x: torch.from_numpy(np.ascontiguousarray(x)).pin_memory()
    p_acc, p_gyr = pin(acc), pin(gyr)
    p_left = None
    with BatchedFilter(B, prm) as a, BatchedFilter(B, prm) as s:
        for f in (a, s):
            f.set_state(nom, rot, P, prev)
        for k in range(K):
            s.predict(acc[k], gyr[k], DT)
            a.predict_async(p_acc[k], p_gyr[k], DT)
        s.correct(ids, f32(pos), f32(quat), capi.MODE_STACKED)
        a.correct_async(pin(ids), pin(f32(pos)), pin(f32(quat)), capi.MODE_STACKED)
        a.async_inputs_consumed()                                      # from here on the pinned arrays may be rewritten
        p_acc.zero_(); p_gyr.zero_()
        st = a.async_stats()
        assert st["calls"] == K + 1 and st["direct_pieces"] == 2 * K + 3, st          # accel, gyro per predict (dt is 4 bytes: staged); ids, pos, quat
        ga, gs = a.get_state(), s.get_state()
    for x, y in zip(ga, gs):
        assert np.array_equal(x, y)


def test_async_pixel_update_and_fp64_handle():
    from util import pixel_scene
    B, M, size = 256, 3, 0.28
    prm = capi.default_params(0)
    prm.marker_size = size
    nom0, _, P, prev = synth.initial_state(0, B, list(prm.p0_diag), 18, mixed_cov=True)
    nom, rot, ids, left, right = pixel_scene(B, M, prm, size, seed=3, noise=5e-4, nominal=nom0)
    acc, gyr = synth.imu_samples(0, B, 0, 2, nom)
    for dtype, npdt in ((32, np.float32), (64, np.float64)):
        c = lambda x: np.ascontiguousarray(x, npdt)
        with BatchedFilter(B, prm, dtype=dtype) as a, BatchedFilter(B, prm, dtype=dtype) as s:
            for f in (a, s):
                f.set_state(c(nom), c(rot), c(P), prev)
            dt = np.array([0.005], npdt)
            s.predict(c(acc[0]), c(gyr[0]), dt); s.correct_pixels(ids, c(left), c(right))
            a.predict_async(c(acc[0]), c(gyr[0]), dt); a.correct_pixels_async(ids, c(left), c(right))
            assert np.array_equal(a.applied(), s.applied()) and s.applied().any()
            for x, y in zip(a.get_state(), s.get_state()):
                assert np.array_equal(x, y)


def test_async_calls_check_their_arguments_and_refuse_graph_capture():
    import ctypes as C
    B = 128
    prm, nom, rot, P, prev, acc, gyr, frames = _inputs(B, 2, 2)
    ids, pos, quat = frames[0]
    pos, quat = f32(pos), f32(quat)
    with BatchedFilter(B, prm) as flt:
        flt.set_state(nom, rot, P, prev)
        before = flt.get_state()
        lib, h = flt._lib, flt._h
        p = lambda x: x.ctypes.data_as(C.c_void_p)
        assert lib.fbus_ekf_predict_async(h, p(acc[0]), None, p(DT), 0) == 1
        assert lib.fbus_ekf_predict_n_async(h, 0, p(acc[0]), p(gyr[0]), p(DT), 0) == 1
        assert lib.fbus_ekf_correct_async(h, 2, p(ids), p(pos), p(quat), 7, None) == 4          # unsupported mode
        assert lib.fbus_ekf_correct_async(h, 0, p(ids), p(pos), p(quat), 1, None) == 1
        assert lib.fbus_ekf_correct_pixels_async(h, 2, p(ids), None, None, None) == 1
        flt.sync()
        for x, y in zip(flt.get_state(), before):
            assert np.array_equal(x, y)
        flt._check(lib.fbus_ekf_graph_begin(h), "graph_begin")
        assert lib.fbus_ekf_predict_async(h, p(acc[0]), p(gyr[0]), p(DT), 0) == 1
        gid = C.c_int(-1)
        lib.fbus_ekf_graph_end(h, C.byref(gid))
        flt.sync()
        for x, y in zip(flt.get_state(), before):
            assert np.array_equal(x, y)
        assert flt.async_stats()["calls"] == 0

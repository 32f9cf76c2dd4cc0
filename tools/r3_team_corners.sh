# correct from stereo corners (refractive triangulation on the device), stacked mode: the one-wave kernel (FBUS_TEAM_CORRECT=1) against
# correct_corners_team_kernel with two (=2) and four (=4) roles; HIP-event bracket per launch
mkdir -p gpurun_out/r03
python -m pytest tests/test_vision_gpu.py tests/test_configs_gpu.py tests/test_pixels_gpu.py -q -k "corners or pixels" 2>&1 | tail -2
python - > gpurun_out/r03/team_corners.txt <<'PY'
import os, sys, subprocess
code = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, 'fbus-ekf_amd'); sys.path.insert(0, 'tools')
from fbus_ekf import BatchedFilter, capi, synth
dev = torch.device('cuda:0'); prm = capi.default_params(0)
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(torch.float32)
B, M, what = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
nom, rot, P, prev = synth.initial_state(0, B, list(prm.p0_diag), 18)
ids, _, _ = synth.marker_frame(0, B, 0, min(M, 12), nom, prm)
if M > 12: ids = np.concatenate([ids, np.full((B, M - 12), -1, np.int32)], axis=1)
c = np.load('tests/golden/vision_water.npz')['corners']
base = c[np.random.default_rng(1).integers(0, len(c), B * M)]
left, right = base[:, 2:10].reshape(B, M, 8), base[:, 10:18].reshape(B, M, 8)
d = (torch.from_numpy(ids).to(dev), up(left), up(right))
if what == 'pixels':
    prm = capi.default_params(0)
    nom2, rot2, pids, pleft = synth.pixel_wall_scene(B, M, prm, 0.28)
    nom, rot = nom2, rot2
    dp = (torch.from_numpy(pids).to(dev), up(pleft))
with BatchedFilter(B, prm, order_streams=False) as flt:
    flt.set_state(nom, rot, P, prev)
    if what == 'pixels':
        fn = lambda: flt.correct_pixels(dp[0], dp[1], None)
    else:
        fn = lambda: flt.correct_corners(d[0], d[1], d[2], capi.VIS_REFRACTIVE, 1)
    for _ in range(5): fn()
    flt.sync(); flt.timing_enable(True); flt.timing_reset()
    for _ in range(100): fn()
    ms, n = flt.timing_read(capi.KERNEL_CORRECT_CORNERS)
    print('%.2f' % (ms / n * 1e3))
"""
print("filters  markers   one-wave    2 roles    4 roles   (us per launch, HIP-event bracket)")
for what in ("corners", "pixels"):
    print(what + (" (stereo pixels -> refractive triangulation -> 12 rows per marker)" if what == "corners" else " (left camera: 8 reprojection rows per marker)"))
    for B, M in ((4096, 4), (16384, 4), (16384, 16), (32768, 4), (32768, 16), (49152, 4)):
        row = []
        for t in (1, 2, 4):
            env = dict(os.environ, FBUS_TEAM_CORRECT=str(t))
            r = subprocess.run([sys.executable, '-c', code, str(B), str(M), what], env=env, capture_output=True, text=True)
            row.append((r.stdout.strip().splitlines() or ['fail: ' + r.stderr.strip()[-200:]])[-1])
        print(f"{B:7d}  {M:7d}   " + "   ".join(f"{x:>8s}" for x in row), flush=True)
PY
cat gpurun_out/r03/team_corners.txt

import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("fbus-ekf_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, sub))
from fbus_ekf import BatchedFilter, capi, synth
from replay_ref import OracleEngine
from util import parity_errors
from test_pixels_gpu import _scene, SIZE
B, M = 320, 4
prm, nom, rot, P, prev, ids, left, right = _scene(B, M, 0, seed=11)
eng = OracleEngine(B, 0, 18)
eng.set_state(nom, rot, P, prev)
eng.orc.correct_pixels(eng.nominal, eng.rot, eng.P, eng.prev, ids, left, None, SIZE, prm.r_pix)
ref = eng.get_state()
for roles in (1, 0):
    with BatchedFilter(B, prm, dtype=64) as flt:
        flt.set_team(0, roles)
        flt.set_state(nom, rot, P, prev)
        flt.correct_pixels(ids, left, None)
        got = flt.get_state()
    e = parity_errors(got, ref)
    print("roles", roles, {k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in e.items() if k != "plain_table"})
    d = np.abs(got[0] - ref[0])
    print("  worst filter", int(d.max(1).argmax()), "per-column max", np.array2string(d.max(0), precision=1))
    dP = np.abs(got[2] - ref[2]) / np.sqrt(np.einsum("bii->bi", ref[2])[:, :, None] * np.einsum("bii->bi", ref[2])[:, None, :])
    print("  cov block worst", np.unravel_index(dP.argmax(), dP.shape), f"{dP.max():.2e}")

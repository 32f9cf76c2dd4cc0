import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for s in ("fbus-ekf_amd", "oracle", "tests"): sys.path.insert(0, os.path.join(ROOT, s))
from fbus_ekf import BatchedFilter, capi, synth
from replay_ref import OracleEngine
from util import parity_errors
r32 = lambda a: np.asarray(a, np.float64).astype(np.float32).astype(np.float64)
B, M, dialect, mode = 16384, 4, 0, 1
prm = capi.default_params(dialect)
nom, rot, P, prev = synth.initial_state(0, B, list(prm.p0_diag), 18, mixed_cov=True)
nom, rot, P = r32(nom), r32(rot), r32(P)
ids, pos, quat = synth.marker_frame(0, B, 0, M, nom, prm)
rng = np.random.default_rng(31)
pos = r32(pos + rng.normal(0, 0.01, pos.shape)); quat = r32(quat)
prev = rng.choice([0, 1, 2, 16], B).astype(np.int32)
with BatchedFilter(B, prm) as flt:
    flt.set_state(nom, rot, P, prev)
    flt.correct(ids, pos, quat, mode)
    g = flt.get_state()
ev = np.linalg.eigvalsh(g[2].astype(np.float64))
print("lib", capi.library_path(), "min eig", ev.min(), "count<=0", (ev.min(axis=1) <= 0).sum(), "finite", np.isfinite(g[2]).all())
sub = np.arange(0, B, 61)
eng = OracleEngine(len(sub), dialect, 18); eng.set_state(nom[sub], rot[sub], P[sub], prev[sub]); eng.correct(ids[sub], pos[sub], quat[sub], mode)
e = parity_errors([x[sub] for x in g], eng.get_state()); print({k: v for k, v in e.items() if k != "plain_table"})
evo = np.linalg.eigvalsh(eng.P); print("oracle min eig", evo.min())
np.save(os.environ.get("OUTNPY", "/tmp/P.npy"), g[2])

#!/usr/bin/env python3
"""Experiment: does a wave64 VALU instruction with EXEC = low 32 lanes only cost half the issue time on gfx950?
Times correct (reference mode: the nearest of M=4 markers applied row by row; skipped lanes leave that kernel at once --
the stacked kernel lets them run along, so it cannot show this) with no skip mask, with lanes 32..63 of every wave
skipped, and with odd lanes skipped.  Round-1 result on the row-by-row stacked kernel of the time: same time in all
three cases, i.e. a half-empty wave buys nothing."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fbus-ekf_amd"))
import torch
from fbus_ekf import BatchedFilter, capi, synth
B, M = 65536, 4
prm = capi.default_params(0)
nom, rot, P, prev = synth.initial_state(0, B, list(prm.p0_diag), 18)
ids, pos, quat = synth.marker_frame(0, B, 0, M, nom, prm)
dev = torch.device("cuda:0")
f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
d_ids, d_pos, d_quat = torch.from_numpy(ids).to(dev), f32(pos), f32(quat)
lane = np.arange(B) % 64
masks = {"none": None, "upper half skipped": (lane >= 32), "odd lanes skipped": (lane % 2 == 1)}
with BatchedFilter(B, prm) as flt:
    flt.set_stream(torch.cuda.current_stream())
    for name, m in masks.items():
        sk = None if m is None else torch.from_numpy(m.astype(np.uint8)).to(dev)
        flt.set_state(nom, rot, P, prev)
        for _ in range(3):
            flt.correct(d_ids, d_pos, d_quat, 0, sk)
        flt.sync()
        flt.timing_enable(True); flt.timing_reset()
        for _ in range(20):
            flt.correct(d_ids, d_pos, d_quat, 0, sk)
        ms, n = flt.timing_read(capi.KERNEL_CORRECT)
        flt.timing_enable(False)
        print(f"{name:22s}: {ms / n * 1e3:7.1f} us per correct launch   applied {int(flt.applied().sum())}")

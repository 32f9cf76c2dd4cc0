#!/usr/bin/env python3
"""set_state_dev / get_state_dev (API arrays <-> packed records on the device) by batch size: time per call and bytes moved."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fbus-ekf_amd"))
import torch
from fbus_ekf import BatchedFilter, capi
dev = torch.device("cuda:0")
prm = capi.default_params(0)
for B in [int(a) for a in sys.argv[1:]] or [65536, 262144, 65536 - 3]:
    nom = torch.zeros(B, 19, device=dev); nom[:, 6] = 1
    rot = torch.eye(3, device=dev).reshape(1, 9).repeat(B, 1).contiguous()
    P = (torch.eye(18, device=dev) * 0.01).reshape(1, 18, 18).repeat(B, 1, 1).contiguous()
    prev = torch.zeros(B, dtype=torch.int32, device=dev)
    with BatchedFilter(B, prm) as flt:
        o = [torch.empty_like(nom), torch.empty_like(rot), torch.empty_like(P), torch.empty_like(prev)]
        for name, fn in (("set_state_dev", lambda: flt.set_state(nom, rot, P, prev)), ("get_state_dev", lambda: flt._lib.fbus_ekf_get_state_dev(flt._h, *[flt._p(x) for x in o])),
                         ("set_state_dev (nominal only)", lambda: flt.set_state(nom, None, None, None)), ("reset_cov", flt.reset_cov)):
            for _ in range(3): fn()
            flt.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): fn()
            flt.sync(); torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / 20 * 1e6
            print(f"B {B:>7} {name:<30} {us:8.1f} us per call")

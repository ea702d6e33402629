#!/usr/bin/env python3
"""Calibration run for the HBM counters: reads then writes a 1 GiB buffer (far beyond L2 + Infinity Cache) with the
library's access shape (one dword per lane, coalesced).  Run under
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d OUT -- python3 tools/pmc_calibrate.py
and again with WRITE_SIZE; compare the counter (KiB) with the known byte count."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from bez_isaacgym_amd.sim import load_library
lib = load_library()
n = 256 * 1024 * 1024
buf = torch.zeros(n, dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
for _ in range(3):
    lib.bez_sim_calibrate(C.c_void_p(buf.data_ptr()), n, 0, None)
for _ in range(3):
    lib.bez_sim_calibrate(C.c_void_p(buf.data_ptr()), n, 1, None)
torch.cuda.synchronize()
print("bytes per launch:", n * 4)

"""Diagnostic (GPU box): the native transport in a process WITHOUT torch, with RCCL's own log on."""
import os, sys
os.environ.setdefault("NCCL_DEBUG", "INFO")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import marlin_amd as M
from marlin_amd import dist as MD, _lib
lib = _lib.load()
M.init(0)
try:
    print("enable:", MD.enable_native_rccl(None))
    print(MD.native_rccl_info())
    print("allgather:", MD.selftest_allgather(None))
except Exception as e:
    print("FAILED:", e)

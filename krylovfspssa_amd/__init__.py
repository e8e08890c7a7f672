"""krylovfspssa_amd: MI355X-native exp(tA)v hot path of the Krylov-FSP CME solver.

csrc/      HIP kernels (gfx950) + the C ABI of include/kfsp.h -> lib/libkfsp_hip.so
host.py    ctypes mirror of the C ABI (tests, bench, Python drivers)
synth.py   synthetic box-structured CME generators (benchmark / test inputs)
fortran/   Fortran host modules keeping the reference's entry points
"""
from .host import KfspContext, KfspError, load_library, padm  # noqa: F401

"""Achievable HBM read rate of the k_stream_read self-test at 4/8/16 bytes per
lane (2 GiB buffer, far beyond the Infinity Cache).  usage: python profiles/stream_widths.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from krylovfspssa_amd import KfspContext  # noqa: E402

with KfspContext(0) as c:
    for w in (4, 8, 16):
        c.selftest_stream(2 << 30, w, 2)
        ms = min(c.selftest_stream(2 << 30, w, 10) for _ in range(3)) / 10
        print(f"{w:2d} B/lane: {ms * 1e3:7.1f} us per 2 GiB  = {(2 << 30) / ms / 1e6:7.1f} GB/s", flush=True)

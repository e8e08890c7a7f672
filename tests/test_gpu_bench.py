"""bench.py on the device through a one-rank RCCL communicator (--force-comm): the JSON contract, and
what happens when the distributed product RAISES on some rank (here: injected) - the communicator is
thrown away and made anew, the next simpler exchange is tried, and only if every mode raises does the
job stop with exit code 3 instead of printing a number."""
import json
import os
import subprocess
import sys

import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(extra_env, *flags):
    env = dict(os.environ, **extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tiny", "--force-comm",
                           "--no-cpu", "--no-expv", "--steps", "5", "--warmup", "2", *flags],
                          cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)


def test_bench_line_through_a_one_rank_communicator():
    r = _bench({})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1                                         # exactly one JSON line on stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["steps"] == 5 and j["warmup"] == 2 and j["unit"] == "GB/s"
    assert j["self_check"]["ok"] and j["config"]["exchange"].startswith("halo strips (banded generator)")
    assert set(j["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "frac_traffic", "alg_GBps"}
    assert 0.0 < j["roofline"]["frac"] <= 1.0                      # bytes that really move / peak: never above 1


def test_a_raising_product_recreates_the_communicator_and_steps_down():
    r = _bench({"KFSP_BENCH_INJECT_RAISE": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["self_check"]["ok"] and "not overlapped" in j["config"]["exchange"]
    assert "product failed on some rank" in r.stderr
    r = _bench({"KFSP_BENCH_INJECT_RAISE": "1,2"})
    assert r.returncode == 0
    assert "all-gather of the whole vector" in json.loads(r.stdout.strip().splitlines()[-1])["config"]["exchange"]


def test_a_product_that_raises_in_every_mode_stops_the_job():
    r = _bench({"KFSP_BENCH_INJECT_RAISE": "1,2,3"})
    assert r.returncode == 3 and not r.stdout.strip()
    assert "raised in every exchange mode" in r.stderr

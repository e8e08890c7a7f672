"""bench.py as its own launcher (`python bench.py --gpus N` without torch.distributed.run): the
ranks are child processes started before anything touches the GPU, each with the environment
torch.distributed.run would give it; rank 0's JSON line is relayed; any failing rank fails the job.
(CPU only: the children here are a stand-in script that echoes its environment.)"""
import json
import os
import sys
import textwrap

from tests.conftest import ROOT

sys.path.insert(0, ROOT)


def _child(tmp_path):
    p = tmp_path / "child.py"
    p.write_text(textwrap.dedent("""
        import json, os, sys
        r = int(os.environ["RANK"])
        if os.environ.get("FAIL_RANK") == str(r):
            sys.exit(7)
        if os.environ.get("HANG_RANK") == str(r):
            import time
            time.sleep(600)            # a peer blocked in the rendezvous / a collective
        if r == 0:
            print(json.dumps({k: os.environ.get(k) for k in
                  ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY")}
                  | {"argv": sys.argv[1:]}))
        else:
            print("noise from rank", r)
    """))
    return str(p)


def test_ranks_get_the_launcher_environment_and_rank0_is_relayed(tmp_path, capfd):
    import bench
    env = {k: v for k, v in os.environ.items() if k not in ("MASTER_PORT", "MASTER_ADDR", "WORLD_SIZE", "RANK")}
    rc = bench.spawn_ranks(3, ["--gpus", "3", "--steps", "5"], script=_child(tmp_path), env=env)
    out = capfd.readouterr().out.strip().splitlines()
    assert rc == 0 and len(out) == 1                      # exactly one line on stdout: rank 0's
    d = json.loads(out[0])
    assert (d["RANK"], d["LOCAL_RANK"], d["WORLD_SIZE"], d["MASTER_ADDR"]) == ("0", "0", "3", "127.0.0.1")
    assert d["MASTER_PORT"].isdigit() and d["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert d["argv"] == ["--gpus", "3", "--steps", "5"]


def test_a_failing_rank_fails_the_job(tmp_path, capfd):
    import bench
    env = dict(os.environ, FAIL_RANK="2")
    env.pop("WORLD_SIZE", None)
    assert bench.spawn_ranks(3, [], script=_child(tmp_path), env=env) == 1


def test_a_failing_rank_ends_ranks_that_would_wait_for_it_for_ever(tmp_path, capfd):
    """one rank exits 7 while another never returns (blocked in a collective with the dead rank): the job
    ends at once with the failing rank named, the blocked one is terminated - not waited for"""
    import time
    import bench
    env = dict(os.environ, FAIL_RANK="1", HANG_RANK="2")
    env.pop("WORLD_SIZE", None)
    t0 = time.monotonic()
    assert bench.spawn_ranks(3, [], script=_child(tmp_path), env=env) == 1
    assert time.monotonic() - t0 < 30.0
    err = capfd.readouterr().err
    assert "(1, 7)" in err and "terminated" in err


def test_the_overall_deadline_ends_a_job_whose_ranks_all_hang(tmp_path, capfd):
    import time
    import bench
    env = dict(os.environ, HANG_RANK="1")
    env.pop("WORLD_SIZE", None)
    t0 = time.monotonic()
    assert bench.spawn_ranks(2, [], script=_child(tmp_path), env=env, deadline_s=2.0) == 1
    assert time.monotonic() - t0 < 30.0
    assert "deadline" in capfd.readouterr().err


def test_main_spawns_only_without_a_launcher(monkeypatch):
    """--gpus N > 1 with WORLD_SIZE unset goes to spawn_ranks before torch is imported; under a
    launcher (WORLD_SIZE set) it does not."""
    import bench
    calls = []
    monkeypatch.setattr(bench, "spawn_ranks", lambda n, argv: calls.append((n, argv)) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    assert calls == [(2, ["--gpus", "2", "--steps", "3"])]

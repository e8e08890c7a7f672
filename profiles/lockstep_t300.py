"""Run kfsp_replay on the full-horizon Goutsias script by hand and show its output (debugging aid for
tests/test_lockstep.py::test_lock_step_over_the_full_horizon_of_the_goutsias_example)."""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import lockstep as L  # noqa: E402

g = np.load(os.path.join(ROOT, "tests/golden/lockstep_digest_goutsias_input_T300.npz"))
L.write_script("/tmp/t300_script.bin", g["script"])
t0 = time.time()
r = subprocess.run([os.path.join(ROOT, "krylovfspssa_amd/fortran/_build/kfsp_replay"), "goutsias_input", "/tmp/t300_script.bin",
                    "/tmp/t300_steps.bin", "/tmp/t300_out.bin", "300.0", "safestop", "digest"],
                   cwd=os.path.join(ROOT, "tests/golden/models"), env=dict(os.environ, KFSP_CASE_CAPACITY="2097169"),
                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
print("rc", r.returncode, "seconds", round(time.time() - t0, 1))
import re
sizes = [int(x) for x in re.findall(r"FSP SIZE\s+=\s+(\d+)", r.stdout)]
tn = [float(x) for x in re.findall(r"T_NOW\s+=\s+([0-9.Ee+-]+)", r.stdout)]
ref = g["n_after"]
print("our sizes ", sizes[:60])
print("ref sizes ", [int(v) for v in ref[:60]])
print("our t_now ", tn[:50])
print("ref t_at  ", [round(float(v), 3) for v in g["t_at"][:50]])
open(os.path.join(ROOT, "gpurun_out", "t300_stdout.txt"), "w").write(r.stdout[-400000:])

ours = [e for e in L.read_trace_digest("/tmp/t300_steps.bin") if e["tag"] == "B"]
print("our B sizes", [e["n"] for e in ours][-12:])
print("ref B sizes", [int(v) for v in g["n_after"][len(ours) - 12:len(ours)]])
print("hash equal ", [bool(np.array_equal(e["list_hash"], g["list_hash"][k])) for k, e in enumerate(ours)][-12:])
print("proj diff  ", [float(np.abs(e["proj"] - g["proj"][k]).max()) for k, e in enumerate(ours)][-12:])
rc, nf, wd, nx, forks = L.read_forks("/tmp/t300_steps.bin.forks")
print("rc", rc, "forks", nf, [f for f in forks][:6])

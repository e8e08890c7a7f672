#!/bin/bash
# Where kfsp_begin_step's wall time goes in the adaptive Goutsias run (T = 300, N -> 1.03e6):
# per call, host time to enqueue its two kernels + copy vs time blocked in hipStreamSynchronize.
# Run from the repo root on a GPU box; writes gpurun_out/r02/bs/begin_step_trace.txt
set -e
R=$PWD
O=$R/gpurun_out/r02/bs
mkdir -p $O
cd tests/golden/models
KFSP_CASE_CAPACITY=2097169 KFSP_TRACE_BEGIN=1 $R/krylovfspssa_amd/fortran/_build/kfsp_dump solve goutsias_input /tmp/kfsp_g.bin 300 > $O/tb.log 2> $O/tb.err
python3 - "$O/tb.err" > $O/begin_step_trace.txt <<'PY'
import re, sys, statistics as st
e = [tuple(map(float, re.findall(r"=([0-9.]+)", l))) for l in open(sys.argv[1]) if "TRACE_BEGIN" in l]
print("calls", len(e), "enqueue total ms %.1f" % (sum(x[1] for x in e) / 1e3), "sync total ms %.1f" % (sum(x[2] for x in e) / 1e3))
for lo, hi in ((0, 1e4), (1e4, 1e5), (1e5, 5e5), (5e5, 2e6)):
    s = [x for x in e if lo <= x[0] < hi]
    if s:
        print(f"n in [{lo:g},{hi:g}): {len(s)} calls, enqueue median {st.median([x[1] for x in s]):.1f} us max {max(x[1] for x in s):.1f}, "
              f"sync median {st.median([x[2] for x in s]):.1f} us max {max(x[2] for x in s):.1f}")
PY
grep "KFSP" $O/tb.log >> $O/begin_step_trace.txt
cat $O/begin_step_trace.txt

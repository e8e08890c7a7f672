#!/bin/bash
# ROUND-1 SCRIPT, kept for the record: it ran the reference's example drivers compiled against our modules on the GPU
# box; since round 2 those binaries no longer travel (.gpurunignore) and end-to-end runs use `kfsp_dump solve` (profiles/begin_step_trace.sh).
# Regenerates the round's measured artefacts on a GPU box (run from the repo root
# through gpurun); outputs land in gpurun_out/ and are copied into profiles/ by hand.
set -e
R=$PWD
O=$R/gpurun_out
mkdir -p $O/prof
python bench.py > $O/bench_final.json 2> $O/bench_final.err
python bench.py --force-comm --no-cpu > $O/bench_final_comm1.json 2> $O/bench_final_comm1.err
python bench.py --workload c4 --no-expv > $O/bench_final_c4.json 2> $O/bench_final_c4.err
python bench.py --workload c5s --no-expv --no-cpu > $O/bench_final_c5s.json 2> $O/bench_final_c5s.err
python bench.py --workload c3x --no-expv --no-cpu > $O/bench_final_c3x.json 2> $O/bench_final_c3x.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o final -- python3 $R/bench.py --no-cpu > $O/bench_final_prof.json 2> $O/bench_final_prof.err
cd $R
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/final_kernel_stats.csv
B=krylovfspssa_amd/fortran/_build
(cd /tmp && /opt/rocm/lib/llvm/bin/flang -O3 -fopenmp -I$R/$B $R/profiles/statespace_bench.f90 $R/$B/libkfsp_fortran.a -o /tmp/ssb)
( lscpu | grep "Model name"; for t in 1 16; do echo "KFSP_HOST_THREADS=$t"; KFSP_HOST_THREADS=$t /tmp/ssb 2.0 38 | tail -2; done ) > $O/statespace_bench.log 2>&1
cd tests/golden/models
( for p in toggle TestSolverFromFile repressilator transcr6d; do
    s=$(date +%s.%N); $R/$B/ref_examples/$p > $O/ex_$p.log 2>&1; e=$(date +%s.%N)
    python3 -c "print('$p wall', round($e-$s,2), 's')"
    grep -i "KFSP WALL\|KFSP HOST\|KFSP STATS\|elapsed" $O/ex_$p.log
  done ) > $O/examples_end_to_end.log 2>&1
cat $O/bench_final.json

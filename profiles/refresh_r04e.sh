#!/bin/bash
# Round 4, end to end (run from the repo root through gpurun):
#  (1) the reference's own four drivers - examples/toggle, test/TestSolverFromFile, examples/repressilator, examples/transcr6d,
#      compiled UNCHANGED against our modules in the build container (krylovfspssa_amd/fortran/_build/ref_examples) - in the
#      default mode (the reference's sampling order) and with KFSP_SSA_STREAMS=1 (resident mode; the three examples attach
#      compiled-in CUSTOMPROP functions: probed, tabulated, verified - DESIGN.md 11.2)
#  (2) kfsp_dump solve goutsias_input 300 (models/goutsias_model.input) in the same two modes and over 2 loop-back ranks
#  (3) rocprofv3 kernel statistics of the resident transcr6d run
R=$PWD
O=$R/gpurun_out/r04
B=krylovfspssa_amd/fortran/_build
mkdir -p $O
cd tests/golden/models
( for mode in default streams; do
    for p in toggle TestSolverFromFile repressilator transcr6d; do
      if [ $mode = default ] && [ $p = transcr6d ]; then extra="KFSP_NOTE=reference_sampling_order"; else extra="KFSP_NOTE=$mode"; fi
      s=$(date +%s.%N)
      if [ $mode = streams ]; then env KFSP_SSA_STREAMS=1 $R/$B/ref_examples/$p > $O/ex_${mode}_$p.log 2>&1; else $R/$B/ref_examples/$p > $O/ex_${mode}_$p.log 2>&1; fi
      e=$(date +%s.%N)
      python3 -c "print('== $p [$mode] process wall', round($e-$s,2), 's')"
      grep -i "KFSP WALL\|KFSP HOST STATE\|KFSP STATS\|KFSP MODE\|KFSP CUSTOMPROP\|elapsed\|REPEATING" $O/ex_${mode}_$p.log
    done
  done ) > $O/examples_end_to_end.log 2>&1
D=$R/$B/kfsp_dump
export KFSP_CASE_CAPACITY=2097169
run() { name=$1; shift; t0=$(date +%s.%N); env "$@" $D solve goutsias_input $O/e2e_$name.bin 300.0 > $O/e2e_$name.log 2>&1; t1=$(date +%s.%N); \
        echo "== $name: $@  (process wall $(python3 -c "print(round($t1 - $t0, 2))") s)"; grep -E "KFSP|FINAL" $O/e2e_$name.log; }
( run default KFSP_NOTHING=1
  run resident KFSP_SSA_STREAMS=1
  run resident_again KFSP_SSA_STREAMS=1
  run resident_ranks2 KFSP_SSA_STREAMS=1 KFSP_NRANKS=2 ) > $O/end_to_end_goutsias.txt 2>&1
cmp $O/e2e_resident.bin $O/e2e_resident_again.bin && echo "resident runs byte-identical" >> $O/end_to_end_goutsias.txt
rm -f $O/e2e_*.bin $O/e2e_*.bin.in
cd /tmp && export TMPDIR=/tmp
( cd $R/tests/golden/models && KFSP_SSA_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e2e -o e2e -- $R/$B/ref_examples/transcr6d > $O/e2e_prof.log 2>&1 )
cp $(find /tmp/prof_e2e -name "e2e_kernel_stats.csv" | head -1) $O/e2e_resident_transcr6d_kernel_stats.csv
cd $R
cat $O/examples_end_to_end.log $O/end_to_end_goutsias.txt

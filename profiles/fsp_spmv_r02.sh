#!/bin/bash
# SpMV rate on an SSA-grown Goutsias FSP (N ~ 1.0e6) in the caller's discovery order vs the
# internal state order (default since round 2); writes gpurun_out/r02/fsp_spmv_timing.log
set -e
R=$PWD
B=$R/krylovfspssa_amd/fortran/_build
mkdir -p $R/gpurun_out/r02
(cd /tmp && /opt/rocm/lib/llvm/bin/flang -O3 -fopenmp -I$B $R/profiles/statespace_bench.f90 $B/libkfsp_fortran.a -L$R/krylovfspssa_amd/lib -lkfsp_hip -Wl,-rpath,$R/krylovfspssa_amd/lib -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib/llvm/lib -o /tmp/ssb)
KFSP_HOST_THREADS=16 /tmp/ssb 2.0 38 /tmp/fsp.bin | tail -2
python profiles/fsp_spmv_timing.py /tmp/fsp.bin | tee $R/gpurun_out/r02/fsp_spmv_timing.log

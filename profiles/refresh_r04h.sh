#!/bin/bash
# Round 4: does the way the host waits for the device matter?  The resident Goutsias run and examples/toggle's workload with
# KFSP_SYNC = (unset) / spin / yield / block (hipSetDeviceFlags before the context is created).
R=$PWD
O=$R/gpurun_out/r04
mkdir -p $O
cd tests/golden/models
D=$R/krylovfspssa_amd/fortran/_build/kfsp_dump
export KFSP_CASE_CAPACITY=2097169
for how in auto spin yield block; do
  if [ $how = auto ]; then unset KFSP_SYNC; else export KFSP_SYNC=$how; fi
  KFSP_SSA_STREAMS=1 timeout -k 10 120 $D solve goutsias_input /tmp/p.bin 300.0 > $O/sync_$how.log 2>&1
  echo "== resident Goutsias T=300, KFSP_SYNC=$how"; grep -E "KFSP WALL" $O/sync_$how.log
  timeout -k 10 60 $D solve toggle_example /tmp/p.bin 100.0 > $O/sync_toggle_$how.log 2>&1
  echo "== toggle_example T=100, KFSP_SYNC=$how"; grep -E "KFSP WALL" $O/sync_toggle_$how.log
done
cd $R
unset KFSP_SYNC
python3 profiles/expv_c2.py 40
KFSP_SYNC=spin python3 profiles/expv_c2.py 40
python3 profiles/pencil_sweep_r04.py c5 waves

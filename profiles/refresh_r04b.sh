#!/bin/bash
# Round 4: small-tile trip orders for the 6-species boxes (profiles/trip_order_sweep_r04.py), matrix-free and stored.
R=$PWD
O=$R/gpurun_out/r04
mkdir -p $O
python3 profiles/trip_order_sweep_r04.py c5s mf > $O/trip_c5s_mf.log 2>&1
tail -15 $O/trip_c5s_mf.log
python3 profiles/trip_order_sweep_r04.py c5 mf 1024 2048 4096 8192 16384 > $O/trip_c5_mf.log 2>&1
tail -12 $O/trip_c5_mf.log
python3 profiles/trip_order_sweep_r04.py c5s stored 2048 4096 8192 > $O/trip_c5s_stored.log 2>&1
tail -8 $O/trip_c5s_stored.log

#!/bin/bash
# ablation timings of the thin-input sliding-window conv (conv_thin.hip): which side binds a launch
# (needs scratch/lib_thin_abl.so from tools/tuning/build_thin_abl.sh)
echo "== in-tree library"; python tools/bench_conv.py thin
export WSR_LIB_PATH=$PWD/scratch/lib_thin_abl.so
for abl in ${ABLS:-0 1 2 4 8 6 7 15 31}; do
  echo "== WSR_CT3_ABL=$abl"
  WSR_CT3_ABL=$abl python tools/bench_conv.py thin 2>/dev/null
done

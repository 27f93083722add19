#!/bin/bash
# scratch/lib_thin_abl.so: the library with conv_thin.hip compiled -DWSR_CT3_ABL_RT (run-time ablation switch WSR_CT3_ABL)
set -e
cd /root/repo/gan_sr_wind_field_amd/csrc
mkdir -p /root/repo/scratch
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -DWSR_CT3_ABL_RT -c conv_thin.hip -o /tmp/conv_thin_abl.o
objs=$(ls *.o | grep -v '^conv_thin.o$' | grep -v 'conv_1x1.o\|conv_tile_w4.o')
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/conv_thin_abl.o -o /root/repo/scratch/lib_thin_abl.so
echo built scratch/lib_thin_abl.so

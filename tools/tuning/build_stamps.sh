#!/bin/bash
# tuning build with in-kernel phase stamps -> scratch/stamps/libwindsr_hip.so (the bf16 tile units and the tile filter
# gradient recompiled -DWSR_CT_STAMPS, everything else linked from the default build's objects)
set -e
cd /root/repo/gan_sr_wind_field_amd/csrc
make -j8 > /dev/null
mkdir -p /root/repo/scratch/stamps
units="conv_tile conv_tile_n144 conv_tile_n128 conv_tile_narrow conv_tile_wide conv_tile_masked conv_tile_narrow_masked conv_tile_small conv_tile_tm3 conv_tile_simple_narrow conv_tile_simple_n128 conv_tile_simple_small conv_wgrad_tile"
for f in $units; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -DWSR_CT_STAMPS -c $f.hip -o /root/repo/scratch/stamps/$f.o &
done
wait
rest=""
for s in $(sed -n 's/^SRCS = //p' Makefile); do
  b=${s%.hip}
  case " $units " in *" $b "*) ;; *) rest="$rest $b.o" ;; esac
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $rest /root/repo/scratch/stamps/*.o -o /root/repo/scratch/stamps/libwindsr_hip.so
echo built scratch/stamps/libwindsr_hip.so

#!/bin/bash
# tuning build with in-kernel phase stamps -> scratch/stamps/libwindsr_hip.so
set -e
cd /root/repo/gan_sr_wind_field_amd/csrc
mkdir -p /root/repo/scratch/stamps
for f in conv_tile conv_tile_n144 conv_tile_n128 conv_tile_narrow conv_tile_wide conv_tile_masked conv_tile_narrow_masked conv_tile_small conv_wgrad_tile; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -DWSR_CT_STAMPS -c $f.hip -o /root/repo/scratch/stamps/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 conv_igemm.o conv_1x1.o conv_1x1_v2.o conv_wgrad.o elementwise.o physics_loss.o conv_tile_strided.o conv_tile_w4.o conv_slide.o conv_wgrad_tile_f32.o /root/repo/scratch/stamps/*.o -o /root/repo/scratch/stamps/libwindsr_hip.so

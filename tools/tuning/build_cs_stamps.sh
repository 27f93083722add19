#!/bin/bash
# tuning build of conv_slide.hip with in-kernel phase stamps -> scratch/lib_cs_stamps${WSR_CS_TAG}.so (other objects: in-tree build)
set -e
R=/root/repo
mkdir -p $R/scratch/stamps
make -C $R/gan_sr_wind_field_amd/csrc > /dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -DWSR_CS_STAMPS=${WSR_CS_STAMPS_MODE:-1} $WSR_CS_EXTRA -c $R/gan_sr_wind_field_amd/csrc/conv_slide.hip -o $R/scratch/stamps/conv_slide.o
OBJS=$(ls $R/gan_sr_wind_field_amd/csrc/*.o | grep -v conv_slide.o | tr '\n' ' ')
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS $R/scratch/stamps/conv_slide.o -o $R/scratch/lib_cs_stamps${WSR_CS_TAG}.so
ls -la $R/scratch/lib_cs_stamps${WSR_CS_TAG}.so

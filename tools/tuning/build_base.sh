#!/bin/bash
# build the library of a given commit (default HEAD) into scratch/base/libwindsr_hip.so for A/B timing
set -e
rev=${1:-HEAD}
rm -rf /root/repo/scratch/base && mkdir -p /root/repo/scratch/base/gan_sr_wind_field_amd/csrc /root/repo/scratch/base/include
cd /root/repo
for f in $(git ls-tree --name-only $rev gan_sr_wind_field_amd/csrc/); do git show $rev:$f > scratch/base/$f; done
git show $rev:include/windsr_hip.h > scratch/base/include/windsr_hip.h
make -C scratch/base/gan_sr_wind_field_amd/csrc -j8 > scratch/base/build.log 2>&1
cp scratch/base/gan_sr_wind_field_amd/csrc/libwindsr_hip.so scratch/base/libwindsr_hip.so
rm -rf scratch/base/gan_sr_wind_field_amd scratch/base/include
ls -la scratch/base/libwindsr_hip.so

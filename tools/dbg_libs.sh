#!/bin/bash
# development aid: run a python probe under every library build in ab_libs/
cp diffsim_amd/libdiffsim_amd.so /tmp/lib_orig.so
for f in ab_libs/lib_*.so; do
  echo "=== $f"; cp $f diffsim_amd/libdiffsim_amd.so; python "$@" 2>&1 | grep -v amdgpu.ids | tail -40
done
cp /tmp/lib_orig.so diffsim_amd/libdiffsim_amd.so

#!/bin/bash
# Diagnostic build: a stamped copy of the library under tools/dbg/ and the kbench harness linked against it.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/dbg
for f in core feed loss metrics optim dnn; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DULTR_STAMPS -Wno-unused-function -c ultra_amd/csrc/$f.hip -o tools/dbg/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dbg/libultr_hip.so tools/dbg/*.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/kbench.hip -Iinclude -Ltools/dbg -lultr_hip -Wl,-rpath,'$ORIGIN/dbg' -o tools/kbench
echo built tools/kbench

#!/bin/bash
# Diagnostic build: a stamped copy of the library under tools/dbg/ and the fbench harness linked against it.
# Only the sources newer than their object are rebuilt.  EXTRA=... adds compiler flags.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/dbg
for f in core feed loss metrics optim dnn fused; do
  if [ ! -f tools/dbg/$f.o ] || [ ultra_amd/csrc/$f.hip -nt tools/dbg/$f.o ] || [ -n "$FORCE" ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DULTR_STAMPS $EXTRA -Wno-unused-function -c ultra_amd/csrc/$f.hip -o tools/dbg/$f.o &
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dbg/libultr_hip.so tools/dbg/*.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/fbench.hip -Iinclude -Ltools/dbg -lultr_hip -Wl,-rpath,'$ORIGIN/dbg' -o tools/fbench
echo built tools/fbench

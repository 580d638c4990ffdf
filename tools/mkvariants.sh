#!/bin/bash
# Build variant libraries locally (CPU container), in parallel, into build/variants/<name>.so (git-ignored, shipped
# to the GPU box by gpurun).  usage: tools/mkvariants.sh "name:-DFLAG=1 -DOTHER=2" ...
mkdir -p build/variants
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  [ "$flags" = "$spec" ] && flags=""
  ( python spectrobot_amd/build.py --out build/variants/$name.so $flags > build/variants/$name.log 2>&1 || echo "BUILD FAILED $name" ) &
done
wait
ls -la build/variants/*.so

#!/bin/bash
# Build a variant of the library for a same-box A/B:  tools/build_variant.sh <tag> <file.hip> <sed-expression>
# -> mmvid_amd/libmmvid_hip.so.<tag> (git-ignored, travels with gpurun); run with MMVID_LIB=$PWD/mmvid_amd/libmmvid_hip.so.<tag>
set -e
cd "$(dirname "$0")/.."
tag=$1; f=$2; expr=$3
python -m mmvid_amd.build > /dev/null
tmp=mmvid_amd/csrc/_variant_$tag.hip
sed -e "$expr" mmvid_amd/csrc/$f.hip > $tmp
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -c $tmp -o mmvid_amd/_obj/_variant_$tag.o
objs=$(ls mmvid_amd/_obj/*.o | grep -v "_variant_" | grep -v "/$f.o")
hipcc --offload-arch=gfx950 -shared -fPIC $objs mmvid_amd/_obj/_variant_$tag.o -o mmvid_amd/libmmvid_hip.so.$tag
rm -f $tmp
echo built mmvid_amd/libmmvid_hip.so.$tag

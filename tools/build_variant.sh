#!/bin/bash
# Experiment builds of the library next to the product build (run here, on the CPU: hipcc cross-compiles):
#   tools/build_variant.sh <name> "<extra compiler flags>"   ->  mizuroute_amd/lib_var/<name>/libmzr_hip.so   (use with MZR_LIB=<that path>)
set -e
name=$1; extra=$2
root=$(cd "$(dirname "$0")/.." && pwd)
b=/tmp/mzr_build_$name
rm -rf $b; mkdir -p $b/csrc $b/include $root/mizuroute_amd/lib_var/$name
cp $root/mizuroute_amd/csrc/*.hip $root/mizuroute_amd/csrc/*.h $root/mizuroute_amd/csrc/Makefile $b/csrc/
mkdir -p $b/../include_$name; cp $root/include/*.h $b/include/
# the Makefile looks for ../../include/mzr.h: give the copy the same shape
mkdir -p $b/pkg/csrc; mv $b/csrc/* $b/pkg/csrc/; rmdir $b/csrc; mkdir -p $b/pkg/lib
( cd $b/pkg/csrc && sed -i 's#\.\./\.\./include#../../include#' Makefile && make -j8 EXTRA="$extra" OUT=../lib/libmzr_hip.so >/dev/null 2>$b/build.err || { cat $b/build.err | grep -E "error" -A3 | head -20; exit 1; } )
cp $b/pkg/lib/libmzr_hip.so $root/mizuroute_amd/lib_var/$name/libmzr_hip.so
echo "built mizuroute_amd/lib_var/$name/libmzr_hip.so [$extra]"

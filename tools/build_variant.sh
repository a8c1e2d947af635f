#!/bin/bash
# Build a compile-time variant of the library beside the default one: neuradar_amd/csrc/variants/<name>.so (git-ignored; travels to
# the GPU box with gpurun).  Select it with NR_LIB_PATH=<path> (tools/ab_env.sh "NR_LIB_PATH=..." "-").
# usage: build_variant.sh <name> "<EXTRA compile flags>"
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
name=$1; flags=$2
tmp=$(mktemp -d)
mkdir -p $tmp/neuradar_amd/csrc $tmp/include $root/neuradar_amd/csrc/variants
cp $root/neuradar_amd/csrc/*.hip $root/neuradar_amd/csrc/*.h $root/neuradar_amd/csrc/Makefile $tmp/neuradar_amd/csrc/
cp $root/include/*.h $tmp/include/
make -s -j8 -C $tmp/neuradar_amd/csrc EXTRA="$flags" > /dev/null
cp $tmp/neuradar_amd/csrc/libneuradar_hip.so $root/neuradar_amd/csrc/variants/$name.so
rm -rf $tmp
echo "built neuradar_amd/csrc/variants/$name.so [$flags]"

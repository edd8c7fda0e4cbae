#!/bin/bash
# Build an A/B variant of libns2hip.so WITHOUT touching the product source: copy csrc to a scratch directory, apply one sed
# expression, build, and leave the library as tools/ab/libns2hip_<name>.so (git-ignored; travels to the GPU box with the snapshot).
#   tools/build_ab_variant.sh nowskip 's/const bool w_skip = .*/constexpr bool w_skip = false;/' gemm2.hip
# Select it on the GPU box with NS2_LIB=tools/ab/libns2hip_<name>.so (naturalspeech2_pytorch_amd/_lib.py).
set -e
name=$1; expr=$2; file=$3
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
mkdir -p $tmp/naturalspeech2_pytorch_amd $root/tools/ab
cp -r $root/naturalspeech2_pytorch_amd/csrc $tmp/naturalspeech2_pytorch_amd/csrc
cp -r $root/include $tmp/include
rm -rf $tmp/naturalspeech2_pytorch_amd/csrc/obj
sed -i "$expr" $tmp/naturalspeech2_pytorch_amd/csrc/$file
if cmp -s $tmp/naturalspeech2_pytorch_amd/csrc/$file $root/naturalspeech2_pytorch_amd/csrc/$file; then echo "sed expression changed nothing" >&2; exit 1; fi
bash $tmp/naturalspeech2_pytorch_amd/csrc/build.sh
cp $tmp/naturalspeech2_pytorch_amd/libns2hip.so $root/tools/ab/libns2hip_$name.so
rm -rf $tmp
echo "built tools/ab/libns2hip_$name.so"

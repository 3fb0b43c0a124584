#!/bin/bash
# Build an experimental variant of one problem's product library into ddp-generator_amd/lib_<name>
# (git-ignored; travels to the GPU box):   tools/variant.sh <name> "<extra hipcc flags>" [problem] [full_ddp]
# Run it with ILQG_LIBDIR=$PWD/ddp-generator_amd/lib_<name> python bench.py ...
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; FLAGS=$2; PROB=${3:-synth16x8}; FD=${4:-1}
make -s -C $R/ddp-generator_amd/csrc PROBLEMS=$PROB WAVE_PROBLEMS= LIBDIR=../lib_$NAME OBJDIR=../build_$NAME \
    EXTRA_HIPFLAGS="$FLAGS" ../lib_$NAME/libilqg_${PROB}_fd${FD}_hip.so 2>&1 | grep -v "argument unused" || true
cp -n $R/ddp-generator_amd/lib/*.so $R/ddp-generator_amd/lib_$NAME/  # everything else: the product build
ls -la $R/ddp-generator_amd/lib_$NAME/ | grep ${PROB}_fd${FD}_hip.so

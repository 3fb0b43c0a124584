#!/bin/bash
# Device assembly + resources of ONE kernel of a problem's translation unit, without the shim and the other kernels
# (seconds instead of minutes):   tools/one_kernel.sh <problem> <fd> '<kernel signature>' [extra hipcc flags] > out.s
#   tools/one_kernel.sh synth16x8 1 'k_backward_quad<true>(DevPtrs, ilqg_dev_opts_t, int, int, int)' -DILQG_QUAD_WAVES=8
# Prints the .s on stdout, the register / spill summary on stderr.  Needs the problem's build directory (make first).
R=$(cd "$(dirname "$0")/.." && pwd)
PROB=$1; FD=$2; SIG=$3; shift 3
PD=$R/problems/$PROB; [ -d "$PD" ] || PD=$R/ddp-generator_amd/build/plain/$PROB
BIG=""; case $PROB in synth16x8*|synth16p*) BIG="-DILQG_SINCOS_CALL -mllvm -disable-machine-licm";; esac
FP="-ffp-contract=fast-honor-pragmas"; [ -n "$STRICT" ] && FP="-ffp-contract=off -DILQG_STRICT_FP=1"   # STRICT=1: the FMA-free twin
OUT=$(mktemp /tmp/onek.XXXXXX.s)
hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -Wno-writable-strings -Wno-extern-c-compat -Wno-unused-value \
  -DHAVE_OCTAVE $BIG $FP -DFULL_DDP=$FD "-DILQG_ONLY_KERNEL=$SIG" "$@" \
  -I$R/include/mex_stub -I$R/include -I$R/ddp-generator_amd/csrc -I$PD -I$R/ddp-generator_amd/build/${PROB}_fd$FD \
  $R/ddp-generator_amd/csrc/ilqg_kernels.hip -o $OUT || exit 1
cat $OUT
grep -E "^\s*; (NumVgprs|NumAgprs|TotalNumVgprs|ScratchSize|Occupancy|NumSgprs|VGPR spill|SGPR spill|LDSByteSize)|\.vgpr_spill_count|^\s*\.name:" $OUT | grep -v "^\s*\.name:.*__" >&2
rm -f $OUT

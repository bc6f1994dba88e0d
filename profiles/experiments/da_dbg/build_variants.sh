#!/bin/bash
# Diagnostic builds of libhdrsky.so for the distortion-aware data-gradient defect: the stock objects with
# da_conv.hip recompiled under different flags.  Outputs: profiles/experiments/da_dbg/lib_<variant>.so (git-ignored).
set -e
ROOT=$(cd "$(dirname "$0")/../../.." && pwd)
PKG=$ROOT/hdr-map-reconstruction-from-a-single-ldr-sky-panoramic-image-for-outdoor-illumination-estimation_amd
OUT=$ROOT/profiles/experiments/da_dbg
HIPCC=/opt/rocm/bin/hipcc
BASE="-O3 -fPIC --offload-arch=gfx950 -std=c++17 -I$ROOT/include -I$PKG/csrc"
make -C $PKG/csrc -j8 >/dev/null
OTHERS=$(ls $PKG/csrc/*.o | grep -v da_conv.o)
build() {  # name, extra flags
  local name=$1; shift
  $HIPCC $BASE "$@" -c $PKG/csrc/da_conv.hip -o $OUT/da_conv_$name.o
  $HIPCC -shared -fPIC --offload-arch=gfx950 -o $OUT/lib_$name.so $OTHERS $OUT/da_conv_$name.o
  echo built $name
}
for v in "$@"; do
  case $v in
    base) build base ;;
    dbg) build dbg -DHDRSKY_DA_DEBUG ;;
    pad) build pad -mllvm -amdgpu-mfma-padding-ratio=100 ;;
    noslp) build noslp -fno-slp-vectorize ;;
    dbg2) build dbg2 -DHDRSKY_DA_DEBUG2 ;;
    exp1) build exp1 -DHDRSKY_DA_DEBUG -DHDRSKY_DA_EXP=1 ;;
    exp2) build exp2 -DHDRSKY_DA_DEBUG -DHDRSKY_DA_EXP=2 ;;
    exp4) build exp4 -DHDRSKY_DA_DEBUG -DHDRSKY_DA_EXP=4 ;;
    vgprform) build vgprform -DHDRSKY_DA_DEBUG -mllvm -amdgpu-mfma-vgpr-form ;;
    dbgh) build dbgh -DHDRSKY_DA_DEBUG -DHDRSKY_DA_DEBUG_SEL=4 ;;
    dbgf) build dbgf -DHDRSKY_DA_DEBUG -DHDRSKY_DA_DEBUG_SEL=5 ;;
    dbgw) build dbgw -DHDRSKY_DA_DEBUG -DHDRSKY_DA_DEBUG_SEL=1 ;;
    dbgo) build dbgo -DHDRSKY_DA_DEBUG -DHDRSKY_DA_DEBUG_SEL=2 ;;
    dbgr) build dbgr -DHDRSKY_DA_DEBUG -DHDRSKY_DA_DEBUG_SEL=3 ;;
    *) echo "unknown variant $v"; exit 1 ;;
  esac
done

#!/bin/bash
# Builds libtgs_hip.so for gfx950 in-tree (touch_gs_amd/lib/).  hipcc cross-compiles without a GPU.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wall -Wno-unused-function $TGS_EXTRA_FLAGS"
SRCS="api project binning raster optim imgloss peer"
# raster.hip: the SLP vectoriser pairs the per-pixel FMAs into v_pk_fma_f32, which on gfx950 costs
# 1.84x a plain v_fma_f32 and needs register-pair shuffles (v_mov) around it: K7 127 -> 100 VGPRs and
# -7 % time, K6 -3 % without it (measured, same box)
# imgloss.hip: the same for the SSIM filters (k_ssim_fwd 2105 static VALU with 341 v_pk_* and 326 v_mov at 110
# VGPRs -> 2232 VALU, 143 v_mov, 92 VGPRs: 58.4 -> 51.8 us at 1080p, k_ssim_bwd unchanged; same box, same checksum).
# The other translation units hold HBM-bound kernels: no difference either way (1.0129 / 1.0122 ms per step).
# project.hip: -ffp-contract=on (fusion per source expression, decided by the front end) instead of =fast (decided by
# the backend after inlining): geom_eval / project_fwd_core are inlined into the stand-alone K1 AND into the fused
# optimizer kernel (front prefetch), and the two must produce the same bits.
declare -A EXTRA=([raster]="-fno-slp-vectorize" [imgloss]="-fno-slp-vectorize" [project]="-ffp-contract=on")
pids=()
for s in $SRCS; do
  [ -f "$HERE/$s.hip" ] || continue
  stale=0
  [ -f "$OUT/$s.o" ] || stale=1
  for dep in "$HERE/$s.hip" "$HERE"/*.h "$HERE/../../include/tgs.h" "$0"; do
    [ "$dep" -nt "$OUT/$s.o" ] && stale=1
  done
  if [ $stale = 1 ]; then
    $HIPCC $FLAGS ${EXTRA[$s]} -c "$HERE/$s.hip" -o "$OUT/$s.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
OBJS=""
for s in $SRCS; do [ -f "$OUT/$s.o" ] && OBJS="$OBJS $OUT/$s.o"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC $OBJS -o "$OUT/libtgs_hip.so"
echo "built $OUT/libtgs_hip.so"

#!/bin/bash
# Builds a variant of libtgs_hip.so for same-box A/B runs (tools/ab.py / tools/abn.py):
#   bash tools/build_variant.sh <name> [extra hipcc flags, e.g. -DTGS_GID_PREFETCH] [RASTER=<path to an alternative raster.hip>]
# Output: build_ab/<name>.so.  Objects of unchanged translation units are taken from touch_gs_amd/lib/.
set -e
NAME=$1; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC=$ROOT/touch_gs_amd/csrc
OUT=$ROOT/build_ab/obj_$NAME
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wall -Wno-unused-function -I$SRC"
declare -A EXTRA=([raster]="-fno-slp-vectorize" [imgloss]="${TGS_IMGLOSS_FLAGS--fno-slp-vectorize}" [project]="-ffp-contract=on")
declare -A ALT
ARGS=()
for a in "$@"; do
  case "$a" in
    -*) ARGS+=("$a");;
    *=*) k=${a%%=*}; ALT[${k,,}]=${a#*=};;
    *) ARGS+=("$a");;
  esac
done
pids=()
for s in api project binning raster optim imgloss peer; do
  f=${ALT[$s]:-$SRC/$s.hip}
  /opt/rocm/bin/hipcc $FLAGS ${EXTRA[$s]} "${ARGS[@]}" -c "$f" -o $OUT/$s.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OUT/*.o -o $ROOT/build_ab/$NAME.so
rm -rf $OUT
echo "built build_ab/$NAME.so"

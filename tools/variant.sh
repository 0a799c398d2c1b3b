#!/bin/bash
# development A/B: bash tools/variant.sh <name> <file.hip> [-DFLAG ...]  ->  fft_amd/lib/libvar_<name>.so (the library with one source rebuilt
# under extra flags; select it with ARMFFT_LIB=fft_amd/lib/libvar_<name>.so)
set -e
NAME=$1; SRC=$2; shift 2
cd "$(dirname "$0")/.."
OBJ=fft_amd/build/var_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -ffp-contract=fast "$@" -x hip -c fft_amd/csrc/$SRC -o $OBJ
OTHERS=$(ls fft_amd/build/*.o | grep -v "/var_" | grep -v "$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o fft_amd/lib/libvar_$NAME.so $OBJ $OTHERS
echo fft_amd/lib/libvar_$NAME.so

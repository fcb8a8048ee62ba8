#!/bin/bash
# Kernel experiments: build another libwaveletsext_hip.so in which ONE translation unit is compiled with extra flags.
#   tools/mkvariant.sh <name> <file.hip> [extra hipcc flags...]   ->   tools/dbg/lib/libwx_<name>.so
# Select it at run time with WX_HIP_LIB=tools/dbg/lib/libwx_<name>.so (see waveletsext.jl_amd/_lib.py).
set -e
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/waveletsext.jl_amd/csrc
mkdir -p $root/tools/dbg/lib
make -s -C $csrc -j8 >/dev/null
obj=$root/tools/dbg/lib/${name}_$(basename ${src%.hip}).o
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast"
/opt/rocm/bin/hipcc $flags -I$root/include -I$csrc "$@" -c $csrc/$src -o $obj
others=$(ls $csrc/*.o | grep -v "/$(basename ${src%.hip}).o$")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $root/tools/dbg/lib/libwx_${name}.so $others $obj
echo $root/tools/dbg/lib/libwx_${name}.so

#!/bin/bash
# Experiment build of the library:
#   bash scripts/build_variant.sh <name> "<extra hipcc flags>" ["objects to rebuild" = "update policy"]
# → cleanrl.jl_amd/variants/<name>/libcleanrl_hip.so (git-ignored; travels to the GPU box). Run it with
#   CRL_LIB_PATH=cleanrl.jl_amd/variants/<name>/libcleanrl_hip.so python bench.py …
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/cleanrl.jl_amd/csrc
mkdir -p $R/cleanrl.jl_amd/variants/$1
make -s -C $C -j8 > /dev/null 2>&1 || true          # the default build first: its objects seed the variant's build directory
rm -rf $C/build_$1 && cp -r $C/build $C/build_$1
for o in ${3:-update policy}; do rm -f $C/build_$1/$o.o; done
make -s -C $C -j8 BUILD=build_$1 OUT=../variants/$1/libcleanrl_hip.so EXTRA="$2" 2>&1 | grep -E "error|Error" || true
ls -la $R/cleanrl.jl_amd/variants/$1/libcleanrl_hip.so

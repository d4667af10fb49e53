#!/bin/bash
# tools/build_variants.sh name1="flags" name2="flags" ...  -> gpurun_variants/lib_<name>.so (only bgk_fused2.hip is recompiled)
cd "$(dirname "$0")/.."
mkdir -p gpurun_variants
python -m bgflow_amd.build --quiet > /dev/null 2>/tmp/build_base.log || { grep -m5 error /tmp/build_base.log; exit 1; }
for spec in "$@"; do
  name="${spec%%=*}"; flags="${spec#*=}"
  BGK_EXTRA_TU=${BGK_EXTRA_TU:-bgk_fused2.hip} BGK_EXTRA_FLAGS="$flags" python -m bgflow_amd.build --quiet --out gpurun_variants/lib_$name.so > /dev/null 2>/tmp/build_$name.log || echo "build of $name failed"
done
ls gpurun_variants/

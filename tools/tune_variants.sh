#!/bin/bash
# Builds tuning variants of the library next to the product build and benches each in the same process environment:
#   tools/tune_variants.sh "<name>:<extra hipcc flags>" ...     (run the build part in the container, the bench part on the GPU box)
# build:  tools/tune_variants.sh build  "bw2:-DGS_BWD_WAVES=2" ...
# bench:  tools/tune_variants.sh bench  bw2 ...
mode=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
if [ "$mode" = build ]; then
  for v in "$@"; do
    name=${v%%:*}; flags=${v#*:}
    d=/tmp/gsvar_$name; rm -rf $d; mkdir -p $d $root/build/variants
    cp $root/easy_gaussian_splatting_amd/csrc/*.hip $root/easy_gaussian_splatting_amd/csrc/*.h $root/easy_gaussian_splatting_amd/csrc/Makefile $d/
    sed -i "s|../../include/gs_raster.h|$root/include/gs_raster.h|g" $d/Makefile $d/*.h
    make -C $d -j4 EXTRA="$flags" LIB=$root/build/variants/libgsraster_$name.so 2>&1 | grep -E "error|Error" | head -3
    ls -la $root/build/variants/libgsraster_$name.so | awk '{print $5, $9}'
  done
else
  for name in base "$@"; do
    lib=$root/build/variants/libgsraster_$name.so; [ $name = base ] && lib=$root/easy_gaussian_splatting_amd/libgsraster.so
    GS_ALLOW_VARIANT=1 GS_LIB_PATH=$lib timeout 200 python $root/bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 20 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$name', d['value'], d['step_ms']['median'], d['forward_fps'], {k[3:]:round(v,3) for k,v in d['stage_ms'].items()})"
  done
fi

#!/bin/bash
# Shader clock under load, on the GPU box:  tools/clock_probe.sh <tag>   -> gpurun_out/<tag>_clock.json
# needs tools/micro/clock_probe (hipcc -O3 --offload-arch=gfx950 -o clock_probe clock_probe.hip) and the -DGS_CLOCK_PROBE variant
# of the library (tools/tune_variants.sh build "clk:-DGS_CLOCK_PROBE"), both built in the container.
tag=${1:-r04}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out; mkdir -p $out
$root/tools/micro/clock_probe > $out/${tag}_clock_micro.jsonl 2>&1
khz=$(python3 -c "import json,sys; print(json.loads(open('$out/${tag}_clock_micro.jsonl').readline())['wall_clock_rate_kHz'])")
GS_WALL_CLOCK_KHZ=$khz GS_ALLOW_VARIANT=1 GS_LIB_PATH=$root/build/variants/libgsraster_clk.so python3 $root/tools/clock_probe.py $out/${tag}_clock_kernels.json > /dev/null 2> $out/${tag}_clock_kernels.err
python3 - <<PY
import json
micro = [json.loads(l) for l in open("$out/${tag}_clock_micro.jsonl") if l.startswith("{")]
try:
    kern = json.load(open("$out/${tag}_clock_kernels.json"))
except Exception as e:
    kern = {"error": repr(e)}
json.dump({"micro": micro, "kernels": kern}, open("$out/${tag}_clock.json", "w"), indent=1)
print(json.dumps({"micro": micro, "kernels": kern})[:3000])
PY

#!/bin/bash
# One GPU-box call that regenerates everything under profiles/ for a round:  tools/final_profiles.sh r06
# (build the diagnostic variants in the container first:
#    tools/tune_variants.sh build "clk:-DGS_CLOCK_PROBE" "acc64:-DGS_BWD_ACC64" "exact:-DGS_BWD_ACC64 -DGS_EXACT_MATH"
#    hipcc -O3 --offload-arch=gfx950 -o tools/micro/clock_probe tools/micro/clock_probe.hip)
# (SQ counters and HBM traffic first: bench.py quotes them in its roofline objects)
tag=${1:-r06}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# the diagnostic variants, re-made on the box unless GS_SKIP_VARIANT_BUILD=1 (a variant older than the sources lacks the symbols newer
# sources export, and the package refuses to bind such a library: the tools that load one would fail silently)
if [ "$GS_SKIP_VARIANT_BUILD" != 1 ]; then
  bash tools/tune_variants.sh build "clk:-DGS_CLOCK_PROBE" "acc64:-DGS_BWD_ACC64" "exact:-DGS_BWD_ACC64 -DGS_EXACT_MATH" > gpurun_out/${tag}_variants.log 2>&1
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o tools/micro/clock_probe tools/micro/clock_probe.hip
fi
# shader clock under load first (bench.py's roofline_compute reads profiles/<tag>_clock.json)
timeout 300 bash tools/clock_probe.sh $tag > gpurun_out/${tag}_clock_stdout.txt 2>&1
cp gpurun_out/${tag}_clock.json profiles/ 2>/dev/null
timeout 600 bash tools/prof_sq.sh $tag > gpurun_out/${tag}_sq.log 2>&1
timeout 400 bash tools/prof_pmc.sh $tag > gpurun_out/${tag}_pmc.log 2>&1
cp gpurun_out/${tag}_sq_counters.json gpurun_out/${tag}_pmc_traffic.json profiles/ 2>/dev/null
python3 - $tag <<'PY'   # the blend kernels' counters on their own (what VERDICT r1 #2 asked for by name)
import json, sys
tag = sys.argv[1]
d = json.load(open(f"gpurun_out/{tag}_sq_counters.json"))
for part in ("fwd", "bwd"):
    json.dump({k: v for k, v in d.items() if f"blend_{part}" in k}, open(f"gpurun_out/{tag}_sq_blend_{part}.json", "w"), indent=1)
PY
timeout 300 bash tools/prof_bench.sh ${tag}_bench > gpurun_out/${tag}_prof_bench.log 2>&1
timeout 600 python bench.py > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.err
for m in tight gsplat_eager; do
  timeout 200 bash tools/prof_cmd.sh ${tag}_longlists_$m tools/long_lists_run.py $m 10 > /dev/null 2>&1
  grep -v "amdgpu.ids\|^W2026\|^E2026" /tmp/${tag}_longlists_$m.log > gpurun_out/${tag}_longlists_$m.txt
done
GS_BINNING=bins timeout 300 bash tools/prof_pmc.sh ${tag}_longlists tools/long_lists_run.py tight 3 > /dev/null 2>&1
timeout 300 bash tools/micro/traffic_cal.sh $tag > gpurun_out/${tag}_traffic_cal.log 2>&1
(GS_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 50 --warmup 10 > gpurun_out/${tag}_bench_forcedist_rccl.json 2> gpurun_out/${tag}_bench_forcedist.err)
# ... and its kernel trace (row_sums_kernel, project_bwd_kernel<3, false, true>, sh_grad_views_kernel<3, true>, RCCL's own kernels)
(cd /tmp && export TMPDIR=/tmp GS_BENCH_FORCE_DIST=1 && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/${tag}_fd -o fd -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 > /tmp/${tag}_fd.log 2>&1; f=$(find /tmp/${tag}_fd -name "*kernel_stats.csv" | head -1); cp "$f" $GRAFT_REPO_ROOT/gpurun_out/${tag}_forcedist_kernel_stats.csv)
timeout 900 python tools/config_table.py > gpurun_out/${tag}_configs.md 2> gpurun_out/${tag}_configs.err
timeout 200 python tools/binning_sweep.py 2>/dev/null > gpurun_out/${tag}_binning_sweep.txt
timeout 200 python tools/blend_time.py 2>/dev/null > gpurun_out/${tag}_blend_time.txt
# the workloads beside the headline: kernel trace + FETCH / WRITE traffic of eager train steps (tools/config_run.py)
for cfg in S3 S5 heavy1M heavy2M; do
  lc=$(echo $cfg | tr A-Z a-z)
  timeout 600 bash tools/prof_cmd.sh ${tag}_${lc} tools/config_run.py $cfg 6 > /dev/null 2>&1
  grep "^{" /tmp/${tag}_${lc}.log | tail -1 > gpurun_out/${tag}_${lc}_run.json
  timeout 900 bash tools/prof_pmc.sh ${tag}_${lc} tools/config_run.py $cfg 3 > /dev/null 2>&1
done
# the loss kernels alone (cold inputs, with and without a mask), the issue cost per operand kind, the loss forward's memory side alone
(timeout 120 python tools/loss_time.py; GS_LOSS_MASK=0 timeout 120 python tools/loss_time.py) 2>/dev/null > gpurun_out/${tag}_loss_time.txt
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o tools/micro/valu_enc tools/micro/valu_enc.hip
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o tools/micro/tile_stream tools/micro/tile_stream.hip
timeout 60 tools/micro/valu_enc > gpurun_out/${tag}_valu_enc.txt 2>/dev/null
timeout 60 tools/micro/tile_stream > gpurun_out/${tag}_tile_stream.txt 2>/dev/null
# where the captured step's memory goes (bytes per listed intersection), the host-fed loop, the end-to-end run, the lazy-binning bound
for c in "bench1M tight" "heavy2M gsplat_eager" "heavy2M tight" "S5 tight"; do timeout 300 python tools/mem_report.py $c 2>/dev/null; done > gpurun_out/${tag}_mem_report.jsonl
for a in "2 eager" "3 eager" "2 lazy"; do timeout 200 python tools/host_feed_probe.py $a 2>/dev/null; done > gpurun_out/${tag}_host_feed_probe.jsonl
timeout 300 python tools/e2e_train.py 3000 both gpurun_out/${tag}_e2e.json > /dev/null 2>&1
for c in "heavy2M tight" "heavy2M gsplat_eager" "heavy1M tight"; do timeout 300 python tools/lazy_bin_bound.py $c 2>/dev/null; done > gpurun_out/${tag}_lazy_bin_bound.jsonl
# depth rounds: the captured step in one round / in two at several splits / as rounds="auto" picks (front round alone where the probe
# finds no live tile), the forward fps through the eager seam, and the kernel traces of both forms at heavy 2 M
for c in "heavy2M tight" "heavy1M tight" "heavy2M gsplat_eager" "heavy1M gsplat_eager" "S3 tight" "longlists tight" "heavy2Mwin tight"; do
  timeout 300 python tools/rounds_time.py $c 0.0625,0.125,0.25 30 2>/dev/null | grep "^{"
done > gpurun_out/${tag}_rounds_time.jsonl
for c in "heavy2M tight" "heavy1M tight" "heavy2M gsplat" "heavy2Mwin tight" "bench1M tight"; do timeout 200 python tools/rounds_fps.py $c 60 2>/dev/null | grep "^{"; done > gpurun_out/${tag}_rounds_fps.jsonl
timeout 300 bash tools/prof_cmd.sh ${tag}_rounds_heavy2m tools/rounds_time.py heavy2M tight auto,auto 40 > /dev/null 2>&1
timeout 300 bash tools/prof_cmd.sh ${tag}_oneround_heavy2m tools/rounds_time.py heavy2M tight off,off 40 > /dev/null 2>&1
# per-Gaussian criterion: fp32 sums / fp64 sums / fp64 sums + exact exp2 and division / fp32 oracle
timeout 900 python tools/acc64_ab.py gpurun_out/${tag}_acc64_ab.json > gpurun_out/${tag}_acc64_ab.log 2>&1
head -c 600 gpurun_out/${tag}_bench_line.json; echo; tail -3 gpurun_out/${tag}_bench.err

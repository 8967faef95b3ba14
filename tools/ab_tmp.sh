for v in _old "" _coop48 _coop128 _coop256; do
  lib=$PWD/easy_gaussian_splatting_amd/libgsraster$v.so
  echo "== lib '$v'"
  GS_LIB_PATH=$lib python bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('bench', d['value'], d['ms_per_step'], d['step_ms']['median'], d['forward_fps'], {k[3:]:round(v,3) for k,v in d['stage_ms'].items()})"
  for c in heavy1M longlists; do GS_LIB_PATH=$lib python tools/config_run.py $c 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['config'], d['stage_ms']['project_bwd'])"; done
done

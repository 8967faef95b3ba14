cd $GRAFT_REPO_ROOT
for v in "" _ck2 _ck4 _ck1024; do GS_LIB_PATH=easy_gaussian_splatting_amd/libgsraster$v.so python tools/blend_time.py; done > gpurun_out/r04d_ckpt_every.txt 2>&1
python -m pytest tests/test_gpu_configs.py -q -m gpu -k "s5_4k_against" 2>&1 | tail -60 > gpurun_out/r04d_s5.log

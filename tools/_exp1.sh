cd $GRAFT_REPO_ROOT
GS_LIB_PATH=easy_gaussian_splatting_amd/libgsraster.so python tools/blend_time.py > gpurun_out/r04e_rows.txt 2>&1
GS_ROW_FLOATS=16 GS_LIB_PATH=easy_gaussian_splatting_amd/libgsraster_row16.so python tools/blend_time.py >> gpurun_out/r04e_rows.txt 2>&1
GS_ROW_FLOATS=16 tools/tune_variants.sh bench row16 >> gpurun_out/r04e_rows.txt 2>&1
GS_ROW_FLOATS=16 GS_LIB_PATH=easy_gaussian_splatting_amd/libgsraster_row16.so python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "parity_small or view" 2>&1 | tail -3 >> gpurun_out/r04e_rows.txt

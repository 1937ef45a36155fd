cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_bf16_maps_gpu.py tests/test_dconv_bf16_gpu.py tests/test_bf16_gpu.py tests/test_dconv_gpu.py tests/test_conv2d_gpu.py -x -q 2>&1 | tail -2
for r in 1 2 3; do
bash tools/dbg/ab_cfg.sh c3 --workload config3
bash tools/dbg/ab_cfg.sh c3bf --workload config3 --bf16-maps
done

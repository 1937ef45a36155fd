mkdir -p gpurun_out/r04z
python -m pytest tests/test_bf16_gpu.py tests/test_dconv_bf16_gpu.py -q -m gpu -x > gpurun_out/r04z/t_bf.log 2>&1; tail -3 gpurun_out/r04z/t_bf.log
python -m pytest tests/test_step_gpu.py -q -m gpu -x -k "bf16 or config3" > gpurun_out/r04z/t_bf2.log 2>&1; tail -2 gpurun_out/r04z/t_bf2.log
for round in 1 2; do for v in 0 1; do
  RTG_BF16_FP32_THIN=$v timeout -k 10 400 python bench.py --workload config3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('fp32-thin $v round $round', d['ms_per_step'])"
done; done

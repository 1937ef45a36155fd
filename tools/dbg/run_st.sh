mkdir -p gpurun_out/r04z
python -m pytest tests/test_conv2d_gpu.py tests/test_models_gpu.py tests/test_f4_gpu.py tests/test_loss_gpu.py -q -m gpu -x > gpurun_out/r04z/t_st.log 2>&1
tail -3 gpurun_out/r04z/t_st.log
python bench.py --no-cpu-baseline > gpurun_out/r04z/bench7.json 2> gpurun_out/r04z/bench7.err
python -c "
import json
d=json.load(open('gpurun_out/r04z/bench7.json')); print(d['ms_per_step'])
b=d['roofline']['bandwidth_kernels']; print(b['stft_fwd'], b['stft_bwd'])
"

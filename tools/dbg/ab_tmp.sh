#!/bin/bash
out=gpurun_out/r06d_ab_hwq.txt; rm -f $out
bash tools/dev_env_ab.sh $out "GPU_MAX_HW_QUEUES=4" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=6"
cat $out

mkdir -p gpurun_out/r04z
for p in 8 2 1; do
  RTG_FLAT_PASSES=$p LP_TOP=400 timeout -k 10 240 python tools/layer_profile.py config2 > gpurun_out/r04z/layers_flat$p.log 2>&1
  echo "passes $p: conv_post dgrad total ms:"
  grep "dgrad discriminators.*conv_post" gpurun_out/r04z/layers_flat$p.log | awk '{s+=$1} END {print s}'
done

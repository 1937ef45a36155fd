#!/bin/bash
# the round's profile set on the final sources, one call: config 2 (kernel stats forked + serialized, PMC, bench line), configs 3
# (both kinds of feature maps) and 4 (forked kernel stats, PMC, bench lines, serialized kernel stats), config 5 bench line
TAG=${1:-r06z}
cd $GRAFT_REPO_ROOT
log=gpurun_out/${TAG}_progress.log
echo start > $log
bash tools/profile_round.sh ${TAG} >> $log 2>&1 && echo "config2 done" >> $log
bash tools/profile_round.sh ${TAG}3 --workload config3 >> $log 2>&1 && echo "config3 done" >> $log
bash tools/profile_round.sh ${TAG}3b --workload config3 --bf16-maps >> $log 2>&1 && echo "config3 bf16 maps done" >> $log
bash tools/profile_round.sh ${TAG}4 --workload config4 >> $log 2>&1 && echo "config4 done" >> $log
timeout -k 10 300 python bench.py --workload config5 --no-cpu-baseline > gpurun_out/${TAG}_bench_config5.json 2> gpurun_out/${TAG}_bench_config5.err && echo "config5 done" >> $log
bash tools/serial_stats.sh ${TAG}2 >> $log 2>&1 && echo "serial 2 done" >> $log
bash tools/serial_stats.sh ${TAG}3 --workload config3 >> $log 2>&1 && echo "serial 3 done" >> $log
bash tools/serial_stats.sh ${TAG}4 --workload config4 >> $log 2>&1 && echo "serial 4 done" >> $log
tail -3 $log

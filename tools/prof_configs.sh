# per-UNet-call kernel tables of configs 2, 3, 5 at full size: rocprofv3 --kernel-trace --stats with 2 and 6 eager calls, differenced
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in ${CONFIGS:-cifar church sd}; do
  for n in 2 6; do
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rc_${c}_$n -- python3 $R/tools/config_bench.py $c --calls $n > /tmp/rc_${c}_$n.log 2>&1
  done
  a=$(ls /tmp/rc_${c}_2/*/*kernel_stats.csv | head -1); b=$(ls /tmp/rc_${c}_6/*/*kernel_stats.csv | head -1)
  python3 $R/tools/prof_diff.py $a $b 4 > $R/gpurun_out/${TAG:-r03a}_${c}_call_kernels.txt
done

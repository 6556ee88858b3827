set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in 1 3; do
  CALLS=$n rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rd_$n -- python3 $R/tools/decoder_prof.py > /tmp/rd_$n.log 2>&1
done
a=$(ls /tmp/rd_1/*/*kernel_stats.csv | head -1); b=$(ls /tmp/rd_3/*/*kernel_stats.csv | head -1)
python3 $R/tools/prof_diff.py $a $b 2 > $R/gpurun_out/${TAG:-r03a}_decoder16_kernels.txt

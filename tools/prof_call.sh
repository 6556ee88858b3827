# per-UNet-call kernel table of the headline config: TAG=r04a bash tools/prof_call.sh  (writes gpurun_out/${TAG}_unet_call_kernels.txt)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${TAG:-r04a}
O=$R/gpurun_out
stats() { ls $1/*/*kernel_stats.csv | head -1; }
for n in 2 12; do
  N_CALLS=$n rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/u_$n -- python3 $R/tools/unet_prof.py > /tmp/u_$n.log 2>&1
done
python3 $R/tools/prof_diff.py $(stats /tmp/u_2) $(stats /tmp/u_12) 10 > $O/${T}_unet_call_kernels.txt
cp $(stats /tmp/u_12) $O/${T}_unet14_kernel_stats.csv
cat $O/${T}_unet_call_kernels.txt

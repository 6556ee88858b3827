# per-call kernel table of the FP reference forward (TDAC / activation caching): bash tools/prof_fp.sh -> gpurun_out/${TAG}_fp_forward_kernels.txt
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${TAG:-r04z}
stats() { ls $1/*/*kernel_stats.csv | head -1; }
for n in 2 6; do
  N_CALLS=$n rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fp_$n -- python3 $R/tools/fp_fwd_prof.py > /tmp/fp_$n.log 2>&1
done
python3 $R/tools/prof_diff.py $(stats /tmp/fp_2) $(stats /tmp/fp_6) 4 > $R/gpurun_out/${T}_fp_forward_kernels.txt
tail -1 /tmp/fp_6.log

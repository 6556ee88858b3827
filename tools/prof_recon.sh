set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for u in "tf 384" "up 384"; do
  tag=$(echo $u | tr ' ' '_')
  for it in 10 40; do
    UNITS="$u" ITERS=$it rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_${tag}_$it -- python3 $R/tools/recon_prof.py > /tmp/rp_${tag}_$it.log 2>&1
  done
  a=$(ls /tmp/rp_${tag}_10/*/*kernel_stats.csv | head -1); b=$(ls /tmp/rp_${tag}_40/*/*kernel_stats.csv | head -1)
  python3 $R/tools/prof_diff.py $a $b 30 > $R/gpurun_out/recon_iter_${tag}.txt
  tail -2 /tmp/rp_${tag}_40.log
done

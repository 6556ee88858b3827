# Every rocprofv3 artefact of a round in one GPU call:  TAG=r03z bash tools/prof_round.sh   (writes gpurun_out/$TAG_*)
# counters (--pmc) are collected in their own runs with --kernel-trace only, as the pool requires
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${TAG:-r03z}
O=$R/gpurun_out
stats() { ls $1/*/*kernel_stats.csv | head -1; }
# 1. per-UNet-call kernel table of the headline config (4 and 14 eager calls as a DDIM step issues them)
for n in 2 12; do
  N_CALLS=$n rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/u_$n -- python3 $R/tools/unet_prof.py > /tmp/u_$n.log 2>&1
done
python3 $R/tools/prof_diff.py $(stats /tmp/u_2) $(stats /tmp/u_12) 10 > $O/${T}_unet_call_kernels.txt
cp $(stats /tmp/u_12) $O/${T}_unet14_kernel_stats.csv
# 2. HBM traffic of the int8 GEMM launches (separate PMC passes)
N_CALLS=2 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- python3 $R/tools/unet_prof.py > /tmp/pf.log 2>&1
N_CALLS=2 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -- python3 $R/tools/unet_prof.py > /tmp/pw.log 2>&1
python3 $R/tools/pmc_traffic.py /tmp/pf /tmp/pw $O/${T}_gemm_traffic.json 4 > $O/${T}_gemm_traffic.log 2>&1 || true
# 3. per-layer GEMM table
python3 $R/tools/gemm_table.py > $O/${T}_gemm_table.txt 2>/dev/null || true
# 4. configs 2, 3, 5 at full size: bench lines + per-call kernel tables
python3 $R/tools/config_bench.py cifar church sd --batches 2 2>/dev/null | grep '^{' > $O/${T}_configs_2_3_5.jsonl || true
TAG=$T bash $R/tools/prof_configs.sh
# 5. one reconstruction iteration of a transformer block and a ResBlock; the first-stage decoder
bash $R/tools/prof_recon.sh > /dev/null 2>&1
mv $O/recon_iter_tf_384.txt $O/${T}_recon_iter_tf384at32.txt; mv $O/recon_iter_up_384.txt $O/${T}_recon_iter_res384to192at64.txt
TAG=$T bash $R/tools/prof_decoder.sh
# 6. kernel statistics of the bench command itself
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/b -- python3 $R/bench.py --steps 3 --warmup 1 --calib none --no-cpu-baseline > $O/${T}_bench_line_profiled.json 2> /tmp/b.log
cp $(stats /tmp/b) $O/${T}_bench_kernel_stats.csv
ls -la $O | grep $T

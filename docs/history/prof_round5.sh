# Round-5 rocprofv3 artefacts in one GPU call:  TAG=r05z bash tools/prof_round5.sh   (writes gpurun_out/$TAG_*; copy what is judged to profiles/)
# counters (--pmc) are collected in their own runs with --kernel-trace only, as the pool requires
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${TAG:-r05z}
O=$R/gpurun_out
stats() { ls $1/*/*kernel_stats.csv | head -1; }
# 1. per-UNet-call kernel table of the headline config (4 and 14 eager calls as a DDIM step issues them)
for n in 2 12; do
  N_CALLS=$n rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/u_$n -- python3 $R/tools/unet_prof.py > /tmp/u_$n.log 2>&1
done
python3 $R/tools/prof_diff.py $(stats /tmp/u_2) $(stats /tmp/u_12) 10 > $O/${T}_unet_call_kernels.txt
cp $(stats /tmp/u_12) $O/${T}_unet14_kernel_stats.csv
# 2. HBM traffic of the int8 GEMM launches, per kernel against the engine's launch list (separate PMC passes)
TAG=$T bash $R/tools/prof_traffic.sh > /dev/null 2>&1 || true
# 3. MFMA utilisation of the dominant kernels of one UNet call (SQ counters, one pass)
N_CALLS=2 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pm -- python3 $R/tools/unet_prof.py > /tmp/pm.log 2>&1
python3 $R/tools/pmc_mfma.py /tmp/pm > $O/${T}_unet_call_mfma_util.txt 2>&1 || true
# 4. per-layer GEMM table
python3 $R/tools/gemm_table.py > $O/${T}_gemm_table.txt 2>/dev/null || true
# 5. one reconstruction iteration of the two profiled units, the decoder, configs 2 / 3 / 5
TAG=$T bash $R/tools/prof_recon.sh > /dev/null 2>&1 || true
TAG=$T bash $R/tools/prof_decoder.sh > /dev/null 2>&1 || true
TAG=$T bash $R/tools/prof_configs.sh > /dev/null 2>&1 || true
# 6. kernel statistics of the bench command itself
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/b -- python3 $R/bench.py --steps 3 --warmup 1 --calib none --no-cpu-baseline --no-configs > $O/${T}_bench_line_profiled.json 2> /tmp/b.log
cp $(stats /tmp/b) $O/${T}_bench_kernel_stats.csv
ls -la $O | grep $T

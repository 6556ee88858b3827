# HBM traffic of the int8 GEMM launches of one UNet call (separate PMC passes), per kernel against its own algorithmic bytes:
#   TAG=r05a bash tools/prof_traffic.sh
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${TAG:-r05a}
O=$R/gpurun_out
export LAUNCH_LIST=/tmp/launch_list.json
N_CALLS=4 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- python3 $R/tools/unet_prof.py > /tmp/pf.log 2>&1
N_CALLS=4 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -- python3 $R/tools/unet_prof.py > /tmp/pw.log 2>&1
cp /tmp/launch_list.json $O/${T}_launch_list.json
python3 $R/tools/pmc_traffic.py /tmp/pf /tmp/pw $O/${T}_gemm_traffic.json /tmp/launch_list.json 4 | tee $O/${T}_gemm_traffic.log

# HBM evidence for the HBM-bound kernels (K1, K2, K5, K7, K8, K9, K10, K12): for each of two workloads -- an eager DDIM sample of one
# batch (tools/sample_prof.py) and reconstruction iterations of three production-size units (tools/recon_prof.py) -- three passes of
# the SAME command: rocprofv3 --kernel-trace --stats (durations), --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate counter passes, as
# MI355X_MICROARCH.md prescribes), with the algorithmic bytes accounted at the C ABI (EDADM_TRACE_BYTES).  tools/elementwise_hbm.py
# joins them into profiles/<TAG>_elementwise_hbm.txt:   TAG=r06 bash tools/prof_elementwise.sh
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${TAG:-r06}
O=$R/gpurun_out
for W in sample recon; do
  if [ $W = sample ]; then CMD="python3 $R/tools/sample_prof.py"; else export ITERS=12 UNITS="res 192@64,tf 384@32,res 960@8"; CMD="python3 $R/tools/recon_prof.py"; fi
  rm -rf /tmp/pe_${W}_t /tmp/pe_${W}_f /tmp/pe_${W}_w
  EDADM_TRACE_BYTES=/tmp/pe_${W}_bytes.json rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe_${W}_t -- $CMD > /tmp/pe_${W}_t.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pe_${W}_f -- $CMD > /tmp/pe_${W}_f.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pe_${W}_w -- $CMD > /tmp/pe_${W}_w.log 2>&1
  python3 $R/tools/elementwise_hbm.py $W /tmp/pe_${W}_t /tmp/pe_${W}_f /tmp/pe_${W}_w /tmp/pe_${W}_bytes.json $O/${T}_elementwise_hbm_${W}.txt
done
cat $O/${T}_elementwise_hbm_sample.txt $O/${T}_elementwise_hbm_recon.txt > $O/${T}_elementwise_hbm.txt

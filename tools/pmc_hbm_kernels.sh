# Counter evidence for the HBM-bound kernels (sampler fwd / bwd / dbwd, marching cubes, interp2x, 3x3 inverse): separate rocprofv3 --pmc
# passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; L2 hit / miss in a third) over tools/kernel_only.py run_pmc, plus the kernel
# trace of the same cases for the durations.   bash tools/pmc_hbm_kernels.sh r04   -> gpurun_out/r04_pmc_hbm_kernels.txt
R=${1:-r04}
REPO=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $REPO/gpurun_out
export TMPDIR=/tmp
cd /tmp
dirs=""
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $grp | tr ' ' '_')
  rm -rf /tmp/pk_$n
  timeout 240 rocprofv3 --pmc $grp --output-format csv -d /tmp/pk_$n -o run -- python $REPO/tools/kernel_only.py run_pmc /tmp/pk_cases.json > /tmp/pk_$n.log 2>&1 || echo "counter pass failed: $grp"
  dirs="$dirs /tmp/pk_$n"
done
rm -rf /tmp/pk_trace
timeout 240 rocprofv3 --kernel-trace -d /tmp/pk_trace -o run -- python $REPO/tools/kernel_only.py run_pmc /tmp/pk_cases_t.json > /tmp/pk_trace.log 2>&1
cd $REPO
(python tools/kernel_only.py report_pmc /tmp/pk_cases.json $dirs; python tools/kernel_only.py report /tmp/pk_trace /tmp/pk_cases_t.json) > gpurun_out/${R}_pmc_hbm_kernels.txt 2>&1
tail -5 gpurun_out/${R}_pmc_hbm_kernels.txt

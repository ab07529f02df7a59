# Counters of the row-tile MLP kernels (rocprofv3 --pmc, separate passes):  bash tools/pmc_mlp_rows.sh > gpurun_out/r04_pmc_mlp_rows.txt
export TMPDIR=/tmp; cd /tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for rt in 1 2; do
  i=0
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU"; do
    i=$((i+1))
    rm -rf /tmp/pr_${rt}_$i
    timeout 120 rocprofv3 --pmc $grp --output-format csv -d /tmp/pr_${rt}_$i -o run -- python $R/tools/mlp_rows_one.py 4096 $rt 4 > /tmp/pr_${rt}_$i.log 2>&1 || echo "group failed: rt=$rt $grp"
  done
done
python - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob('/tmp/pr_*/**/*counter_collection.csv', recursive=True):
    rt = f.split('/tmp/pr_')[1][0]
    for row in csv.DictReader(open(f, newline='')):
        if 'mlp_rows' in row.get('Kernel_Name', ''):
            k = 'fwd' if 'fwd' in row['Kernel_Name'] else 'vjp'
            acc[(rt, k, row['Counter_Name'])].append(float(row['Counter_Value']))
for (rt, k, c), v in sorted(acc.items()):
    print(f"row_tiles={rt} {k} {c:32s} n={len(v):2d} mean={sum(v)/len(v):16.1f}")
PY

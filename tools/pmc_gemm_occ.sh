# Matrix-pipe occupancy of the large f32 product, high-occupancy kernel against the one it replaced (rocprofv3 --pmc, separate passes):
#   bash tools/pmc_gemm_occ.sh > gpurun_out/pmc_gemm_occ.txt
export TMPDIR=/tmp; cd /tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for occ in 1 0; do
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU"; do
    n=$(echo ${occ}_$grp | tr ' ' '_' | cut -c1-40)
    RECMV_GEMM_OCC=$occ timeout 120 rocprofv3 --pmc $grp --output-format csv -d /tmp/po_$n -o run -- python $R/tools/gemm_one.py 460800 512 512 0 2 5 > /tmp/po_$n.log 2>&1 || echo "group failed: $occ $grp"
  done
done
python - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob('/tmp/po_*/**/*counter_collection.csv', recursive=True):
    occ = f.split('/tmp/po_')[1][0]
    for row in csv.DictReader(open(f, newline='')):
        if 'gemm_nt' in row.get('Kernel_Name', ''):
            k = row['Kernel_Name'].split('(anonymous namespace)::')[-1].split('(')[0]
            acc[(occ, k, row['Counter_Name'])].append(float(row['Counter_Value']))
res = {}
for (occ, k, c), v in sorted(acc.items()):
    res[(occ, k, c)] = sum(v) / len(v)
    print(f"RECMV_GEMM_OCC={occ} {k:48s} {c:28s} n={len(v):2d} mean={sum(v)/len(v):16.1f}")
for occ in ('1', '0'):
    ks = {k for (o, k, c) in res if o == occ}
    for k in ks:
        b, g = res.get((occ, k, 'SQ_VALU_MFMA_BUSY_CYCLES')), res.get((occ, k, 'GRBM_GUI_ACTIVE'))
        if b and g:
            print(f"RECMV_GEMM_OCC={occ} {k}: matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs) = {b / 1024 / (g / 8) * 100:.1f} %")
PY

# Run-to-run reproducibility of the loop on one MI355X box: N runs of tools/determinism_probe.py per configuration, digests counted.
#   bash tools/loop_repro.sh [runs=8] [iterations=1]
# A configuration whose runs are bit-identical prints ONE digest with count N.  Round 4 (profiles/r04_b3_presplit.txt): f32 mode 8 of 8
# identical in every configuration tried; bf16x6 mode with the second garment's render-loss chain on the side stream 2-3 of 8-10 runs part.
N=${1:-8}; IT=${2:-1}
run() {  # label, then env assignments
  label=$1; shift
  for i in $(seq $N); do env "$@" timeout 60 python tools/determinism_probe.py $IT 2>/dev/null | md5sum | cut -c1-6; done | sort | uniq -c | tr "\n" " "
  echo " <- $label"
}
run "f32, default switches" RECMV_GEMM_MODE=0
run "f32, per-garment implicit differentiation" RECMV_GEMM_MODE=0 RECMV_PROP_JOINT=0
run "bf16x6, default switches (no render side stream in this mode)" RECMV_GEMM_MODE=1
run "bf16x6, render side stream ON" RECMV_GEMM_MODE=1 RECMV_RENDER_STREAMS=1
run "bf16x6, render side stream ON, per-garment implicit differentiation" RECMV_GEMM_MODE=1 RECMV_RENDER_STREAMS=1 RECMV_PROP_JOINT=0

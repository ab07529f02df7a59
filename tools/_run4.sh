set -x
O=gpurun_out/r06d; mkdir -p $O
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-config2 --no-mc --no-hbm-kernels --no-serial-pass"
for i in 1 2; do
python bench.py $Q > $O/bench_rows3328_$i.json 2> $O/bench_rows3328_$i.log
RECMV_MLP_ROWS_MIN=2048 python bench.py $Q > $O/bench_rows2048_$i.json 2> $O/bench_rows2048_$i.log
RECMV_MLP_ROWS_MIN=1024 python bench.py $Q > $O/bench_rows1024_$i.json 2> $O/bench_rows1024_$i.log
done
RECMV_MLP_ROWS_MIN=2048 python tools/phase_overlap.py > $O/phase_overlap_rows2048.txt 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mc --no-hbm-kernels --no-serial-pass > $O/bench_config2.json 2> $O/bench_config2.log

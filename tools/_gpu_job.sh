set -x
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > gpurun_out/r2_gputests.log
timeout 200 python tools/seg3d_timing.py > gpurun_out/r2_seg3d.log 2>&1
rm -rf /tmp/run1; mkdir -p /tmp/run1
timeout 300 python rec-mv_amd/train.py --conf configs/synthetic/people_snapshot_like.conf --data /tmp/run1 --save-folder out --frames 12 --max-iters 5 > gpurun_out/r2_train_smoke.log 2>&1
ls -la /tmp/run1/out >> gpurun_out/r2_train_smoke.log
cp /tmp/run1/out/latest.pth /tmp/run1/out/a-pose.pth
timeout 300 python rec-mv_amd/train_large_pose.py --conf configs/synthetic/people_snapshot_like.conf --data /tmp/run1 --save-folder out --frames 12 --max-iters 4 --project_name p --exp_name e --data_type large_pose > gpurun_out/r2_train_large_smoke.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 1 --no-cpu-baseline > gpurun_out/r2_bench2.json 2> gpurun_out/r2_bench2.err

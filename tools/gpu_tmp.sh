mkdir -p gpurun_out/a1
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/a1/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-utts 4 --no-f32-leg 2>&1 | tail -1 | tee gpurun_out/a1/c2.json

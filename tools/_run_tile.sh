cd /root/repo
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 100 python tools/fuzz.py 60 5151 2>&1 | tail -1
EVS_FUSED_TILE_MIN_B=1 timeout 130 python tools/fuzz.py 90 5152 2>&1 | tail -1
timeout 300 python bench.py > gpurun_out/bench_tile.json 2> gpurun_out/bench_tile.err; tail -1 gpurun_out/bench_tile.json | cut -c1-1500
bash tools/prof_bench.sh prof_tile

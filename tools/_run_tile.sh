cd /root/repo
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
EVS_FUSED_TILE_MIN_B=1 timeout 100 python tools/fuzz.py 60 6161 2>&1 | tail -1
for tm in 0 1; do
  echo "== tile=$tm"
  EVS_FUSED_TILE=$tm timeout 120 python tools/kbench.py --fused-only --batch 4096 8192 16384 32768 65536 131072 --iters 300 2>&1 | grep "one index"
done
timeout 300 python bench.py --no-extras --no-cpu-baseline --no-cache-tier 2>/dev/null | tail -1 | cut -c1-400

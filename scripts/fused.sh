for f in 0 1 0 1; do echo "== FUSED_TRACE=$f"; RGPU_FUSED_TRACE=$f python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
python scripts/gpu_probe.py 2>&1 | grep "TOTAL mism"

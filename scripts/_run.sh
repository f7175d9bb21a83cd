mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/t5.log 2>&1; tail -3 gpurun_out/t5.log
timeout 600 python bench.py > gpurun_out/bench5.json 2> gpurun_out/bench5.err; tail -c 600 gpurun_out/bench5.err; head -c 3000 gpurun_out/bench5.json
timeout 300 python bench.py --mode random --steps 5 2>/dev/null | head -c 1200

mkdir -p gpurun_out/r05t
timeout 120 python3 -X faulthandler bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-others --no-extras --no-preheat --batch 32 2>&1 | grep -v "Extension modules" | tail -30 | cut -c1-200 > gpurun_out/r05t/dbg2.txt
echo ---- eager >> gpurun_out/r05t/dbg2.txt
UNIVL_AB=auto_graph=0 timeout 120 python3 -X faulthandler bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-others --no-extras --no-preheat --batch 32 --no-graph 2>&1 | grep -v "Extension modules" | tail -30 | cut -c1-300 >> gpurun_out/r05t/dbg2.txt
cat gpurun_out/r05t/dbg2.txt

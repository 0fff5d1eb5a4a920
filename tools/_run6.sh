rm -f gpurun_out/r05_wgrad9_*.txt
bash tools/wgrad9_bench.sh > /dev/null 2>&1
tail -3 gpurun_out/r05_wgrad9_test.txt
tail -44 gpurun_out/r05_wgrad9_bench.txt

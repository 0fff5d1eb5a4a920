python -m pytest tests/test_gpu_conv_variants.py -q -x --tb=line -k "eight_phase" > gpurun_out/r05_i8_tests.txt 2>&1; tail -2 gpurun_out/r05_i8_tests.txt
cd /tmp && export TMPDIR=/tmp
for v in tap slice; do
  rm -rf /tmp/pmcF_$v /tmp/pmcW_$v
  if [ $v = slice ]; then export MTE_LIB_PATH=$GRAFT_REPO_ROOT/ab_lib/libmte_hip_slicemajor.so; else unset MTE_LIB_PATH; fi
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmcF_$v -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmcW_$v -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
  echo "== K order of the 8-phase kernels: $v-major" >> $GRAFT_REPO_ROOT/gpurun_out/r05_i8_korder_pmc.txt
  python3 $GRAFT_REPO_ROOT/tools/pmc_traffic.py /tmp/pmcF_$v /tmp/pmcW_$v 6 | head -18 >> $GRAFT_REPO_ROOT/gpurun_out/r05_i8_korder_pmc.txt
done
unset MTE_LIB_PATH
cat $GRAFT_REPO_ROOT/gpurun_out/r05_i8_korder_pmc.txt

for i in 1 2 3; do for lib in a_w128_p256 b_w128_p192 c_w128_p128 d_w128_p256_wide128 e_w128_p256_p3w384; do
MTE_LIB_PATH=$PWD/ab_lib/libmte_$lib.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'])"
done; done

for d in 0 2 1; do
MLHOT_BENCH_KERNELS=gpurun_out/k$d.json python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dbg $d 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_us_per_step']; print($d, d['ms_per_step'], {x:k[x] for x in k if 'conv12' in x})"
done

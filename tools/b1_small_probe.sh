for f in 64 32 16 0; do
  MVSGI_B3_SMALL=$f python bench.py --batch 1 --steps 200 --warmup 20 --no-extras --no-cpu-baseline | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('force=$f', d['ms_per_step'], {k[21:45]:(v['launches']//200, v['avg_us']) for k,v in d['kernels'].items() if ' 1, 4, 16, 1, 3, false, false, false' in k})"
done

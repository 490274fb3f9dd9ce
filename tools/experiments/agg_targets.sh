# small-batch aggregation by targets (default) against the work list by molecule parts: step time at bs = 128 / 256
for set in C B; do for m in 128 256; do for t in 512 0; do
  echo "set $set mols $m GEOSSL_AGG_TARGETS_MAX=$t:"; GEOSSL_AGG_TARGETS_MAX=$t python bench.py --mols $m --set $set --cutoff 10 --steps 60 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %.4f ms/step' % d['ms_per_step'])"
done; done; done
for m in 128 256; do for t in 512 0; do
  echo "set B mols $m NO_RAGGED_LOOP GEOSSL_AGG_TARGETS_MAX=$t:"; GEOSSL_NO_RAGGED_LOOP=1 GEOSSL_AGG_TARGETS_MAX=$t python bench.py --mols $m --set B --steps 60 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %.4f ms/step' % d['ms_per_step'])"
done; done

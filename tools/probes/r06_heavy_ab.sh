# experiments build: the one-key sweep with and without the heavy-partition extras (DBHIP_GBC_HEAVY=0), same box
mkdir -p gpurun_out
for rep in 1 2; do for h in 1 0; do
  DBHIP_GBC_HEAVY=$h timeout 300 python tools/microbench.py --only groupby --gb-card 5000,20000,100000,1000000 --out gpurun_out/r06_heavy_${h}_$rep.json > /dev/null 2>&1
done; done
python - <<'PY'
import json
def rows(p): return {r['name']: r['ms_best'] for r in json.load(open(p)) if 'add_block i64 key' in r['name']}
t={(h,rep): rows(f'gpurun_out/r06_heavy_{h}_{rep}.json') for h in (1,0) for rep in (1,2)}
for n in t[(1,1)]: print('%-60s heavy on %.3f %.3f   off %.3f %.3f' % (n, t[(1,1)][n], t[(1,2)][n], t[(0,1)][n], t[(0,2)][n]))
PY

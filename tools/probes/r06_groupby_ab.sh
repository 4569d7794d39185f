# the one-key cardinality sweep of the round-4 library (git worktree ab_r04 = 65149d0, built beside HEAD) and of HEAD on the SAME box,
# twice each, interleaved (r04, HEAD, r04, HEAD)
R=$PWD; mkdir -p gpurun_out
for rep in 1 2; do
  for v in ab_r04 head; do
    d=$R; [ $v = ab_r04 ] && d=$R/ab_r04
    (cd $d && DBHIP_JIT_CACHE_DIR=/tmp/jit_$v timeout 600 python tools/microbench.py --only groupby --out $R/gpurun_out/r06_gbab_${v}_$rep.json > /dev/null 2>&1)
  done
done
python - <<'PY'
import json
def rows(p):
    try: return {r['name']: r['ms_best'] for r in json.load(open(p)) if 'add_block i64 key' in r['name']}
    except Exception as e: return {'ERR '+str(e): 0}
t={(v,rep): rows(f'gpurun_out/r06_gbab_{v}_{rep}.json') for v in ('ab_r04','head') for rep in (1,2)}
names=list(t[('head',1)].keys())
print('%-60s %8s %8s %8s %8s' % ('case','r04 #1','HEAD #1','r04 #2','HEAD #2'))
for n in names:
    print('%-60s %8.3f %8.3f %8.3f %8.3f' % (n, t[('ab_r04',1)].get(n,-1), t[('head',1)].get(n,-1), t[('ab_r04',2)].get(n,-1), t[('head',2)].get(n,-1)))
PY

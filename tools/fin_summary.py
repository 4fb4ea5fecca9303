import json, sys
for l in open(sys.argv[1]):
    try: d=json.loads(l)
    except Exception: print(l[:300]); continue
    print('var',d['var'],'bad',d['bad_runs'],'/',d['runs'], {k:d[k] for k in ('checks','bad_after_barrier','bad_at_end') if k in d})
    for x in d.get('detail',[])[:int(sys.argv[2]) if len(sys.argv)>2 else 0]:
        print('  slice',x['slice'],'n',x['n'],'unique',x['unique'],'cdiffs',x['coeff_diffs'],'rdiffs',x['recon_diffs'])
        for r in x['rows[idx,leaf,weight,step,got,want]'][:12]: print('    ',r)

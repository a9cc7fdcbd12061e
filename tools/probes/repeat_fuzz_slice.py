import sys, json, os
sys.argv = [sys.argv[0]]
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+'/tools'); sys.path.insert(0, ROOT+'/tests')
import fuzz_parity as fz
n_bad = 0
for rep in range(14):
    for seed in range(0, 20):
        try:
            r = fz.run_case(seed)
        except Exception as ex:
            r = dict(seed=seed, ok=False, error=repr(ex)[:600])
        if not r["ok"]:
            n_bad += 1
            print(json.dumps(dict(rep=rep, **{k: r[k] for k in r if k in ("seed","bad","error","worst","worst_key","n_cams","n_points","n_obs","shape","shared","prior_rows","locked","max_track")})), flush=True)
print("DONE", n_bad, flush=True)

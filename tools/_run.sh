python - <<'PY'
import subprocess, json, os
r = {}
for i in range(2):
    for v in ("512", "256", "1024", "4096", "1000000"):
        env = dict(os.environ, S2F_PG_G2_MAXWG=v)
        out = subprocess.run(["python","bench.py","--steps","20","--warmup","4","--no-cpu-baseline","--no-kernel-events"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        r.setdefault(v, []).append(json.loads(out)["ms_per_step"])
for v in r: print("G2 max WGs =", v, r[v], "mean %.3f" % (sum(r[v]) / len(r[v])))
PY

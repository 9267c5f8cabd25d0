python - <<'PY'
import subprocess, json, os
r = {}
for i in range(3):
    for v in ("1", "2", "0"):
        env = dict(os.environ, S2F_CONV3_DX_PIPE=v)
        out = subprocess.run(["python","bench.py","--steps","20","--warmup","4","--no-cpu-baseline","--no-kernel-events"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        r.setdefault(v, []).append(json.loads(out)["ms_per_step"])
for v in r: print("conv3 dx pipe =", v, r[v], "mean %.3f" % (sum(r[v]) / len(r[v])))
PY

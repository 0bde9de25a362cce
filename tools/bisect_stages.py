"""One-off measurement aid: per-stage kernel time with early-return points (build with -DVGPMP_BISECT)."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
env0 = dict(os.environ, VGPMP_HIP_LIB=os.path.join(root, "tools", "libvgpmp_bisect.so"))
def run(extra):
    env = dict(env0, **extra)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "20", "--warmup", "5", "--unroll", "0",
                          "--no-cpu-baseline", "--profile-steps", "30", "--allow-nan"], env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(out.stderr[-2000:]); raise SystemExit(1)
    return json.loads(line[-1])["stage_ms"]
import ast
SETS = ast.literal_eval(os.environ.get("BISECT_SETS", "[('VGPMP_STOP_FINAL', 'final_adam', [7, 5, 6, 1, 2])]"))
for var, stage, stops in SETS:
    base = run({})[stage]
    print(stage, "full", round(base * 1e3, 1), "us")
    for k in stops:
        print("  stop", k, round(run({var: str(k)})[stage] * 1e3, 1), "us")

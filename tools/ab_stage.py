#!/usr/bin/env python
"""Measurement aid: interleaved bench.py runs over variant libraries tools/libvgpmp_<name>.so ("product" = the built one), printing
us per step and the event-timed stages.   tools/ab_stage.py "product base" [rounds] -- [bench.py arguments]"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = sys.argv[1].split()
rounds = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] != "--" else 2
extra = sys.argv[sys.argv.index("--") + 1:] if "--" in sys.argv else []
for i in range(rounds):
    for v in names:
        env = dict(os.environ)
        env.pop("VGPMP_HIP_LIB", None)
        if v != "product":
            env["VGPMP_HIP_LIB"] = os.path.join(root, "tools", f"libvgpmp_{v}.so")
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-solve", "--min-seconds", "1",
                              "--allow-nan"] + extra, capture_output=True, text=True, env=env)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            st = d["stage_ms"]
            print(f"{v:>16s} {1e3 * d['ms_per_step']:8.2f} us/step | " + " ".join(f"{k}={1e3 * x:.1f}" for k, x in st.items()), flush=True)
        except Exception as e:
            print(v, "failed", e, out.stderr[-300:], flush=True)

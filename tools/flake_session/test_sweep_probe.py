import os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_hand_reduced_sweep_across_preemption():
    """tools/sweep_probe (SWEEP_PROBE=sweep_probe | sweep_probe_fenced: the hand-reduced second sweep, plain HIP, no library) beside the
    starting rank processes of this session."""
    exe = os.path.join(ROOT, "tools", os.environ.get("SWEEP_PROBE", "sweep_probe"))
    res = subprocess.run([exe, os.environ.get("FLAKE_SECONDS", "12"), os.environ.get("PROBE_REP", "40"), os.environ.get("PROBE_WGS", "4096")],
                         capture_output=True, text=True, timeout=120)
    print("\n" + res.stdout[-3000:] + res.stderr[-1000:], flush=True)
    assert res.returncode == 0

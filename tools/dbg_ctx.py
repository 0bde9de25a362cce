"""Measurement aid: the tests that precede the merged-vs-unmerged comparison in the flaky pytest selection, called directly, then the comparison
with a check after every call.  Loop from the shell."""
import sys, runpy
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_gpu_surface as T
for P, M in [(1, 12), (3, 12), (1, 30), (2, 30), (1, 20), (6, 12), (30, 12), (40, 30), (112, 30), (110, 20)]:
    T.test_pipelined_steps_equal_single_step_calls(P, M)
T.test_pipelined_steps_with_trainable_likelihood_constants()
for (S, N, M, P) in [(128, 100, 30, 64), (70, 20, 5, 63), (37, 50, 10, 40)]:
    T.test_reverse_path_pass_over_several_chunks_per_workgroup_is_bitwise_the_same(S, N, M, P)
sys.argv = ["dbg_fresh.py", sys.argv[1] if len(sys.argv) > 1 else "ab"]
runpy.run_path("/root/repo/tools/dbg_fresh.py", run_name="__main__")

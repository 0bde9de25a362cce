import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_gpu_surface as T
mode = sys.argv[1] if len(sys.argv) > 1 else "all"
if mode in ("all", "pipe"):
    for P, M in [(1, 12), (3, 12), (1, 30), (2, 30), (1, 20), (6, 12), (30, 12), (40, 30), (112, 30), (110, 20)]:
        T.test_pipelined_steps_equal_single_step_calls(P, M)
    for P in (4, 8):
        pass
    T.test_pipelined_steps_with_trainable_likelihood_constants()
if mode in ("all", "chunks"):
    for (S, N, M, P) in [(128, 100, 30, 64), (70, 20, 5, 63), (37, 50, 10, 40)]:
        T.test_reverse_path_pass_over_several_chunks_per_workgroup_is_bitwise_the_same(S, N, M, P)
try:
    T.test_merged_launches_of_the_batch_schedule_change_nothing(64, 30, 40, 12)
    print("same")
except AssertionError as e:
    print("DIFF", str(e)[:80])

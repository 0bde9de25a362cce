"""utils/miscellaneous.py of the reference: star-imported by its driver (benchmarking.py:3)."""
from vgpmp_amd.host.miscellaneous import *  # noqa: F401,F403
from vgpmp_amd.host.miscellaneous import (gpflow, np, os, p, sys, time, get_root_package_path, init_trainset,  # noqa: F401
                                          disable_param_opt, optimization_step, training_loop, solve_planning_problem,
                                          solve_planning_problems_batched)

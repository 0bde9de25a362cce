# A/B of the likelihood kernel forms: BASELINE config 5 share on a cache-resident (64^3) and an HBM-resident (512^3) table,
# and 64 Franka problems.  Run on the GPU box from the repo root: bash tools/ab_lik.sh [forms...]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
FORMS=${@:-"lanes lanes-lds"}
S="--workload stress --problems 64 --steps 10 --warmup 3 --no-cpu-baseline --no-solve --profile-steps 10"
for g in 64 512; do for form in $FORMS; do for sm in off on; do
python bench.py $S --grid $g --lik-form $form --summary $sm 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read()); r=b['roofline']
print('stress grid $g form $form summary $sm:', round(1e3*r['avg_launch_ms'],1),'us  step',round(b['ms_per_step'],3))"
done; done; done
F="--problems 64 --scene synthetic --steps 10 --warmup 3 --no-cpu-baseline --no-solve --profile-steps 10"
for form in $FORMS; do
python bench.py $F --lik-form $form 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read()); r=b['roofline']
print('franka x64 form $form:', round(1e3*r['avg_launch_ms'],1),'us  step',round(b['ms_per_step'],3))"
done

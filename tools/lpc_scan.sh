# crossover between the few-problem (8 lanes per configuration) and the batch form of the likelihood: ms per step by problem count
cd $GRAFT_REPO_ROOT
for p in 1 2 3 4 5 6 8 12; do for form in auto lanes; do
python bench.py --problems $p --steps 50 --warmup 10 --no-cpu-baseline --no-solve --profile-steps 10 --lik-form $form --min-seconds 0.2 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read()); r=b['roofline']
print('problems $p form $form:', r['kernel'][:28], round(1e3*r['avg_launch_ms'],1),'us lik;  step',round(1e3*b['ms_per_step'],1),'us')"
done; done

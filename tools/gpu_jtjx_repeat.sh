export CERES_HIP_PROBLEM_CACHE=/tmp/ceres_problem_cache
timeout 300 python tools/kernel_times.py venice1778 > /dev/null 2>&1
for R in 1 2 3 4; do
  timeout 300 python tools/kernel_times.py venice1778 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('jtjx_ms','jtjx_frac','sx_ms','sx_frac','schur_init_ms','schur_jacobi_ms','back_substitute_ms','cgnr_setup_ms')})"
done
timeout 600 python bench.py --gpus 1 --solver cgnr --both-solvers 0 --minimizer-iterations 0 --host-boundary-steps 0 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cgnr bench', d['ms_per_step'], d['roofline'])"

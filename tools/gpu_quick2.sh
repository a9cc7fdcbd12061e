cd /root/repo
timeout 600 python tools/kernel_times.py venice1778 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:v for k,v in d.items() if k.endswith('_ms')}); print(d['cgnr_solve']); print(d['schur_solve'])"
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -5
for S in cgnr iterative_schur; do timeout 600 python bench.py --no-cpu-baseline --both-solvers 0 --minimizer-iterations 0 --solver $S 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['solver'], 'steps/s', d['value'], 'ms', d['ms_per_step'], d['roofline']['frac'], d['extra']['solve_phases_ms'])"; done

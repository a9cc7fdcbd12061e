cd /root/repo
for V in 1 0 1 0; do echo "PIPELINE=$V"; CERES_HIP_PIPELINE=$V timeout 600 python tools/kernel_times.py venice1778 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:v for k,v in d.items() if k.endswith('_ms')}); print(d['cgnr_solve']); print(d['schur_solve'])"; done
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -8

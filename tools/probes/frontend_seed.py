import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, torch
import __graft_entry__ as e
pkg=e.load_package(); oracle=e.load_oracle(); hip=pkg.hip_solver; hip.load_library()
seed=int(sys.argv[1])
rng = np.random.default_rng(900007 * seed + 5)
nc = int(rng.choice([3, 7, 16, 40, 65, 200, 700, 2600])); npts = int(rng.choice([20, 64, 65, 300, 1500, 5000])); per = float(rng.choice([2.0, 3.0, 5.0, 9.0]))
nobs = int(min(max(2 * npts, per * npts), 0.8 * nc * npts)); skew = float(rng.choice([0.0, 0.5, 1.0]))
solver_type, pre = [(5, 2), (6, 1), (5, 1)][int(rng.integers(3))]
pn=float(rng.choice([0.1, 0.5, 2.0])); qn=float(rng.choice([0.005, 0.02]))
print(dict(nc=nc,npts=npts,nobs=nobs,skew=skew,solver=(solver_type,pre),pixel_noise=pn,param_noise=qn))
op = oracle.BalProblem.generate(nc, npts, nobs, seed=seed + 1, skew=skew, pixel_noise=pn, param_noise=qn)
bs, nelim = op.build_structure(True)
cam, pt, obs = op.indices()
o = hip.LinearSolverOptions(type=solver_type, preconditioner_type=pre, min_num_iterations=0, max_num_iterations=500)
gp = hip.BalProblem(o, op.num_cameras, op.num_points, cam, pt, obs)
x0=op.state()
Sa = op.lm_solve(solver_type=solver_type, preconditioner=pre, max_it=500, max_num_iterations=8)
x, Sb = gp.minimize(x0, max_num_iterations=8)
for i in range(max(Sa.num_iterations_logged,Sb.num_iterations_logged)):
    a=Sa.iterations[i] if i<Sa.num_iterations_logged else None; b=Sb.iterations[i] if i<Sb.num_iterations_logged else None
    fa=lambda a,rad: (a.cost, a.step_is_successful, a.linear_solver_iterations, rad, getattr(a,'step_norm',None), getattr(a,'relative_decrease',None)) if a else None
    print(i, 'oracle', fa(a, a.radius if a else None), '| hip', fa(b, b.trust_region_radius if b else None))
# condition: the first linear system
c0,r0,v0=op.evaluate(x0)
m=oracle.Matrix(bs,0)
d=m.squared_column_norm(v0); print('column norm range', d.min(), d.max())

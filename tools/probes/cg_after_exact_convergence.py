import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
import __graft_entry__ as e
pkg=e.load_package(); oracle=e.load_oracle(); hip=pkg.hip_solver; hip.load_library()
from test_gpu_operators import make_solver
p=pkg.problems.random_schur_problem(num_e_blocks=1,num_f_blocks=40,max_rows_per_e=1,num_no_e_rows=0,static_sizes=(1,1,1),seed=264)
m=oracle.Matrix(p.bs,p.num_eliminate_blocks)
for kk in (1,2,3,4):
    s=make_solver(hip,p,hip.ITERATIVE_SCHUR,hip.SCHUR_JACOBI,min_it=kk,max_it=kk)
    x,summ=s.solve(p.values,p.b,hip.PerSolveOptions(D=p.D,q_tolerance=-1.0,r_tolerance=-1.0)); s.close()
    xo,so=m.iterative_schur_solve(p.values,p.b,p.D,preconditioner=2,min_it=kk,max_it=kk,q_tol=-1.0,r_tol=-1.0)
    print(kk, summ, '|', so, 'nan hip', np.isnan(x).sum(), 'nan oracle', np.isnan(xo).sum(), np.linalg.norm(x-xo))

import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, torch
import __graft_entry__ as e
pkg=e.load_package(); oracle=e.load_oracle(); hip=pkg.hip_solver; hip.load_library()
import fuzz_cases
seed=int(sys.argv[1]); pre=int(sys.argv[2]); reset=int(sys.argv[3]); schur=int(sys.argv[4])
case,k,_=fuzz_cases.draw_case(seed); p=fuzz_cases.build(pkg.problems,case,k)
print(case)
q=type(p)(p.bs,p.values,p.b,p.D,0)
pp=p if schur else q
m=oracle.Matrix(p.bs,p.num_eliminate_blocks); m0=oracle.Matrix(p.bs,0)
fn=m.iterative_schur_solve if schur else m0.cgnr_solve
for kk in [int(a) for a in sys.argv[5:]]:
    o=hip.LinearSolverOptions(type=hip.ITERATIVE_SCHUR if schur else hip.CGNR, preconditioner_type=pre, min_num_iterations=kk, max_num_iterations=kk, residual_reset_period=reset, elimination_groups=[pp.num_eliminate_blocks])
    s=hip.HipLinearSolver(o); s.set_structure(pp.bs); path=s.info().kernel_path
    x,summ=s.solve(p.values,p.b,hip.PerSolveOptions(D=p.D,q_tolerance=-1.0,r_tolerance=-1.0)); s.close()
    xo,so=fn(p.values,p.b,p.D,preconditioner=pre,min_it=kk,max_it=kk,reset_period=reset,q_tol=-1.0,r_tol=-1.0)
    print(kk,'path',path,'rel',np.linalg.norm(x-xo)/np.linalg.norm(xo),'|x|',np.linalg.norm(xo))
for qt in (0.01,):
    o=hip.LinearSolverOptions(type=hip.ITERATIVE_SCHUR if schur else hip.CGNR, preconditioner_type=pre, min_num_iterations=0, max_num_iterations=500, residual_reset_period=reset, elimination_groups=[pp.num_eliminate_blocks])
    s=hip.HipLinearSolver(o); s.set_structure(pp.bs)
    x,summ=s.solve(p.values,p.b,hip.PerSolveOptions(D=p.D,q_tolerance=qt,r_tolerance=-1.0)); s.close()
    xo,so=fn(p.values,p.b,p.D,preconditioner=pre,min_it=0,max_it=500,reset_period=reset,q_tol=qt,r_tol=-1.0)
    print(summ.message,'|',so.message)

import sys, time, os, numpy as np
os.environ['EHM_KEEP_GOING']='1'
sys.path.insert(0,'.')
from explicit_hybrid_mpc_amd import engine, examples, tools
mpc=examples.linear_mpc(0); can=mpc.compile()
gp=engine.GpuProblem(can,1.,1.)
V=examples.box_vertices(examples.theta_box(mpc))
roots,_=tools.delaunay_roots(V)
abs_frac,eps_r=0.03,0.01
J,_,_=gp.solve_pt(abs_frac*V); eps_a=float(J.max()); gp.set_eps(eps_a,eps_r)
flat=gp.partition(roots,max_nodes=1<<22)
bad=np.nonzero(flat.flags & 24)[0]
print('bad nodes',bad, flat.flags[bad])
np.savez('gpurun_out/bad_nodes.npz', R=flat.vertices[bad], V=flat.vertex_costs[bad], flags=flat.flags[bad], eps_a=eps_a, eps_r=eps_r, tstar=flat.tstar[bad])

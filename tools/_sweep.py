import sys, time, os, numpy as np
os.environ['EHM_KEEP_GOING']='1'
sys.path.insert(0,'.')
from explicit_hybrid_mpc_amd import engine, examples, tools
mpc=examples.linear_mpc(0); can=mpc.compile()
gp=engine.GpuProblem(can,1.,1.)
V=examples.box_vertices(examples.theta_box(mpc))
roots,_=tools.delaunay_roots(V)
for abs_frac,eps_r in [(0.1,0.01),(0.05,0.01),(0.03,0.01),(0.02,0.01)]:
    J,_,_=gp.solve_pt(abs_frac*V); eps_a=float(J.max()); gp.set_eps(eps_a,eps_r)
    s0=gp.stats()
    t=time.perf_counter()
    try:
        info=gp.partition(roots,export=False,with_volume=False,max_nodes=1<<23)
    except Exception as e:
        print(abs_frac,eps_r,'ERR',e); continue
    dt=time.perf_counter()-t
    s1=gp.stats()
    print(abs_frac,eps_r,'eps_a %.4g'%eps_a,'nodes',info['n_nodes'],'closed',info['n_closed'],'solves',info['lp_solves'],'it/solve %.2f'%(info['ipm_iters']/info['lp_solves']),'wall %.3f'%dt,'dev %.3f'%info['device_seconds'],'decide %.3f expand %.3f'%(info['decide_seconds'],info['expand_seconds']),'solves/s %.3g'%(info['lp_solves']/dt),'depth',info['max_depth'],'margin %.2e'%info['min_margin'],'stalled',s1['stalled']-s0['stalled'])
    sys.stdout.flush()

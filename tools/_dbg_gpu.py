import numpy as np, sys
sys.path.insert(0,'.')
from explicit_hybrid_mpc_amd import examples, engine
from oracle.oracle_cpu import OracleCPU
from oracle.partition_cpu import PartitionCPU
from tests import helpers
mpc=helpers.make_instance('lin',0); eps_r=1.0
eps_a=helpers.eps_a_rule(mpc,0.5)
roots,locs=helpers.roots_of(mpc)
orc=OracleCPU(mpc,eps_a,eps_r); cpu=PartitionCPU(orc); cpu.run(roots,locs,'ecc')
gp=engine.GpuProblem(mpc.compile(),eps_a,eps_r)
flat=gp.partition(np.array(roots),action='ecc')
loc=flat.locations(locs)
gl=set(loc); cl=set(cpu.nodes)
print('only cpu',sorted(cl-gl),'only gpu',sorted(gl-cl))
idx={n:k for k,n in enumerate(loc)}
for name in sorted((cl-gl)|(gl-cl)):
    par=name[:-1]
    k=idx[par]; ref=cpu.nodes[par]
    print(par,'gpu t',flat.tstar[k],'leaf',flat.is_leaf(k),'flags',flat.flags[k])
    print('  verts equal',np.array_equal(flat.vertices[k],ref['vertices']))
    print('  vcost gpu',flat.vertex_costs[k],'cpu',ref['vertex_costs'])
    t_ref,_=orc.slack(ref['vertices'],ref['vertex_costs'],0)
    t_g,_,st=gp.slack(ref['vertices'][None],ref['vertex_costs'][None])
    t_g2,_,st2=gp.slack(flat.vertices[k][None],flat.vertex_costs[k][None])
    print('  cpu slack',t_ref,'gpu slack on cpu data',t_g,st,'gpu slack on gpu data',t_g2,st2)

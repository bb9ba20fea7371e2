import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, blu_amd
from blu_amd import keys as K
from oracle import orc
for spec in [(300,5,4,0.5,1,0.3),(2000,8,8,0.5,1,0.3),(1500,8,16,0.2,3,1.0),(10000,8,8,0.5,1,0.3)]:
    cp,ri,v=orc.gen_lp_basis(*spec); m=spec[0]
    g=blu_amd.BLU(m,len(ri)); o=orc.OracleBLU(m,32*len(ri)); o.set_fix_d3(True)
    g.factorize(cp[:-1],cp[1:],ri,v); o.factorize(cp[:-1],cp[1:],ri,v)
    out=[]
    for c in ("CONDEST_L","CONDEST_U","NORM_L","NORM_U","NORMEST_L_INV","NORMEST_U_INV","ONENORM","INFNORM","RESIDUAL_TEST"):
        a,b=g.stat(getattr(K,"STAT_"+c)),o.stat(getattr(K,"STAT_"+c))
        out.append("%s %s"%(c, "==" if a==b else "%.1e"%(abs(a-b)/abs(b))))
    print(m, " | ".join(out))

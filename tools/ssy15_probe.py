import sys; sys.path.insert(0,'/root/repo')
import sdfs_via_autodiff_amd as S, numpy as np, time
m=S.SSY(); shp=(15,)*4
T=S.ssy_operator(shp,m.params,S.discretize_ssy(m,shp))
print(T.describe_plan())
for ce in (16,32,64,128,256):
    T.solve(np.full(shp,800.0),"successive_approx",tol=1e-8,max_iter=64,check_every=ce)
    t=time.perf_counter(); x,n,info=T.solve(np.full(shp,800.0),"successive_approx",tol=1e-8,check_every=ce); dt=time.perf_counter()-t
    print("check_every",ce,"iters",n,"s",round(dt,4),"it/s",round(n/dt))
T.set_profiling(True); T.reset_counters()
x,n,info=T.solve(np.full(shp,800.0),"successive_approx",tol=1e-8,max_iter=2000,use_graph=0)
for c in T.counters(): print(c["name"], c["launches"], round(c["total_ms"]/c["launches"]*1e3,2),"us")

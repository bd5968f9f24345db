import sys; sys.path.insert(0,'/root/repo')
import sdfs_via_autodiff_amd as S
m=S.GCY(); shp=(20,)*6
T=S.gcy_operator(shp,m.params,S.discretize_gcy(m,shp))
print(T.describe_plan())

import sys; sys.path.insert(0,'.')
import numpy as np, time
from oracle import oracle
from visual_odometry_amd import synth, ops
K=synth.KITTI_K
rng=np.random.default_rng(3)
n=3000
X=np.stack([rng.uniform(-20,20,n), rng.uniform(-5,5,n), rng.uniform(8,60,n)],1)
R=synth.rot_y(2.0); C=np.array([0.8,0.05,0.3])
p0,_=synth.project(K,np.eye(3),np.zeros(3),X); p1,_=synth.project(K,R,C,X)
p0+=rng.normal(0,0.05,p0.shape); p1+=rng.normal(0,0.05,p1.shape)
out=rng.random(n)<0.2; p1[out]+=rng.uniform(-30,30,(out.sum(),2))
o=oracle.essential_ransac(p0,p1,K,num_hyp=2000,threads=8,want_all=True)
t=time.time(); g=ops.essential_ransac(p0,p1,K,num_hyp=2000,want_all=True); print('gpu ms',1e3*(time.time()-t))
t=time.time(); g=ops.essential_ransac(p0,p1,K,num_hyp=2000,want_all=True); print('gpu ms',1e3*(time.time()-t))
print('status',o['status'],g['status'],'best',o['best_hyp'],g['best_hyp'],'trials',o['trials'],g['trials'],'inl',o['inliers'].sum(),g['num_inliers'])
print('E hyp max diff', np.abs(o['Es']-g['Es']).max(), 'bit-equal frac', (o['Es']==g['Es']).all(axis=(1,2)).mean())
print('counts equal', np.array_equal(o['counts'],g['counts']), 'costs equal', np.array_equal(o['costs'],g['costs']), 'inliers equal', np.array_equal(o['inliers'],g['inliers']))
g2=ops.essential_ransac(p0,p1,K,num_hyp=2000,hyp_E=o['Es'],want_all=True)
print('identical hypotheses: counts', np.array_equal(o['counts'],g2['counts']), 'costs', np.array_equal(o['costs'],g2['costs']), 'inl', np.array_equal(o['inliers'],g2['inliers']), 'E', np.array_equal(o['E'],g2['E']))

import cProfile, pstats, numpy as np, torch, time
from voicemap_amd.engine import HipEncoderEngine
F,E=128,64
blocks=[(32,F,4),(3,2*F,2),(3,3*F,2),(3,4*F,2)]
eng=HipEncoderEngine(blocks,E,dropout=0.0,head="uniform_euclidean",dtype="f16",seed=1)
pairs=8
rng=np.random.default_rng(0)
x=torch.from_numpy(rng.normal(0,0.05,(2*pairs,48000)).astype(np.float32)).cuda()
y=torch.cat([torch.zeros(pairs//2),torch.ones(pairs-pairs//2)]).cuda()
pl=eng.plan(2*pairs,12000,True)
def step():
    eng.preprocess(pl,x,4,True,pairs); eng.forward(pl,pairs,None); eng.siamese_head(pl,y,"contrastive"); eng.backward(pl); eng.optimizer_step()
for _ in range(20): step()
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(200): step()
t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
print("host enqueue %.3f ms/step, wall %.3f ms/step" % ((t1-t0)/200*1e3,(t2-t0)/200*1e3))
pr=cProfile.Profile(); pr.enable()
for _ in range(200): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)

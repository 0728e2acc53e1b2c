import sys,time
sys.path.insert(0,'/root/repo')
from webdgs_amd import ops
dev=ops.HipDevice(0)
def rnd(label):
    t0=time.perf_counter(); bs=[dev.createBuffer(1<<20) for _ in range(50)]; dev.synchronize(); c=(time.perf_counter()-t0)/50*1e3
    t0=time.perf_counter()
    for b in bs: b.destroy()
    d=(time.perf_counter()-t0)/50*1e3
    print(label,'create ms',round(c,4),'destroy ms',round(d,4))
rnd('python fresh'); rnd('python again')

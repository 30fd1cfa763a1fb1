import os, time, torch
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads())
try:
    print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e: print("no cgroup", e)
a=torch.randn(2048,1536); w=torch.randn(8192,1536)
for nt in (8,16,32,64,128,256):
    torch.set_num_threads(nt)
    (a@w.t()); t=time.time()
    for _ in range(3): (a@w.t())
    dt=(time.time()-t)/3
    print(nt, "threads:", round(2*2048*1536*8192/dt/1e9,1), "GFLOP/s")

import torch, time
n = 1<<30
p = torch.empty(n, dtype=torch.uint8).pin_memory()
q = torch.empty(n, dtype=torch.uint8)
q.fill_(1)
d = torch.empty(n, dtype=torch.uint8, device='cuda')
for name, src in (("pinned", p), ("pageable", q)):
    for _ in range(2):
        torch.cuda.synchronize(); t=time.perf_counter(); d.copy_(src, non_blocking=True); torch.cuda.synchronize(); dt=time.perf_counter()-t
    print(name, "%.1f GB/s" % (n/dt/1e9))
import numpy as np, threading
a = np.frombuffer(q.numpy(), dtype=np.uint8); b = p.numpy()
def cp(lo,hi): b[lo:hi] = a[lo:hi]
for T in (1,2,4,8):
    t=time.perf_counter(); th=[threading.Thread(target=cp,args=(i*n//T,(i+1)*n//T)) for i in range(T)]; [x.start() for x in th]; [x.join() for x in th]; dt=time.perf_counter()-t
    print("memcpy threads", T, "%.1f GB/s" % (n/dt/1e9))

"""Development aid: ways of getting one tick's packets (128 x 40 x 64 float64 = 2.6 MB) from host to device.  On the MI355X box: pageable
copy_ 62 us, pinned async copy 55 us, numpy copy into a pinned buffer + async copy 111 us -- the plain copy the pipelines use is at the DMA floor."""
import time, numpy as np, torch
S,P,C=128,40,64
dev=torch.empty((S,P,C),dtype=torch.float64,device='cuda')
pin=torch.empty((S,P,C),dtype=torch.float64).pin_memory()
pin_np=pin.numpy()
rng=np.random.default_rng(0)
def t(fn,n=200):
    xs=[]
    for _ in range(n):
        pk=rng.standard_normal((S,P,C))
        torch.cuda.synchronize(); t0=time.perf_counter(); fn(pk); torch.cuda.synchronize(); xs.append((time.perf_counter()-t0)*1e6)
    return np.percentile(xs[20:],50)
print('pageable copy_            p50 us', t(lambda pk: dev.copy_(torch.from_numpy(pk))))
def viapin(pk):
    np.copyto(pin_np, pk); dev.copy_(pin, non_blocking=True)
print('np.copyto pinned + async  p50 us', t(viapin))
def pinned_only(pk):
    dev.copy_(pin, non_blocking=True)
print('pinned async only         p50 us', t(pinned_only))
def host_copy(pk):
    np.copyto(pin_np, pk)
print('np.copyto into pinned     p50 us', t(host_copy))
tmp=np.empty((S,P,C))
print('np.copyto into pageable   p50 us', t(lambda pk: np.copyto(tmp,pk)))

import time, numpy as np, torch
from concurrent.futures import ThreadPoolExecutor
dev = torch.device('cuda', 0)
x = np.random.default_rng(0).standard_normal((512, 6, 48, 48, 14)).astype(np.float32)
pin = [torch.empty((32, 6, 48, 48, 14), dtype=torch.float32).pin_memory() for _ in range(4)]
pv = [p.numpy() for p in pin]
sel = np.random.default_rng(1).permutation(512)[:32]
def t(f, n=20):
    f(); t0 = time.time()
    for _ in range(n): f()
    return (time.time() - t0) / n * 1e3
mb = pv[0].nbytes / 1e6
print('batch array %.1f MB' % mb)
print('copyto slice -> pinned, 1 thread: %.2f ms' % t(lambda: np.copyto(pv[0], x[32:64])))
print('take -> pinned, 1 thread: %.2f ms' % t(lambda: np.take(x, sel, axis=0, out=pv[0])))
print('fancy a[sel] (pageable): %.2f ms' % t(lambda: x[sel]))
pool = ThreadPoolExecutor(4)
def par_copy():
    list(pool.map(lambda i: np.copyto(pv[0][8*i:8*i+8], x[32+8*i:40+8*i]), range(4)))
def par_take():
    list(pool.map(lambda i: np.take(x, sel[8*i:8*i+8], axis=0, out=pv[0][8*i:8*i+8]), range(4)))
print('copyto slice -> pinned, 4 threads: %.2f ms' % t(par_copy))
print('take -> pinned, 4 threads: %.2f ms' % t(par_take))
pool8 = ThreadPoolExecutor(8)
def par_take8():
    list(pool8.map(lambda i: np.take(x, sel[4*i:4*i+4], axis=0, out=pv[0][4*i:4*i+4]), range(8)))
print('take -> pinned, 8 threads: %.2f ms' % t(par_take8))
d = torch.empty((32, 6, 48, 48, 14), dtype=torch.float32, device=dev)
def h2d_pin():
    d.copy_(pin[0], non_blocking=True); torch.cuda.synchronize()
print('H2D from pinned: %.2f ms (%.1f GB/s)' % (t(h2d_pin), mb / t(h2d_pin)))
pg = torch.from_numpy(np.ascontiguousarray(x[:32]))
def h2d_page():
    d.copy_(pg); torch.cuda.synchronize()
print('H2D from pageable: %.2f ms' % t(h2d_page))
def to_bf():
    d.to(torch.bfloat16); torch.cuda.synchronize()
print('device fp32 -> bf16: %.3f ms' % t(to_bf))

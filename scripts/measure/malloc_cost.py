"""hipMalloc's cost by size on the GPU box (through torch's allocator, cache emptied between): the first genome of a fresh context at 5 kb took
1.9 s in hpk_submit_batch - workspaces of 15-19 GB each."""
import time, torch
torch.cuda.init(); torch.zeros(1, device='cuda'); torch.cuda.synchronize()
def one(gb):
    t0 = time.perf_counter(); x = torch.empty(int(gb * (1 << 30)), dtype=torch.uint8, device='cuda'); torch.cuda.synchronize(); t1 = time.perf_counter()
    return x, (t1 - t0) * 1e3
for gb in (1, 4, 8, 12, 16, 17, 20, 24, 32, 16, 32):
    x, ms = one(gb)
    print('%5.1f GiB: hipMalloc %.1f ms' % (gb, ms))
    del x; torch.cuda.empty_cache()
keep = []
t0 = time.perf_counter()
for i in range(6):
    x, ms = one(12)
    keep.append(x)
    print('  12 GiB #%d (kept): %.1f ms' % (i, ms))
print('six kept allocations of 12 GiB: %.1f ms' % ((time.perf_counter() - t0) * 1e3))

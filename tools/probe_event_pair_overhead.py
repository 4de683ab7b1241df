import torch
dev=torch.device("cuda:0")
big = torch.empty(1 << 29, dtype=torch.uint8, device=dev)
def head():
    for _ in range(80): big.fill_(0)
def run(nbytes, reps=256):
    buf = torch.empty(max(nbytes,64), dtype=torch.uint8, device=dev)
    for _ in range(20): buf.fill_(1)
    torch.cuda.synchronize()
    pairs=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    head()
    for a,b in pairs:
        a.record(); buf.fill_(1); b.record()
    torch.cuda.synchronize()
    wp=sum(a.elapsed_time(b) for a,b in pairs)
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    head(); s.record()
    for _ in range(reps): buf.fill_(1)
    e.record(); torch.cuda.synchronize()
    pl=s.elapsed_time(e)
    # third arm: events between kernels but measured by the outer pair (total timeline cost of having events)
    s2,e2=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    head(); s2.record()
    for a,b in pairs:
        a.record(); buf.fill_(1); b.record()
    e2.record(); torch.cuda.synchronize()
    tl=s2.elapsed_time(e2)
    print("%10d B: per launch: pairs' sum %.2f us | back-to-back %.2f us | timeline with events %.2f us | pair - b2b = %.2f us" % (nbytes, wp/reps*1e3, pl/reps*1e3, tl/reps*1e3, (wp-pl)/reps*1e3))
for n in (64, 1<<20, 8<<20, 32<<20, 128<<20):
    run(n)

import torch, json
dev=torch.device('cuda',0)
for mib in (16, 64, 128, 192, 512, 4096):
    x=torch.zeros(mib*1024*1024//4, device=dev)
    for _ in range(5): x.add_(1.0)
    torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    n=50
    a.record()
    for _ in range(n): x.add_(1.0)
    b.record(); torch.cuda.synchronize()
    us=a.elapsed_time(b)/n*1e3
    print(json.dumps({"buffer_MiB":mib,"us_per_pass":us,"rmw_GBps":2*mib*1.048576e6/us/1e3}))

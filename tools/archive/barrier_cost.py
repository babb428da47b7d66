import os, time, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29519", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
dist.barrier(); torch.cuda.synchronize()
for name, fn in (("dist.barrier", lambda: dist.barrier()),
                 ("all_reduce+sync", lambda: (dist.all_reduce(t), torch.cuda.synchronize())),
                 ("all_reduce async + item", lambda: (dist.all_reduce(t), t.item()))):
    t = torch.zeros(1, device="cuda:0")
    fn()
    ts = []
    for _ in range(20):
        a = time.perf_counter(); fn(); ts.append(time.perf_counter() - a)
    ts.sort(); print(f"{name:28s} median {ts[10]*1e6:8.1f} us  min {ts[0]*1e6:8.1f}  max {ts[-1]*1e6:8.1f}")
dist.destroy_process_group()

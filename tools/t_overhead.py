"""GPU box: host-side cost of the torch.distributed calls bench.py's exchange makes per step (one-rank RCCL group):
counts all-gather + download, building 14 P2POps, one batch_isend_irecv.  Round 1: 26 us, 28 us, 41 us."""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29544")
dev=torch.device("cuda",0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
mine=torch.tensor([123],dtype=torch.int64,device=dev); counts=torch.empty(1,dtype=torch.int64,device=dev)
buf=torch.zeros(1<<20,dtype=torch.uint8,device=dev)
for _ in range(5):
    dist.all_gather_into_tensor(counts,mine); counts.cpu()
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(200):
    dist.all_gather_into_tensor(counts,mine); c=counts.cpu().tolist()
torch.cuda.synchronize()
print("all_gather_into_tensor + .cpu(): %.1f us per call" % ((time.perf_counter()-t0)/200*1e6))
# P2POp construction cost (no peers: objects only)
t0=time.perf_counter()
for _ in range(200):
    ops=[]
    for peer in range(7):
        ops.append(dist.P2POp(dist.isend, buf[:1000], 0)); ops.append(dist.P2POp(dist.irecv, buf[2000:3000], 0))
print("14 P2POp objects + slices: %.1f us" % ((time.perf_counter()-t0)/200*1e6))
# self send/recv batch (world 1: rank 0 to itself) to price batch_isend_irecv's fixed cost
try:
    a=buf[:4096]; b=buf[8192:8192+4096]
    for _ in range(3):
        reqs=dist.batch_isend_irecv([dist.P2POp(dist.isend,a,0),dist.P2POp(dist.irecv,b,0)])
        for r in reqs: r.wait()
    torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(100):
        reqs=dist.batch_isend_irecv([dist.P2POp(dist.isend,a,0),dist.P2POp(dist.irecv,b,0)])
        for r in reqs: r.wait()
    torch.cuda.synchronize()
    print("batch_isend_irecv (1 send + 1 recv to self): %.1f us per call" % ((time.perf_counter()-t0)/100*1e6))
except Exception as e:
    print("self p2p failed:", str(e)[:200])
dist.destroy_process_group()

import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
os.environ.update(TRICOLO_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", RANK="0", WORLD_SIZE="1")
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from tricolo_amd import config as tcfg, ops, parallel
from tricolo_amd.model.tricolo_net import TriCoLoNet
from tricolo_amd.data import synthetic as syn
from oracle.recipe import fill_module
ops.set_default_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16x3")
S, nv, B = int(os.environ.get("S", 64)), int(os.environ.get("NV", 2)), int(os.environ.get("B", 8))
def build():
    cfg = tcfg.compose(overrides=["model.text_encoder=BiGRUEncoder", "model.image_encoder=MVCNNEncoder", "model.voxel_encoder=SparseCNNEncoder",
                                  "data=synthetic", "data.voxel_size=32", f"data.num_views={nv}", f"data.image_size={S}", "experiment_name=t"])
    net = TriCoLoNet(cfg); fill_module(net); return net.to("cuda")
batch = syn.batch_to_device(syn.make_batch(B, voxel_size=32, num_views=nv, image_size=S, seed=syn.BASE_SEED + 31), "cuda")
if os.environ.get("FIRST", "0") == "1":
    n0 = build(); o0 = n0.configure_optimizers(); o0.prepare()
    for _ in range(2): parallel.dp_training_step(n0, batch, o0)
    torch.cuda.synchronize(); g0 = parallel.GraphedDPStep(n0, o0, batch); print("non-split graph ok", g0.replay().item(), flush=True)
net = build(); opt = net.configure_optimizers(); opt.prepare()
split = parallel.BackwardSplit.for_net(net)
for i in range(2):
    print("eager split step", parallel.dp_training_step(net, batch, opt, split=split)["train_loss/total_loss"].item(), flush=True)
torch.cuda.synchronize()
print("capturing", flush=True)
g = parallel.GraphedDPStep(net, opt, batch, split=split)
print("captured", flush=True)
for i in range(3):
    print("replay", g.replay().item(), flush=True)
import time
for name, gg in (("split", g),) + ((("single", g0),) if os.environ.get("FIRST", "0") == "1" else ()):
    for _ in range(20): gg.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): gg.replay()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name}: host enqueue {(t1 - t0) / 50 * 1e3:.3f} ms/step, total {(t2 - t0) / 50 * 1e3:.3f} ms/step", flush=True)
    # pieces of the split replay on the host
    if name == "split":
        import torch.distributed as dist
        def tm(fn, n=50):
            torch.cuda.synchronize(); a = time.perf_counter()
            for _ in range(n): fn()
            b = time.perf_counter(); torch.cuda.synchronize(); return (b - a) / n * 1e3, (time.perf_counter() - a) / n * 1e3
        print("gA", tm(gg.gA.replay), "gB", tm(gg.gB.replay), "gB2", tm(gg.gB2.replay), "gC", tm(gg.gC.replay))
        print("allgather", tm(lambda: dist.all_gather_into_tensor(gg.full, gg.packed)))
        s_, e_, _ = gg.early_runs[-1]
        print("allreduce sync", tm(lambda: dist.all_reduce(gg.flat[s_:e_])), "async+wait", tm(lambda: dist.all_reduce(gg.flat[s_:e_], async_op=True).wait()))
dist.destroy_process_group()

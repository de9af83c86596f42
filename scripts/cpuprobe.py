import os, time, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import torch, iq_oracle as O
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads())
kw = dict(in_channels=1, img_size_h=224, img_size_w=224, patch_size=16, num_classes=19, d_model=192, n_head=3, n_layers=12, ffn_hidden=768)
cfg = O.OracleConfig(kind="vit", drop_prob=0.1, **kw); sd = O.init_state(cfg, 0); st = O.adamw_init(sd)
x = torch.randn(8, 1, 224, 224); y = torch.randint(0, 19, (8,))
for nt in (8, 16, 32):
    torch.set_num_threads(nt)
    O.train_step(cfg, sd, st, x, y)
    t0 = time.perf_counter(); O.train_step(cfg, sd, st, x, y); O.train_step(cfg, sd, st, x, y); el = time.perf_counter() - t0
    print(f"threads {nt}: {16/el:.1f} frames/s", flush=True)

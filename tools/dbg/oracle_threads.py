import os, time, torch
import sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from oracle import lr2ppo_oracle as O
print("default threads", torch.get_num_threads(), "affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
try:
    print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("no cpu.max", e)
P = O.seeded_params(O.head_param_spec("actor"), seed=7)
text = torch.randn(8, 20, 196, 768); img = torch.randn(8, 16, 768).unsqueeze(1).repeat(1, 20, 1, 1)
for nt in (torch.get_num_threads(), 64, 32, 16, 8):
    torch.set_num_threads(nt)
    with torch.no_grad():
        O.actor_forward(P, text[:2], img[:2], None)
        t0 = time.time(); O.actor_forward(P, text, img, None); dt = time.time() - t0
    print(nt, "threads:", round(dt, 2), "s for 8 items", flush=True)

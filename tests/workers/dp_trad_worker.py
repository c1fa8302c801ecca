"""2-rank rehearsal (gloo; both ranks on cuda:0) of the `_trad` pointwise twins under data parallelism -- the one script the reference
wraps in DistributedDataParallel (finetune/pointwise_trad.py:446-448).  Each rank steps pointwise_trad / pointwise_2data_trad on ITS half
of a batch (gradients averaged over the ranks by pointwise_trad.average_grads); a second copy of the model steps on the WHOLE batch
(every rank computes the same gradient there, so the average changes nothing).  Afterwards the replicas must hold the same bits and the
half-batch model must equal the whole-batch one up to summation order.  Launched by tests/test_dp_gpu.py through torch.distributed.run."""
import argparse
import copy
import os
import sys
import warnings

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def flat(model):
    return torch.cat([q.detach().double().flatten() for _, q in sorted(model.named_parameters(), key=lambda kv: kv[0])]).cpu()


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from lr2ppo_amd.finetune import pointwise_2data_trad as p2, pointwise_trad as pt, ppo
    args = argparse.Namespace(mode="reg", labels_num=3, optimizer="adamw", scheduler="linear", learning_rate=1e-3, train_steps=21,
                              warmup=0.1, device=dev, batch_size=2)
    for mod, widths in ((pt, (768, 768)), (p2, (46, 136))):
        torch.manual_seed(9)                                        # identical replicas
        half = mod.Classifier(args, None)
        ppo._init_normal(half)
        whole = copy.deepcopy(half)
        half, whole = half.to(dev).eval(), whole.to(dev).eval()     # dropout off: the two models must see the same function
        start = flat(half)
        opts = [mod.build_optimizer(args, m) for m in (half, whole)]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for _, sch in opts:
                sch.step()                                          # past lambda(0) = 0
        g = torch.Generator().manual_seed(31)                       # the same 2 * world queries on every rank
        for step, width in enumerate(widths):
            feats = torch.randn(2 * world, 20, width, generator=g).to(dev)
            tgts = torch.randint(0, 3, (2 * world, 20), generator=g).float().to(dev)
            mine = slice(2 * rank, 2 * rank + 2)
            l_half = mod.train_model(args, half, *opts[0], feats[mine].contiguous(), None, tgts[mine].contiguous())
            l_whole = mod.train_model(args, whole, *opts[1], feats, None, tgts)
            dist.all_reduce(l_half.div_(world))                      # the launcher's logging all-reduce: mean of the rank means
            assert abs(l_half.item() - l_whole.item()) < 1e-5 * max(1.0, abs(l_whole.item())), (step, l_half.item(), l_whole.item())
        a, b = flat(half), flat(whole)
        every = [torch.zeros_like(a) for _ in range(world)]
        dist.all_gather(every, a)
        assert all(torch.equal(every[0], e) for e in every), f"{mod.__name__}: replicas differ after the steps"
        moved = (a - start).norm().item()
        rel = ((a - b).norm() / (b - start).norm()).item()
        assert moved > 0 and rel < 1e-3, (mod.__name__, moved, rel)   # AdamW's first steps are sign-like: a few near-zero gradients may flip
        if rank == 0:
            print(f"{mod.__name__}: update norm {moved:.3e}, half-batch vs whole-batch relative difference of the update {rel:.2e}", flush=True)
    dist.barrier()
    if rank == 0:
        print("DP_TRAD_REPLICAS_IDENTICAL_OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

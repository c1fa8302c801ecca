#!/usr/bin/env python3
"""PPO steps/s of the LR2PPO stage-3 hot path on MI355X (BASELINE.json metric).

One "PPO step"  = one rollout timestep (actor + critic + reward no-grad forwards on a batch,
finetune/ppo.py:844-883) + one update minibatch (actor fwd/bwd/AdamW + critic fwd/bwd/AdamW with the fused PPO
loss, finetune/ppo.py:518-587, dropout active as under model.train()) on a batch of the same shape.
Synthetic LRMovieNet-shaped inputs resident in HBM, random N(0, 0.02) weights of the reference architecture
(519 M-parameter actor, 526 M critic and reward), batch 32 items x 2 tags per GPU.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line on rank 0.  `roofline` describes the kernel signature that took the most device time in the
timed region (HIP events on the launch stream); `cpu_baseline` is the CPU oracle ("port") timed on the host cores
of this box on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.3 TB/s achievable)
MFMA_BF16_PEAK_TF = 2500.0   # dense bf16 MFMA peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="items per GPU per step (BASELINE: 32)")
    ap.add_argument("--tags", type=int, default=2)
    ap.add_argument("--passes", type=int, default=3, choices=[1, 3], help="GEMM precision: 3 = split-bf16 (parity mode)")
    ap.add_argument("--fuse-fc1", type=int, default=1, choices=[0, 1],
                    help="1: AdamW step of out_layer.fc1.weight inside its weight-gradient GEMM (default); 0: separate passes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=1, help="measured CPU-baseline steps at --batch (after one untimed warm-up step)")
    ap.add_argument("--no-online", action="store_true",
                    help="skip the second timed loop (frames + token ids -> ViT-B/16 + RoBERTa-base -> PPO step)")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel HIP events in the timed region")
    ap.add_argument("--no-stage1", action="store_true", help="skip the secondary BASELINE configs[1] figure (stage-1 step at 32 x 20 tags)")
    ap.add_argument("--serial-streams", action="store_true",
                    help="one HIP stream for the whole PPO step (LR2_PPO_STREAMS=0): every launch runs alone, so a kernel trace "
                         "of this command shows exclusive per-kernel durations; the default schedule runs the critic beside the actor")
    return ap.parse_args()


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product path has no CPU fallback)")
    local = local % torch.cuda.device_count()        # rehearsals of N > 1 on a one-GPU box share the device
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL ("nccl") is the product path; LR2_BENCH_BACKEND=gloo exists only to rehearse the N > 1 code on one GPU
        dist.init_process_group(os.environ.get("LR2_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
    if a.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {a.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)

    if a.serial_streams:
        os.environ["LR2_PPO_STREAMS"] = "0"
    from lr2ppo_amd import _native
    if rank == 0:
        _native.build()
    if world > 1:
        dist.barrier()
    from lr2ppo_amd import ops, runtime
    from lr2ppo_amd.finetune import ppo

    ops.set_gemm_passes(a.passes)
    margs = argparse.Namespace(mode="reg", labels_num=3, seq_length=196, max_imgs=16, visual_feat_dim=768, is_master=rank == 0,
                               kl_div_loss_weight=0.001, entropy_weight=0.001, value_clip=0.5, optimizer="adamw",
                               scheduler="linear", learning_rate=1e-3, critic_learning_rate=1e-3, train_steps=1000, warmup=0.1,
                               device=dev, fuse_fc1_update=bool(a.fuse_fc1))
    # identical replicas on every rank: same seed for the weights, rank-specific seed for the data
    torch.manual_seed(7)
    torch.cuda.manual_seed(7)
    model = ppo.ActorCritic(margs, None).to(dev)
    reward = ppo.Reward(margs, None).to(dev).eval()
    with torch.no_grad():
        for p in list(model.parameters()) + list(reward.parameters()):
            p.normal_(0, 0.02)                     # the reference's initialiser (finetune/ppo.py:362-365), on device
    opt, copt, sch, csch = ppo.build_optimizer(margs, model)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(20):                        # leave the lr-0 first cycle (quirk 15): lr = 20/100 of 1e-3
            sch.step(), csch.step()
    model.actor.bind_grads(), model.critic.bind_grads()
    runtime.set_dropout_seed(1234 + rank)
    dp = ppo._DataParallel()

    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    n_batches = 4
    data = [(torch.randn(a.batch, a.tags, 196, 768, device=dev, generator=g),
             torch.randn(a.batch, 16, 768, device=dev, generator=g),          # shared by the tags of an item
             torch.randint(0, 3, (a.batch, a.tags), device=dev, generator=g)) for _ in range(n_batches)]

    def step(i):
        text, img, tgts = data[i % n_batches]
        model.eval()
        rec = ppo.rollout_step(model, reward, text, img, tgts)
        model.train()
        return ppo.update_minibatch(margs, model, opt, copt, rec, dp)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    class exclusive_launches:
        """Inside: the PPO step on ONE stream (what LR2_PPO_STREAMS=0 / --serial-streams selects), so that HIP events around a
        launch time that launch alone.  Under the default schedule the critic's launches share the chip with the actor's and
        a launch's wall duration is not a measurement of its bandwidth."""

        def __enter__(self):
            self.old = os.environ.get("LR2_PPO_STREAMS")
            os.environ["LR2_PPO_STREAMS"] = "0"

        def __exit__(self, *exc):
            if self.old is None:
                os.environ.pop("LR2_PPO_STREAMS", None)
            else:
                os.environ["LR2_PPO_STREAMS"] = self.old

    # Per-kernel HIP events: every GEMM / AdamW signature during the LAST warm-up step (ranking, `top_ms_per_step`);
    # inside the timed region only the dominant signature is bracketed, so that the measurement does not slow the
    # thing it measures (two event records per launch on ~120 launches made the step host-bound).
    survey = {}
    for i in range(a.warmup):
        last = (i == a.warmup - 1) and not a.no_profile
        if last:
            fence()
            ops.profile_start()
            with exclusive_launches():
                m = step(i)
            survey = ops.profile_stop()
        else:
            m = step(i)
    fence()
    # Host cost of enqueueing one step, measured on an EMPTY launch queue (3 steps after a synchronise).  Inside the timed
    # loop the host runs ahead of the GPU until the HIP launch queue is full and then spins on back-pressure, so wall time
    # around a 20-step loop says how long the GPU took, not what the host spent (measured: 3.5 ms per step on an empty
    # queue, 11.6 "ms" in a 20-step loop of the same code).
    t0 = time.perf_counter()
    for i in range(3):
        step(i)
    host_enqueue_ms = (time.perf_counter() - t0) / 3 * 1e3
    fence()
    dominant = max(survey.items(), key=lambda kv: kv[1]["ms"])[0] if survey else None
    if not a.no_profile:
        ops.profile_start(only=None if dominant is None else [dominant])
    t0 = time.perf_counter()
    for i in range(a.steps):
        m = step(a.warmup + i)
    t_host = time.perf_counter() - t0       # launches enqueued; the GPU may still be running
    fence()
    dt = time.perf_counter() - t0
    prof = ops.profile_stop() if not a.no_profile else {}
    if not torch.isfinite(m).all():
        raise SystemExit("bench: non-finite PPO metrics")
    # The dominant signature again, each launch ALONE on the chip: EXCL extra steps on one stream right after the timed
    # region (same process, same inputs, same launches).  This is the duration the roofline fraction is computed from; the
    # duration inside the timed region (two streams) is reported beside it.
    prof_excl, EXCL = {}, 4
    multi_stream = os.environ.get("LR2_PPO_STREAMS", "1") != "0"
    if prof and multi_stream:
        with exclusive_launches():
            step(a.warmup + a.steps)
            fence()
            ops.profile_start(only=[dominant])
            for i in range(EXCL):
                step(a.warmup + a.steps + 1 + i)
            fence()
            prof_excl = ops.profile_stop()
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    ms_per_step = dt / a.steps * 1e3
    value = world * a.steps / dt
    out = {
        "metric": "ppo_steps_per_sec", "value": round(value, 3), "unit": "PPO steps/s (1 step = 1 rollout batch + 1 update minibatch, 32 items x 2 tags per GPU)",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (GEMMs: split-bf16 x3 on MFMA, fp32 accumulate)" if a.passes == 3 else "bf16 inputs, fp32 accumulate (1 pass)",
        "data": "synthetic",
        "config": {"workload": "LR2PPO stage-3 head-only PPO step (actor 519M + critic 526M + reward 526M params), "
                               "LRMovieNet-shaped synthetic features: text_emb [32,2,196,768], img_emb [32,16,768]",
                   "batch_per_gpu": a.batch, "tags": a.tags, "global_batch": a.batch * world, "parallelism": f"dp{world}",
                   "schedule": "one HIP stream" if os.environ.get("LR2_PPO_STREAMS", "1") == "0" else
                               "critic forward / backward / optimizer step on a second HIP stream beside the actor's",
                   "items_per_sec": round(value * a.batch, 1),
                   "algorithmic_tflop_per_step": 3.44 if (a.batch, a.tags) == (32, 2) else None},
    }
    # ---- roofline of the dominant kernel signature in the timed region ----
    if prof:
        key, rec_timed = max(prof.items(), key=lambda kv: kv[1]["ms"])
        rec = prof_excl.get(key, rec_timed)
        avg_ms = rec["ms"] / rec["n"]
        if survey:   # ranking from the fully instrumented warm-up step
            top = sorted(((k, round(v["ms"], 3)) for k, v in survey.items()), key=lambda kv: -kv[1])[:8]
            timed_all = sum(v["ms"] for v in survey.values())
        else:
            top = sorted(((k, round(v["ms"] / a.steps, 3)) for k, v in prof.items()), key=lambda kv: -kv[1])[:8]
            timed_all = sum(v["ms"] for v in prof.values()) / a.steps
        # ridge point: a GEMM is matrix-core bound when its flops/byte exceeds (MFMA peak / passes) / HBM peak
        ridge = (MFMA_BF16_PEAK_TF * 1e12 / a.passes) / (HBM_PEAK_GBS * 1e9)
        if key.startswith("gemm") and rec["flops"] / max(rec["bytes"], 1) > ridge:
            ach = rec["flops"] / (avg_ms * 1e-3) / 1e12
            out["roofline"] = {"kernel": key, "bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_BF16_PEAK_TF,
                               "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_PEAK_TF, 4), "traffic": None,
                               "avg_launch_ms": round(avg_ms, 4), "launches": rec["n"],
                               "note": "achieved counts algorithmic 2MNK flops; passes=%d bf16 MFMA products per flop" % a.passes}
        else:
            ach = rec["bytes"] / (avg_ms * 1e-3) / 1e9
            out["roofline"] = {"kernel": key, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "avg_launch_ms": round(avg_ms, 4),
                               "launches": rec["n"]}
        # HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 cannot wrap this process from
        # the inside; the file says how it was collected and is keyed by the same kernel signature)
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")) as f:
                pmc = json.load(f)["by_bench_label"].get(key)
            if pmc:
                out["roofline"]["traffic"] = pmc["hbm_bytes"]
                out["roofline"]["traffic_unit"] = "bytes/launch (2*FETCH_SIZE + WRITE_SIZE, profiles/pmc_traffic.json)"
                out["roofline"]["algorithmic_bytes"] = int(rec["bytes"])
        except (OSError, KeyError, ValueError):
            pass
        if prof_excl:
            t_ms = rec_timed["ms"] / rec_timed["n"]
            t_ach = (rec_timed["bytes"] / (t_ms * 1e-3) / 1e9) if out["roofline"]["bound"] == "hbm" else (rec_timed["flops"] / (t_ms * 1e-3) / 1e12)
            out["roofline"]["measured"] = (f"HIP events around each launch of this signature in {EXCL} extra steps right after the "
                                           "timed region, on one stream (the --serial-streams schedule): each launch alone on the chip")
            out["roofline"]["in_timed_region"] = {
                "avg_launch_ms": round(t_ms, 4), "launches": rec_timed["n"], "achieved": round(t_ach, 1),
                "frac": round(t_ach / out["roofline"]["peak"], 4),
                "note": "two-stream schedule: this launch shares HBM and CUs with the other model's kernels, so its wall "
                        "duration is longer than its exclusive one while the step as a whole is shorter"}
        out["roofline"]["top_ms_per_step"] = top
        out["roofline"]["timed_kernels_ms_per_step"] = round(timed_all, 3)
        out["roofline"]["top_ms_per_step_note"] = "exclusive durations: the instrumented warm-up step runs on one stream"
        out["roofline"]["host_enqueue_ms_per_step"] = round(host_enqueue_ms, 3)
        out["roofline"]["host_enqueue_note"] = "3 steps enqueued on an empty launch queue (no back-pressure from the GPU)"
    # ---- the composed path: raw frames + token ids -> ViT-B/16 + RoBERTa-base -> features -> the same PPO step, MEASURED in a
    # second loop of the same K steps with the same bracketing (barrier + synchronize on both sides).  Not `value`: the
    # reference's PPO loop reads pre-extracted features (finetune/ppo.py:115-148); this is the "ViT-B+RoBERTa-base" reading
    # of the BASELINE metric.  The encoders are frozen feature extractors (inference schedule, no dropout).
    if world == 1 and not a.no_online:
        from lr2ppo_amd.finetune.features import FeatureExtractor, synthetic_raw_batch
        torch.cuda.empty_cache()
        torch.manual_seed(8)
        fx = FeatureExtractor()
        fx.init_normal()
        fx = fx.to(dev).eval()
        graw = torch.Generator(device=dev).manual_seed(2000 + rank)
        raw = [synthetic_raw_batch(a.batch, a.tags, device=dev, generator=graw) for _ in range(2)]

        def ppo_step(text, img, tg):
            model.eval()
            rec = ppo.rollout_step(model, reward, text, img, tg)
            model.train()
            return ppo.update_minibatch(margs, model, opt, copt, rec, dp)

        def online_steps(n, first):
            """n composed steps: features extracted in line, then the PPO step.  (Extracting batch k+1 on a second stream
            while batch k's PPO step runs was measured: 77.08 vs 76.73 ms per step -- the encoder GEMMs hold every CU's LDS, so
            nothing runs beside them; not kept.)"""
            m_ = None
            for i in range(n):
                frames, ids, seg, tg = raw[(first + i) % len(raw)]
                text, img = fx.extract(frames, ids, seg, check_ids=False)
                m_ = ppo_step(text, img, tg)
            return m_

        m2 = online_steps(max(1, min(a.warmup, 2)), 0)
        fence()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        enc_ms = 0.0
        t0 = time.perf_counter()
        m2 = online_steps(a.steps, 2)
        fence()
        dt2 = time.perf_counter() - t0
        fx.text.embedding.check_ids()
        if not torch.isfinite(m2).all():
            raise SystemExit("bench: non-finite PPO metrics in the composed loop")
        # the extractor alone, same inputs (HIP events on the launch stream), for the MFMA-utilisation figure
        iters = 3
        ev0.record()
        for i in range(iters):
            fx.extract(*raw[i % len(raw)][:3], check_ids=False)
        ev1.record()
        torch.cuda.synchronize()
        enc_ms = ev0.elapsed_time(ev1) / iters
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
        import encoder_bench
        fl = (encoder_bench.flops(a.batch * 16, 197, first_token_only=True) + encoder_bench.flops(a.batch * a.tags, 196)
              + 2.0 * a.batch * 16 * 196 * 768 * 768)
        out["config"]["with_online_feature_extraction"] = {
            "ms_per_step": round(dt2 / a.steps * 1e3, 3), "steps_per_sec": round(a.steps / dt2, 3), "measured": True,
            "steps": a.steps,
            "workload": f"frames uint8 [{a.batch},16,3,224,224] + token ids [{a.batch},{a.tags},196] -> ViT-B/16 + RoBERTa-base "
                        "(random weights, inference) -> text_emb / img_emb -> rollout + update"}
        out["config"]["dual_encoder_forward"] = {
            "ms": round(enc_ms, 3), "algorithmic_tflop": round(fl / 1e12, 2), "tflops": round(fl / enc_ms / 1e9, 1),
            "mfma_issue_frac": round(a.passes * fl / enc_ms / 1e9 / MFMA_BF16_PEAK_TF, 4), "passes": a.passes,
            "includes": "uint8 normalise + patchify + patch projection, token embedding, 2 x 12 encoder layers, pooling; the image "
                        "stack's last layer is evaluated for the pooled [CLS] row only (keys / values for all rows) and "
                        "algorithmic_tflop counts it that way"}
        # BASELINE configs[1] as written: ViT-B/16 + RoBERTa-base in front of finetune/pointwise.py's Classifier (the Actor
        # architecture, SmoothL1, AdamW, per-step scheduler) at batch 32 x 20 tags (pointwise.sh:28): uint8 frames + token ids
        # -> features (text encoder over 640 sequences) -> one stage-1 train step at M = 125 440 token rows.  Measured like
        # the loops above (synchronize on both sides); a secondary figure, never `value`.
        if not a.no_stage1:
            from lr2ppo_amd.finetune import pointwise
            pargs = argparse.Namespace(**{**vars(margs), "train_steps": 1000, "batch_size": a.batch})
            torch.manual_seed(9)
            pmodel = pointwise.Classifier(pargs, None).to(dev)
            with torch.no_grad():
                for p in pmodel.parameters():
                    p.normal_(0, 0.02)
            popt, psch = pointwise.build_optimizer(pargs, pmodel)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                for _ in range(20):
                    psch.step()
            pmodel.train()
            praw = [synthetic_raw_batch(a.batch, 20, device=dev, generator=graw) for _ in range(2)]

            def stage1_step(i):
                frames, ids, seg, tg = praw[i % len(praw)]
                text, img = fx.extract(frames, ids, seg, check_ids=False)
                return pointwise.train_model(pargs, pmodel, popt, psch, text, img, tg)

            for i in range(2):
                l1 = stage1_step(i)
            fence()
            n1 = max(3, min(a.steps, 6))
            t0 = time.perf_counter()
            for i in range(n1):
                l1 = stage1_step(i)
            fence()
            dt1 = time.perf_counter() - t0
            if not torch.isfinite(l1):
                raise SystemExit("bench: non-finite stage-1 loss")
            out["config"]["stage1_pointwise_with_online_feature_extraction"] = {
                "ms_per_step": round(dt1 / n1 * 1e3, 3), "steps_per_sec": round(n1 / dt1, 3), "steps": n1, "measured": True,
                "workload": f"BASELINE configs[1]: frames uint8 [{a.batch},16,3,224,224] + token ids [{a.batch},20,196] -> ViT-B/16 + "
                            "RoBERTa-base (random weights, inference) -> finetune/pointwise.py train step (Actor architecture, "
                            f"{a.batch} x 20 tags, dropout on, fused out_layer.fc1 update)"}
            del pmodel, popt, psch, praw
        del fx, raw
    # ---- CPU baseline: the oracle on this box's host cores, bounded sample (one untimed warm-up step
    # that allocates the Adam state, then the measured step(s) at the benchmark batch -- no extrapolation) ----
    if world == 1 and not a.no_cpu_baseline:
        del model, reward, opt, copt, data
        torch.cuda.empty_cache()
        from oracle import cpu_baseline
        r = cpu_baseline.time_ppo_steps(a.batch, a.tags, steps=a.cpu_steps, warmup_bs=2)
        out["cpu_baseline"] = {"value": round(1.0 / r["total_s"], 5), "unit": "PPO steps/s", "cores": r["threads"], "kind": "port",
                               "sample": f"oracle (torch CPU fp32, dropout on in the update) on {r['steps']} measured PPO step(s) at "
                                         f"batch {a.batch} x {a.tags} tags after one untimed warm-up step at batch {r['warmup_bs']} "
                                         f"(Adam state allocation): rollout {r['rollout_s']:.1f}s + update fwd/bwd {r['fwd_bwd_s']:.1f}s "
                                         f"+ AdamW(1.045B) {r['adamw_s']:.1f}s = {r['total_s']:.1f}s per step"}
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

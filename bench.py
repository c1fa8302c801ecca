#!/usr/bin/env python3
"""PPO steps/s of the LR2PPO hot path on MI355X -- BASELINE.json's metric "PPO steps/sec (ViT-B+RoBERTa-base, batch 32)".

`value`: one PPO step on its own configuration = uint8 frames [32,16,3,224,224] + tag token ids [32,2,196] -> ViT-B/16 +
RoBERTa-base (random weights, frozen, the shipped configs) -> text_emb / img_emb -> one rollout timestep (actor + critic + reward
no-grad forwards, finetune/ppo.py:844-883) -> one update minibatch (actor fwd/bwd/AdamW + critic fwd/bwd/AdamW with the fused PPO
loss, finetune/ppo.py:518-587, dropout on as under model.train()), 32 items x 2 tags per GPU, everything resident in HBM.
`head_only_steps_per_sec` (top level): the same rollout + update on PRE-EXTRACTED features, which is what the reference's own PPO
loop runs (it reads clean_feat.h5, finetune/ppo.py:115-148) -- rounds 1-2 reported that one as `value`.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line on rank 0.  `roofline`: the kernel signature with the most device time in the `value` loop (the ViT FFN-1
GEMM of the 256 x 256 MFMA kernel), HIP events on its launch stream inside the timed region; `roofline_hbm`: the dominant
HBM-bound kernel of the head (fused out_layer.fc1 weight gradient + AdamW).  Single-GPU extras (top-level scalars + `config`):
dual-encoder forward and forward+backward (MFMA issue fraction), BASELINE configs[1] (stage-1 pointwise at 32 x 20 tags) with
frozen and with fine-tuned encoders.  `cpu_baseline`: the CPU oracle ("port") on this box's host cores, bounded sample.
"""
import argparse
import json
import os
import sys
import time
import warnings

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))

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
    ap.add_argument("--cpu-steps", type=int, default=3,
                    help="measured CPU-baseline steps at --batch (after one untimed warm-up step); the median is reported (BASELINE.md 3)")
    ap.add_argument("--no-config5", action="store_true", help="skip the BASELINE configs[4] figures (ViT-L/14 tower + projection, MX-FP8)")
    ap.add_argument("--no-online", action="store_true",
                    help="head-only run: skip the composed loop (then `value` is the head-only rate and the line says so)")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel HIP events in the timed regions")
    ap.add_argument("--no-extras", action="store_true", help="skip the single-GPU extras (encoder fwd / fwd+bwd, stage-1 figures)")
    ap.add_argument("--no-stage1", action="store_true", help="skip the BASELINE configs[1] figures (stage-1 step at 32 x 20 tags)")
    ap.add_argument("--no-graph", action="store_true", help="skip the HIP-graph replay of the head-only step (head_only_graph)")
    ap.add_argument("--serial-streams", action="store_true",
                    help="one HIP stream for the whole PPO step (LR2_PPO_STREAMS=0): every launch runs alone, so a kernel trace "
                         "of this command shows exclusive per-kernel durations; the default schedule runs the critic beside the actor")
    return ap.parse_args()


class exclusive_launches:
    """Inside: the PPO step on ONE stream (what LR2_PPO_STREAMS=0 / --serial-streams selects), so that HIP events around a
    launch time that launch alone.  Under the two-stream schedule the critic's launches share the chip with the actor's and a
    launch's wall duration is not a measurement of its bandwidth."""

    def __enter__(self):
        self.old = os.environ.get("LR2_PPO_STREAMS")
        os.environ["LR2_PPO_STREAMS"] = "0"

    def __exit__(self, *exc):
        if self.old is None:
            os.environ.pop("LR2_PPO_STREAMS", None)
        else:
            os.environ["LR2_PPO_STREAMS"] = self.old


def _roofline(key, rec, passes, note=None):
    """roofline object of one kernel signature from its HIP-event record {"ms", "n", "flops", "bytes"}."""
    avg_ms = rec["ms"] / rec["n"]
    ridge = (MFMA_BF16_PEAK_TF * 1e12 / passes) / (HBM_PEAK_GBS * 1e9)       # flops/byte above which a GEMM is matrix-core bound
    if key.startswith(("gemm", "selfattn")) and rec["flops"] / max(rec["bytes"], 1) > ridge:
        alg = rec["flops"] / (avg_ms * 1e-3) / 1e12
        ach = passes * alg if key.startswith("gemm") else alg
        return {"kernel": key, "bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                "frac": round(ach / MFMA_BF16_PEAK_TF, 4), "frac_algorithmic": round(alg / MFMA_BF16_PEAK_TF, 4), "traffic": None,
                "avg_launch_ms": round(avg_ms, 4), "launches": rec["n"],
                "algorithmic_tflops_fp32_equivalent": round(alg, 1), "flops_per_launch": int(passes * rec["flops"]),
                "algorithmic_flops_per_launch": int(rec["flops"]),
                "note": note or ("achieved = bf16 MFMA flops issued per launch (%d split-bf16 products x 2MNK, the algorithm's own "
                                 "count) / average launch duration; frac_algorithmic = (2MNK / duration) / the same bf16 peak -- SURVEY "
                                 "8(d)'s count, i.e. the useful fp32-grade flops (the chip's native fp32-matrix peak is 157 TFLOP/s = "
                                 "0.063 on this scale)" % passes)}
    ach = rec["bytes"] / (avg_ms * 1e-3) / 1e9
    return {"kernel": key, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "avg_launch_ms": round(avg_ms, 4), "launches": rec["n"]}


def _pmc(key):
    try:
        with open(os.path.join(REPO, "profiles", "pmc_traffic.json")) as f:
            return json.load(f)["by_bench_label"].get(key)
    except (OSError, KeyError, ValueError):
        return None


def launcher_command(gpus: int, argv, script: str, port: int):
    """What `python bench.py --gpus N` (N > 1) runs when no launcher set WORLD_SIZE: the reference's own way of starting ranks
    (ppo.sh:59: torchrun, one process per GPU), on this node, rendezvous on 127.0.0.1."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), script] + list(argv)


def self_launch(gpus: int, argv, script: str = os.path.abspath(__file__)) -> int:
    """Start `gpus` ranks of `script` as a FRESH child process tree (this process has made no GPU call and makes none: it never
    re-execs, it only waits), relay rank 0's JSON line, and fail loudly -- non-zero exit -- when a rank fails, when no line comes
    back, or when the line does not report n_gpus == gpus."""
    import socket
    import subprocess
    import time
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), LR2_BENCH_CHILD="1")
    for attempt in range(2):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        t0 = time.time()
        p = subprocess.run(launcher_command(gpus, argv, script, port), env=env, stdout=subprocess.PIPE, text=True)
        # a rendezvous that dies within seconds (the free port was taken between the probe above and the launcher's bind) is started
        # once more on another port; a run that fails after it got going is a failure
        if p.returncode == 0 or time.time() - t0 > 20.0 or attempt == 1:
            break
        print(f"[bench] the launcher exited with code {p.returncode} after {time.time() - t0:.1f} s: one more try on another port", file=sys.stderr)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if p.returncode != 0:
        print(f"[bench] the {gpus}-rank run failed (exit code {p.returncode})", file=sys.stderr)
        return p.returncode or 1
    try:
        n = json.loads(line)["n_gpus"] if line else None
    except (ValueError, KeyError):
        n = None
    if n != gpus:
        print(f"[bench] asked for {gpus} ranks, the result line reports n_gpus = {n}: refusing to pass it on", file=sys.stderr)
        return 3
    print(line)
    return 0


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher (before anything touches the GPU)
        raise SystemExit(self_launch(a.gpus, sys.argv[1:]))
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
        # a collective that never completes (a rank that died, a schedule mismatch) aborts the job after this long -- with an error in
        # the log and a non-zero exit -- instead of hanging until somebody kills it (this path's first multi-rank runs are the driver's)
        import datetime
        dist.init_process_group(os.environ.get("LR2_BENCH_BACKEND", "nccl"), rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=int(os.environ.get("LR2_BENCH_PG_TIMEOUT_S", "300"))))
    if a.gpus != world:
        raise SystemExit(f"bench: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    ranks_seen = 1
    if world > 1:        # every rank really is in the job: an all-reduce of ones over the product's backend
        probe = torch.ones(1, device=dev)
        dist.all_reduce(probe)
        ranks_seen = int(probe.item())
        if ranks_seen != dist.get_world_size():
            raise SystemExit(f"bench: the all-reduce probe saw {ranks_seen} ranks of {dist.get_world_size()}")

    if a.serial_streams:
        os.environ["LR2_PPO_STREAMS"] = "0"
    from lr2ppo_amd import _native
    if rank == 0:
        _native.build()
    if world > 1:
        dist.barrier()
    from lr2ppo_amd import ops, runtime
    from lr2ppo_amd.finetune import ppo
    from lr2ppo_amd.finetune.features import FeatureExtractor, synthetic_raw_batch

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
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(20):                        # leave the lr-0 first cycle (quirk 15): lr = 20/100 of 1e-3
            sch.step(), csch.step()
    model.actor.bind_grads(), model.critic.bind_grads()
    runtime.set_dropout_seed(1234 + rank)
    dp = ppo._DataParallel()
    two_streams = ppo._Side(dev).on

    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    n_batches = 4
    data = [(torch.randn(a.batch, a.tags, 196, 768, device=dev, generator=g),
             torch.randn(a.batch, 16, 768, device=dev, generator=g),          # shared by the tags of an item
             torch.randint(0, 3, (a.batch, a.tags), device=dev, generator=g)) for _ in range(n_batches)]

    def ppo_step(text, img, tgts):
        model.eval()
        rec = ppo.rollout_step(model, reward, text, img, tgts)
        model.train()
        return ppo.update_minibatch(margs, model, opt, copt, rec, dp)

    def step(i):
        return ppo_step(*data[i % n_batches])

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(dt):
        if world == 1:
            return dt
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ================= [A] head-only loop (pre-extracted features: the reference's real PPO loop) =================
    # Per-kernel HIP events: every GEMM / AdamW signature during the LAST warm-up step (ranking, `top_ms_per_step`); inside the
    # timed region only the dominant signature is bracketed, so that the measurement does not slow the thing it measures (two
    # event records per launch on ~120 launches made the step host-bound).
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
    # Host cost of enqueueing one step, measured on an EMPTY launch queue (3 steps after a synchronise): inside the timed loop
    # the host runs ahead until the HIP launch queue is full and then spins on back-pressure.
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
    fence()
    head_dt = time.perf_counter() - t0
    prof = ops.profile_stop() if not a.no_profile else {}
    if not torch.isfinite(m).all():
        raise SystemExit("bench: non-finite PPO metrics")
    # the dominant signature again, each launch ALONE on the chip: EXCL extra steps on one stream right after the timed region
    prof_excl, EXCL = {}, 4
    if prof and two_streams:
        with exclusive_launches():
            step(a.warmup + a.steps)
            fence()
            ops.profile_start(only=[dominant])
            for i in range(EXCL):
                step(a.warmup + a.steps + 1 + i)
            fence()
            prof_excl = ops.profile_stop()
    head_dt = max_over_ranks(head_dt)
    # the ONE-stream schedule, eager: what world > 1 runs by default, so that the N = 1 -> N = 2 step of a scaling
    # curve can be read against the same schedule at N = 1
    one_stream_ms = None
    if two_streams:
        with exclusive_launches():
            step(0)
            fence()
            t0 = time.perf_counter()
            for i in range(a.steps):
                step(1 + i)
            fence()
            one_stream_ms = max_over_ranks(time.perf_counter() - t0) / a.steps * 1e3

    # the same head-only step captured in a HIP graph (ppo.GraphedPPOStep: dropout seeds and learning rates in device memory),
    # two-stream schedule and one-stream schedule: host cost of one replay on an empty queue, and the step time
    graphed = None
    if world == 1 and not a.no_graph:
        def graph_measure():
            gstep = ppo.GraphedPPOStep(margs, model, reward, opt, copt)
            for i in range(3):                     # eager (sizes the workspaces), capture + replay, replay
                gm = gstep(*data[i % n_batches])
            fence()
            t0 = time.perf_counter()
            for i in range(3):
                gstep(*data[i % n_batches])
            host = (time.perf_counter() - t0) / 3 * 1e3
            fence()
            t0 = time.perf_counter()
            for i in range(a.steps):
                gm = gstep(*data[i % n_batches])
            fence()
            dt = (time.perf_counter() - t0) / a.steps * 1e3
            if not torch.isfinite(gm).all():
                raise SystemExit("bench: non-finite PPO metrics in the graphed step")
            gstep.release()
            return {"host_enqueue_ms_per_step": round(host, 3), "ms_per_step": round(dt, 3)}
        graphed = graph_measure()
        with exclusive_launches():
            graphed["one_stream"] = graph_measure()
        graphed["note"] = ("head-only step as ONE hipGraphLaunch (inputs copied into static buffers + one 64-byte scalar store + "
                           "replay); same bits as the eager step (tests/test_graph_gpu.py); `value` and head_only_* are the eager step")

    # ================= [B] the metric's own configuration: ViT-B/16 + RoBERTa-base in line (frozen), then the PPO step =================
    online = None
    if not a.no_online:
        torch.manual_seed(8)
        fx = FeatureExtractor()
        fx.init_normal()
        fx = fx.to(dev).eval()
        graw = torch.Generator(device=dev).manual_seed(2000 + rank)
        raw = [synthetic_raw_batch(a.batch, a.tags, device=dev, generator=graw) for _ in range(2)]

        def online_step(i):
            """features extracted in line, then the PPO step.  (Extracting batch k+1 on a second stream while batch k's PPO step
            runs was measured: 77.08 vs 76.73 ms per step -- the encoder GEMMs hold every CU's LDS; not kept.)"""
            frames, ids, seg, tg = raw[i % len(raw)]
            text, img = fx.extract(frames, ids, seg, check_ids=False)
            return ppo_step(text, img, tg)

        osurvey = {}
        nw = max(2, min(a.warmup, 3))
        for i in range(nw):
            if i == nw - 1 and not a.no_profile:
                fence()
                ops.profile_start()
                with exclusive_launches():
                    m2 = online_step(i)
                osurvey = ops.profile_stop()
            else:
                m2 = online_step(i)
        fence()
        odom = max(osurvey.items(), key=lambda kv: kv[1]["ms"])[0] if osurvey else None
        if odom is not None:
            ops.profile_start(only=[odom])
        t0 = time.perf_counter()
        for i in range(a.steps):
            m2 = online_step(nw + i)
        fence()
        dt = time.perf_counter() - t0
        oprof = ops.profile_stop() if odom is not None else {}
        fx.text.embedding.check_ids()
        if not torch.isfinite(m2).all():
            raise SystemExit("bench: non-finite PPO metrics in the composed loop")
        online = {"dt": max_over_ranks(dt), "prof": oprof, "survey": osurvey, "dominant": odom}

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    sched = ("critic forward / backward / optimizer step on a second HIP stream beside the actor's" if two_streams
             else "one HIP stream")
    head_rate = world * a.steps / head_dt
    if online is not None:
        dt = online["dt"]
        value, ms_per_step = world * a.steps / dt, dt / a.steps * 1e3
        unit = ("PPO steps/s (1 step = ViT-B/16 over 32 x 16 frames + RoBERTa-base over 32 x 2 tag sequences, frozen, in line -> "
                "1 rollout batch + 1 update minibatch, 32 items x 2 tags per GPU; head_only_steps_per_sec = the same step on "
                "pre-extracted features)")
        workload = ("BASELINE metric 'PPO steps/sec (ViT-B+RoBERTa-base, batch 32)': frames uint8 [32,16,3,224,224] + token ids "
                    "[32,2,196] -> ViT-B/16 + RoBERTa-base (random weights, inference) -> text_emb [32,2,196,768], img_emb [32,16,768] "
                    "-> LR2PPO stage-3 rollout + update (actor 519M + critic 526M + reward 526M params)")
    else:
        value, ms_per_step = head_rate, head_dt / a.steps * 1e3
        unit = "PPO steps/s, HEAD ONLY (--no-online): 1 rollout batch + 1 update minibatch on pre-extracted features, 32 items x 2 tags per GPU"
        workload = "LR2PPO stage-3 head-only PPO step on LRMovieNet-shaped synthetic features: text_emb [32,2,196,768], img_emb [32,16,768]"
    out = {
        "metric": "ppo_steps_per_sec", "value": round(value, 3), "unit": unit, "n_gpus": world, "rccl_ranks_seen": ranks_seen,
        "backend": (dist.get_backend() if world > 1 else None), "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (GEMMs: split-bf16 x3 on MFMA, fp32 accumulate)" if a.passes == 3 else "bf16 inputs, fp32 accumulate (1 pass)",
        "data": "synthetic",
        "head_only_steps_per_sec": round(head_rate, 3), "head_only_ms_per_step": round(head_dt / a.steps * 1e3, 3),
        "head_only_one_stream_ms": None if one_stream_ms is None else round(one_stream_ms, 3),
        "head_only_graph": graphed,
        "config": {"workload": workload, "batch_per_gpu": a.batch, "tags": a.tags, "global_batch": a.batch * world,
                   "parallelism": f"dp{world}", "schedule": sched, "items_per_sec": round(value * a.batch, 1),
                   "algorithmic_tflop_per_step": {"head": 3.44, "dual_encoder_forward": 18.96} if (a.batch, a.tags) == (32, 2) else None},
    }
    # ---- roofline of the `value` loop's dominant kernel signature (measured inside its timed region; the encoder launches run
    # alone on the main stream: the side stream only carries the critic inside the PPO step) ----
    if online is not None and online["prof"]:
        key, rec = max(online["prof"].items(), key=lambda kv: kv[1]["ms"])
        out["roofline"] = _roofline(key, rec, a.passes)
        out["roofline"]["ms_per_step"] = round(rec["ms"] / a.steps, 3)
        sv = online["survey"]
        out["roofline"]["top_ms_per_step"] = sorted(((k, round(v["ms"], 3)) for k, v in sv.items()), key=lambda kv: -kv[1])[:8]
        out["roofline"]["timed_kernels_ms_per_step"] = round(sum(v["ms"] for v in sv.values()), 3)
        out["roofline"]["top_ms_per_step_note"] = ("GEMM / attention / LayerNorm / AdamW signatures of one instrumented warm-up step "
                                                   "of the value loop, one stream (exclusive durations)")
        pm = _pmc(key)
        if pm and "hbm_bytes" in pm:
            out["roofline"]["traffic"] = pm["hbm_bytes"]
            out["roofline"]["traffic_unit"] = "HBM bytes/launch (2*FETCH_SIZE + WRITE_SIZE, separate PMC passes over the encoder forward, profiles/pmc_traffic.json)"
            out["roofline"]["algorithmic_bytes"] = int(rec["bytes"])
        if pm and "mfma_busy" in pm:
            out["roofline"]["mfma_busy_pmc"] = pm["mfma_busy"]
            out["roofline"]["mfma_busy_note"] = "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs) for this signature, profiles/pmc_traffic.json"
    # ---- the head's dominant HBM-bound signature (fused out_layer.fc1 weight gradient + AdamW) ----
    if prof:
        key, rec_timed = max(prof.items(), key=lambda kv: kv[1]["ms"])
        rec = prof_excl.get(key, rec_timed)
        rl = _roofline(key, rec, a.passes)
        pmc = _pmc(key)
        if pmc and "hbm_bytes" in pmc:
            rl["traffic"] = pmc["hbm_bytes"]
            rl["traffic_unit"] = "bytes/launch (2*FETCH_SIZE + WRITE_SIZE, profiles/pmc_traffic.json)"
            rl["algorithmic_bytes"] = int(rec["bytes"])
        if prof_excl:
            t_ms = rec_timed["ms"] / rec_timed["n"]
            t_ach = (rec_timed["bytes"] / (t_ms * 1e-3) / 1e9) if rl["bound"] == "hbm" else (a.passes * rec_timed["flops"] / (t_ms * 1e-3) / 1e12)
            rl["measured"] = (f"HIP events around each launch of this signature in {EXCL} extra head-only steps right after the timed "
                              "region, on one stream (the --serial-streams schedule): each launch alone on the chip")
            rl["in_timed_region"] = {"avg_launch_ms": round(t_ms, 4), "launches": rec_timed["n"], "achieved": round(t_ach, 1),
                                     "frac": round(t_ach / rl["peak"], 4),
                                     "note": "two-stream schedule: this launch shares HBM and CUs with the other model's kernels"}
        if survey:
            rl["top_ms_per_step"] = sorted(((k, round(v["ms"], 3)) for k, v in survey.items()), key=lambda kv: -kv[1])[:8]
            rl["timed_kernels_ms_per_step"] = round(sum(v["ms"] for v in survey.values()), 3)
            rl["top_ms_per_step_note"] = "head-only step; exclusive durations: the instrumented warm-up step runs on one stream"
        rl["host_enqueue_ms_per_step"] = round(host_enqueue_ms, 3)
        rl["host_enqueue_note"] = "head-only step, 3 steps enqueued on an empty launch queue (no back-pressure from the GPU)"
        if graphed is not None:
            rl["host_enqueue_ms_per_step_graph"] = graphed["host_enqueue_ms_per_step"]
            rl["host_enqueue_ms_per_step_graph_one_stream"] = graphed["one_stream"]["host_enqueue_ms_per_step"]
        out["roofline_hbm"] = rl
        if "roofline" not in out:
            out["roofline"] = rl

    # ================= [C] single-GPU extras =================
    if world == 1 and online is not None and not a.no_extras:
        import encoder_bench
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # the extractor alone, same inputs (HIP events on the launch stream), for the MFMA-utilisation figure
        iters = 3
        ev0.record()
        for i in range(iters):
            fx.extract(*raw[i % len(raw)][:3], check_ids=False)
        ev1.record()
        torch.cuda.synchronize()
        enc_ms = ev0.elapsed_time(ev1) / iters
        fl_pruned = (encoder_bench.flops(a.batch * 16, 197, first_token_only=True) + encoder_bench.flops(a.batch * a.tags, 196)
                     + 2.0 * a.batch * 16 * 196 * 768 * 768)
        out["dual_encoder_forward_ms"] = round(enc_ms, 3)
        out["dual_encoder_forward_mfma_frac"] = round(a.passes * fl_pruned / enc_ms / 1e9 / MFMA_BF16_PEAK_TF, 4)
        out["config"]["dual_encoder_forward"] = {
            "ms": round(enc_ms, 3), "algorithmic_tflop": round(fl_pruned / 1e12, 2), "tflops": round(fl_pruned / enc_ms / 1e9, 1),
            "mfma_issue_frac": out["dual_encoder_forward_mfma_frac"], "passes": a.passes,
            "includes": "uint8 normalise + patchify + patch projection, token embedding, 2 x 12 encoder layers, pooling; the image "
                        "stack's last layer is evaluated for the pooled [CLS] row only (keys / values for all rows) and "
                        "algorithmic_tflop counts it that way"}
        # forward + backward of both stacks in train mode (dropout 0.1 at every reference site), every parameter gradient written:
        # FeatureExtractor.forward_train / backward_train, the schedule finetune_pointwise_step drives from a head's loss
        fx.train()
        fx.bind_grads()
        gd = torch.Generator(device=dev).manual_seed(3000)
        d_text = torch.randn(a.batch, a.tags, 196, 768, device=dev, generator=gd) * 1e-3
        d_img = torch.randn(a.batch, 16, 768, device=dev, generator=gd) * 1e-3

        def enc_train(i):
            frames, ids, seg, _ = raw[i % len(raw)]
            t_, i_, ctx = fx.forward_train(frames, ids, seg)
            fx.backward_train(ctx, d_text, d_img)

        for i in range(2):            # two untimed passes: the first process on a fresh box needs more than one to settle (arenas, clocks)
            enc_train(i)
        torch.cuda.synchronize()
        n_tr = 3
        ev0.record()
        for i in range(n_tr):
            enc_train(2 + i)
        ev1.record()
        torch.cuda.synchronize()
        tr_ms = ev0.elapsed_time(ev1) / n_tr
        fl_full = (encoder_bench.flops(a.batch * 16, 197) + encoder_bench.flops(a.batch * a.tags, 196)
                   + 2.0 * a.batch * 16 * 196 * 768 * 768)
        fl_train = 3.0 * fl_full - 2.0 * a.batch * 16 * 196 * 768 * 768          # the patch projection has no input gradient
        out["dual_encoder_train_ms"] = round(tr_ms, 3)
        out["dual_encoder_train_mfma_frac"] = round(a.passes * fl_train / tr_ms / 1e9 / MFMA_BF16_PEAK_TF, 4)
        out["config"]["dual_encoder_train"] = {
            "ms": round(tr_ms, 3), "algorithmic_tflop": round(fl_train / 1e12, 2), "tflops": round(fl_train / tr_ms / 1e9, 1),
            "mfma_issue_frac": out["dual_encoder_train_mfma_frac"], "passes": a.passes,
            "workload": f"ViT-B/16 over {a.batch * 16} frames + RoBERTa-base over {a.batch * a.tags} sequences, forward (saving, dropout "
                        "0.1) + hand-written backward incl. embeddings: 3 x the forward's matrix flops (dgrad + wgrad)"}
        # stage 3 with the encoders trained THROUGH the PPO step (features.finetune_ppo_step: rollout on eval-mode features, update on
        # train-mode features, the heads hand d (policy + value loss) / d features back, encoder + embedding backward, AdamW over both
        # stacks): the metric's step plus a second (training) pass of both stacks and their backward -- single rank only
        if world == 1 and not a.no_stage1:
            from lr2ppo_amd.finetune.features import build_encoder_optimizer, finetune_ppo_step
            eargs = argparse.Namespace(**{**vars(margs), "train_steps": 1000, "batch_size": a.batch})
            eopt3, esch3 = build_encoder_optimizer(eargs, fx)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                for _ in range(20):
                    esch3.step()

            def stage3_finetune(i):
                frames, ids, seg, tg = raw[i % len(raw)]
                return finetune_ppo_step(margs, fx, model, reward, opt, copt, eopt3, frames, ids, seg, tg)

            for i in range(2):
                m3 = stage3_finetune(i)
            fence()
            n3 = max(3, min(a.steps, 5))
            t0 = time.perf_counter()
            for i in range(n3):
                m3 = stage3_finetune(i)
            fence()
            dt3 = time.perf_counter() - t0
            if not torch.isfinite(m3).all():
                raise SystemExit("bench: non-finite PPO metrics in the stage-3 fine-tune step")
            fx.text.embedding.check_ids()
            out["stage3_finetune_steps_per_sec"] = round(n3 / dt3, 3)
            out["config"]["stage3_ppo_with_encoder_finetune"] = {
                "ms_per_step": round(dt3 / n3 * 1e3, 3), "steps_per_sec": round(n3 / dt3, 3), "steps": n3, "measured": True,
                "workload": f"the metric's PPO step ({a.batch} items x {a.tags} tags) with ViT-B/16 + RoBERTa-base TRAINED through it: "
                            "eval-mode extraction -> rollout; train-mode extraction (activations kept) -> update with input gradients -> "
                            "encoder + embedding backward -> AdamW over both stacks (beyond the reference, which trains stage 3 on "
                            "pre-extracted features)"}
            del eopt3, esch3
        fx.eval()
        torch.cuda.empty_cache()
        # BASELINE configs[1] as written: ViT-B/16 + RoBERTa-base in front of finetune/pointwise.py's Classifier (the Actor
        # architecture, SmoothL1, AdamW, per-step scheduler) at batch 32 x 20 tags (pointwise.sh:28): (i) encoders frozen
        # (feature extraction in line), (ii) encoders fine-tuned end to end from the head's loss (finetune_pointwise_step).
        if not a.no_stage1:
            from lr2ppo_amd.finetune import pointwise
            from lr2ppo_amd.finetune.features import build_encoder_optimizer, finetune_pointwise_step
            pargs = argparse.Namespace(**{**vars(margs), "train_steps": 1000, "batch_size": a.batch})
            torch.manual_seed(9)
            pmodel = pointwise.Classifier(pargs, None).to(dev)
            with torch.no_grad():
                for p in pmodel.parameters():
                    p.normal_(0, 0.02)
            popt, psch = pointwise.build_optimizer(pargs, pmodel)
            eopt, esch = build_encoder_optimizer(pargs, fx)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                for _ in range(20):
                    psch.step(), esch.step()
            pmodel.train()
            praw = [synthetic_raw_batch(a.batch, 20, device=dev, generator=graw) for _ in range(2)]

            def stage1_frozen(i):
                frames, ids, seg, tg = praw[i % len(praw)]
                text, img = fx.extract(frames, ids, seg, check_ids=False)
                return pointwise.train_model(pargs, pmodel, popt, psch, text, img, tg)

            def stage1_finetune(i):
                frames, ids, seg, tg = praw[i % len(praw)]
                return finetune_pointwise_step(pargs, fx, pmodel, popt, psch, eopt, esch, frames, ids, seg, tg)

            for tag, fn, label in (("stage1_frozen", stage1_frozen, "inference"), ("stage1_finetune", stage1_finetune, "TRAINED")):
                if tag == "stage1_finetune":
                    fx.train()
                for i in range(2):
                    l1 = fn(i)
                fence()
                n1 = max(3, min(a.steps, 5))
                t0 = time.perf_counter()
                for i in range(n1):
                    l1 = fn(i)
                fence()
                dt1 = time.perf_counter() - t0
                if not torch.isfinite(l1):
                    raise SystemExit(f"bench: non-finite stage-1 loss ({tag})")
                out[tag + "_steps_per_sec"] = round(n1 / dt1, 3)
                out["config"]["stage1_pointwise_with_online_feature_extraction" if tag == "stage1_frozen"
                              else "stage1_pointwise_with_encoder_finetune"] = {
                    "ms_per_step": round(dt1 / n1 * 1e3, 3), "steps_per_sec": round(n1 / dt1, 3), "steps": n1, "measured": True,
                    "workload": f"BASELINE configs[1]: frames uint8 [{a.batch},16,3,224,224] + token ids [{a.batch},20,196] -> ViT-B/16 + "
                                f"RoBERTa-base (random weights, {label}) -> finetune/pointwise.py train step (Actor architecture, "
                                f"{a.batch} x 20 tags, dropout on, fused out_layer.fc1 update)"
                                + ("; encoder + embedding backward from the head's input gradients, AdamW over both stacks (2 x 85 M + "
                                   "39.7 M parameters)" if tag == "stage1_finetune" else "")}
            fx.text.embedding.check_ids()
            del pmodel, popt, psch, eopt, esch, praw
            torch.cuda.empty_cache()
            # BASELINE configs[4], the parts that exist at fp32-grade numerics: (i) one stage-2 reward-pair step (finetune/
            # reward_pair_dataloader.py:347-365, the launcher's batch 64 x 20 tags, chosen / reject orderings of 4 positions each:
            # the Reward architecture's trunk over 2 x 64 x 4 = 512 pairs, M = 100 352 token rows) on pre-extracted features as
            # upstream; (ii) the ViT-L/14 encoder swap (lr2ppo_amd/configs/vit_large_14_224.json) forward over the step's 512 frames.
            from lr2ppo_amd.finetune import reward_pair_dataloader as rp
            torch.manual_seed(10)
            rmodel = rp.Classifier(pargs, None).to(dev)
            with torch.no_grad():
                for p in rmodel.parameters():
                    p.normal_(0, 0.02)
            ropt, rsch = rp.build_optimizer(pargs, rmodel)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                for _ in range(20):
                    rsch.step()
            rmodel.train()
            gs = torch.Generator(device=dev).manual_seed(4000)
            rb, rt = 64, 20
            rtext = torch.randn(rb, rt, 196, 768, device=dev, generator=gs)
            rimg = torch.randn(rb, 16, 768, device=dev, generator=gs)
            rtg = torch.randint(0, 3, (rb, rt), device=dev, generator=gs)
            chosen = torch.stack([torch.randperm(rt, device=dev, generator=gs)[:4] for _ in range(rb)])
            reject = chosen.flip(1)
            for _ in range(2):
                l2, acc = rp.train_model(pargs, rmodel, ropt, rsch, rtext, rimg, rtg, chosen, reject)
            fence()
            n2 = 4
            t0 = time.perf_counter()
            for _ in range(n2):
                l2, acc = rp.train_model(pargs, rmodel, ropt, rsch, rtext, rimg, rtg, chosen, reject)
            fence()
            dt2 = time.perf_counter() - t0
            if not torch.isfinite(l2):
                raise SystemExit("bench: non-finite stage-2 loss")
            out["stage2_reward_pair_steps_per_sec"] = round(n2 / dt2, 3)
            out["config"]["stage2_reward_pair_step"] = {
                "ms_per_step": round(dt2 / n2 * 1e3, 3), "steps": n2, "measured": True,
                "workload": "finetune/reward_pair_dataloader.py train step on pre-extracted features: 64 items x 20 tags, chosen / reject "
                            "orderings of 4 tags -> Reward-architecture trunk over 512 pairs (M = 100352 token rows), pair hinge, "
                            "fused out_layer.fc1 update"}
            del rmodel, ropt, rsch, rtext, rimg
            torch.cuda.empty_cache()
        if not a.no_stage1:
            from lr2ppo_amd.finetune.features import EncoderStack, encoder_args
            cfg_l = os.path.join(REPO, "lr2ppo_amd", "configs", "vit_large_14_224.json")
            vl = EncoderStack(encoder_args(cfg_l), 10)
            with torch.no_grad():
                for n_, p in vl.named_parameters():
                    if "gamma" not in n_ and "beta" not in n_:
                        p.normal_(0, 0.02)
            vl = vl.to(dev).eval()
            nfr = a.batch * 16
            limg = torch.randn(nfr, 3, 224, 224, device=dev)
            lseg = torch.ones(nfr, 257, dtype=torch.long, device=dev)
            with torch.no_grad():
                vl(limg, lseg)
                torch.cuda.synchronize()
                ev0.record()
                for _ in range(2):
                    vl(limg, lseg)
                ev1.record()
                torch.cuda.synchronize()
            l_ms = ev0.elapsed_time(ev1) / 2
            l_fl = 24 * (2.0 * nfr * 257 * 1024 * (4 * 1024 + 2 * 4096) + 4.0 * nfr * 16 * 257 * 257 * 64) + 2.0 * nfr * 256 * 640 * 1024
            out["vit_l14_forward_ms"] = round(l_ms, 3)
            out["vit_l14_forward_mfma_frac"] = round(a.passes * l_fl / l_ms / 1e9 / MFMA_BF16_PEAK_TF, 4)
            out["config"]["vit_l14_forward"] = {"ms": round(l_ms, 3), "algorithmic_tflop": round(l_fl / 1e12, 1), "frames": nfr,
                                                "workload": "ViT-L/14 (hidden 1024, 24 layers, 16 heads, 257 tokens: key-block attention) "
                                                            "full forward, random weights"}
            # the same stack with its projections as MX-FP8 products (BASELINE configs[4] "fp8 MFMA"; csrc/fp8.hip): an explicit,
            # non-parity fast mode -- time of the encoder stack alone (embedding excluded in both figures) and its distance
            # from the split-bf16 forward
            with torch.no_grad():
                lemb = vl.embedding(limg, lseg)
                ref = vl.encoder(lemb, lseg)
                got = vl.encoder.forward_fp8(lemb, lseg)
                rel = float((got - ref).norm() / ref.norm())
                torch.cuda.synchronize()
                ev0.record()
                for _ in range(2):
                    vl.encoder.forward_fp8(lemb, lseg)
                ev1.record()
                torch.cuda.synchronize()
                f8_ms = ev0.elapsed_time(ev1) / 2
                ev0.record()
                for _ in range(2):
                    vl.encoder(lemb, lseg)
                ev1.record()
                torch.cuda.synchronize()
                b3_ms = ev0.elapsed_time(ev1) / 2
            out["config"]["vit_l14_encoder_mxfp8"] = {
                "ms": round(f8_ms, 3), "split_bf16_x3_ms": round(b3_ms, 3), "speedup": round(b3_ms / f8_ms, 3),
                "relative_l2_distance_from_the_split_bf16_output": round(rel, 4),
                "note": "24 encoder layers over 512 x 257 tokens; QKV / output / FFN projections as MX-FP8 products on "
                        "v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 elements: NOT within north_star's 1e-3 -- a separate mode, never "
                        "part of `value`), LayerNorm / attention / residual stream as in the default path"}
            del vl, limg, lseg, lemb, ref, got
        # BASELINE configs[4] COMPOSED: "Reward-pair training + PPO, ViT-L/14 encoder swap, fp8 MFMA" -- the image tower is ViT-L/14
        # (24 layers, 257 tokens) ending in the 1024 -> 768 visual projection, the text tower RoBERTa-base, both frozen and in line,
        # every encoder projection an MX-FP8 product (FeatureExtractor(precision="mxfp8")); behind them (i) the PPO step of `value`
        # (32 items x 2 tags) and (ii) one stage-2 reward-pair step at the launcher's batch (reward_pair_dataloader.sh:21: 64 items,
        # 2 tags per training item).  The same two steps with the towers in split-bf16 are timed beside them.
        if not a.no_config5:
            from lr2ppo_amd.finetune import reward_pair_dataloader as rp
            from lr2ppo_amd.finetune.features import VIT_L14_CONFIG, encoder_args
            torch.manual_seed(12)
            fx5 = FeatureExtractor(encoder_args(VIT_L14_CONFIG), precision="mxfp8")
            fx5.init_normal()
            fx5 = fx5.to(dev).eval()
            pargs5 = argparse.Namespace(**{**vars(margs), "train_steps": 1000, "batch_size": 64})
            rmodel = rp.Classifier(pargs5, None).to(dev)
            with torch.no_grad():
                for p in rmodel.parameters():
                    p.normal_(0, 0.02)
            ropt, rsch = rp.build_optimizer(pargs5, rmodel)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                for _ in range(20):
                    rsch.step()
            rmodel.train()
            g5 = torch.Generator(device=dev).manual_seed(5000)
            raw64 = synthetic_raw_batch(64, 2, device=dev, generator=g5)
            lay = torch.tensor([[0, 1, 0, 1], [1, 0, 0, 1]], device=dev)[torch.randint(0, 2, (64,), device=dev, generator=g5)]
            chosen5, reject5 = lay, torch.cat([lay[:, :2], lay[:, 2:].flip(1)], dim=1)

            def c5_ppo(i):
                frames, ids, seg, tg = raw[i % len(raw)]
                text, img = fx5.extract(frames, ids, seg, check_ids=False)
                return ppo_step(text, img, tg)

            def c5_pair(i):
                frames, ids, seg, tg = raw64
                text, img = fx5.extract(frames, ids, seg, check_ids=False)
                return rp.train_model(pargs5, rmodel, ropt, rsch, text, img, tg, chosen5, reject5)[0]

            c5 = {}
            n5 = max(3, min(a.steps, 5))
            for prec in ("mxfp8", "split_bf16"):
                fx5.precision = prec
                for name, fn in (("ppo_step", c5_ppo), ("reward_pair_step", c5_pair)):
                    for i in range(2):
                        r5 = fn(i)
                    fence()
                    t0 = time.perf_counter()
                    for i in range(n5):
                        r5 = fn(i)
                    fence()
                    dt5 = (time.perf_counter() - t0) / n5
                    if not torch.isfinite(r5).all():
                        raise SystemExit(f"bench: non-finite result in the config-5 {name} ({prec})")
                    c5[f"{name}_{prec}_ms"] = round(dt5 * 1e3, 3)
            fx5.text.embedding.check_ids()
            out["config5_step_ms"] = c5["ppo_step_mxfp8_ms"]
            out["config5_steps_per_sec"] = round(1e3 / c5["ppo_step_mxfp8_ms"], 3)
            out["config5_reward_pair_step_ms"] = c5["reward_pair_step_mxfp8_ms"]
            out["config"]["config5"] = dict(c5, steps=n5, measured=True, workload=(
                "BASELINE configs[4] on one GPU: uint8 frames + token ids -> ViT-L/14 (24 layers, 257 tokens) + visual projection "
                "1024 -> 768 and RoBERTa-base, frozen, in line -> (ppo_step) the PPO step of `value`, 32 items x 2 tags; "
                "(reward_pair_step) one finetune/reward_pair_dataloader.py train step, 64 items x 2 tags (1024 frames).  mxfp8: every "
                "encoder projection on v_mfma_scale_f32_16x16x128_f8f6f4 -- features a few per cent from the split-bf16 ones, score / "
                "pair-order / NDCG@3 drift measured in tests/test_config5_gpu.py; config5_step_ms = ppo_step_mxfp8_ms"))
            del fx5, rmodel, ropt, rsch, raw64
            torch.cuda.empty_cache()
        del fx, raw
    # ================= [D] CPU baseline: the oracle on this box's host cores, bounded sample =================
    if world == 1 and not a.no_cpu_baseline:
        del model, reward, opt, copt, data
        torch.cuda.empty_cache()
        from oracle import cpu_baseline
        r = cpu_baseline.time_ppo_steps(a.batch, a.tags, steps=a.cpu_steps, warmup_bs=2)
        sample = (f"oracle (torch CPU fp32, dropout on in the update): MEDIAN of {r['steps']} measured PPO step(s) at batch {a.batch} x {a.tags} "
                  f"tags after one untimed warm-up step at batch {r['warmup_bs']} (Adam state allocation): rollout {r['rollout_s']:.1f}s + "
                  f"update fwd/bwd {r['fwd_bwd_s']:.1f}s + AdamW(1.045B) {r['adamw_s']:.1f}s = {r['total_s']:.1f}s per head-only step")
        total = r["total_s"]
        if online is not None:
            fb = max(1, a.batch // 4)          # 8 of 32 items (~40 s of host time); the encoders are per-frame / per-sequence
            fe = cpu_baseline.time_feature_extraction(fb, a.tags) * (a.batch / fb)
            total += fe
            sample += (f"; + dual-encoder forward (oracle ViT-B/16 + RoBERTa-base, 12 layers each) timed on {fb} of the {a.batch} items "
                       f"({fb * 16} frames + {fb * a.tags} sequences) and scaled linearly to the batch: {fe:.1f}s -> {total:.1f}s per step")
        out["cpu_baseline"] = {"value": round(1.0 / total, 5), "unit": "PPO steps/s", "cores": r["threads"], "kind": "port",
                               "sample": sample, "head_only_value": round(1.0 / r["total_s"], 5)}
    sys.stdout.write(json.dumps(out) + "\n")          # one write: under a launcher the ranks share the pipe
    sys.stdout.flush()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

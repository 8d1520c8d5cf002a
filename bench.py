#!/usr/bin/env python
"""Headline benchmark: Vid2Seq train-step samples/sec on synthetic (100 frames x 768, 1000 ASR tokens, 256 target
tokens) batches, t5-base, bf16 MFMA compute with fp32 accumulation/master weights, one process per GPU.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = dvc.py's training step with the generative pass (ViT + T5 encoder + decoder + label-smoothed CE forward and
backward, clip_grad_norm_, Adam, time-token renorm), reference-default dropout 0.1; --denoising 1 adds the second
(span-corruption) pass of the reference's default recipe.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import math
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # dense MFMA bf16 peak, MI355X_MICROARCH.md


def flops_per_sample(T=100, L=1000, Lo=256, V=32200, d=768, ff=3072, n_enc=12, n_dec=12, vit_depth=12, vit_mlp=2048):
    """Algorithmic forward FLOPs per sample: GEMMs 2mnk, attention dense (causal not halved)."""
    S = T + L
    vit = vit_depth * (T * (2 * d * 3 * d + 2 * d * d + 4 * d * vit_mlp) + 4 * T * T * d)
    enc = n_enc * (L * (8 * d * d + 4 * d * ff) + 4 * L * L * d)
    dec = n_dec * (Lo * (8 * d * d + 4 * d * ff) + Lo * 4 * d * d + S * 4 * d * d + 4 * Lo * Lo * d + 4 * Lo * S * d)
    head = 2 * Lo * d * V
    return vit + enc + dec + head


_T0 = time.perf_counter()


def log(msg):
    """progress on stderr (stdout carries only the JSON line)"""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def rccl_choices(path: str) -> dict:
    """Summary of what RCCL logged for this rank (NCCL_DEBUG=INFO, subsystems INIT,GRAPH,TUNING): the (collective size -> algorithm,
    protocol) choices, ring / tree channel counts.  Best effort: a missing or differently formatted log yields {"log": ...} only."""
    import re
    info = {"log": path}
    try:
        txt = open(path, errors="replace").read()
    except OSError:
        return info
    algo = {0: "Tree", 1: "Ring", 2: "CollNetDirect", 3: "CollNetChain", 4: "NVLS", 5: "NVLSTree"}
    proto = {0: "LL", 1: "LL128", 2: "Simple"}
    seen = {}
    for m in re.finditer(r"(?:(\w+): )?(\d+) Bytes -> Algo (\d+) proto (\d+)", txt):
        key = f"{m.group(1) or 'coll'} {algo.get(int(m.group(3)), m.group(3))}/{proto.get(int(m.group(4)), m.group(4))}"
        lo, hi, n = seen.get(key, (1 << 62, 0, 0))
        seen[key] = (min(lo, int(m.group(2))), max(hi, int(m.group(2))), n + 1)
    if seen:
        info["algo_proto"] = {k: {"bytes_min": v[0], "bytes_max": v[1], "calls": v[2]} for k, v in sorted(seen.items())}
    m = re.search(r"(\d+) coll channels.*?(\d+) p2p channels", txt)
    if m:
        info["coll_channels"], info["p2p_channels"] = int(m.group(1)), int(m.group(2))
    info["rings_connected"] = "Connected all rings" in txt
    info["trees_connected"] = "Connected all trees" in txt
    return info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60, help="timed steps (default 60 = a 3 s region: box-to-box noise of +-3 %% needs more than a second of signal)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (weak scaling)")
    ap.add_argument("--asr-tokens", type=int, default=1000)
    ap.add_argument("--target-tokens", type=int, default=256)
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--denoising", type=float, default=0.0)
    ap.add_argument("--model", default="t5-base")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-all-cores", action="store_true", help="also time the CPU baseline on os.cpu_count() threads (15 minutes on the 256-thread GPU host)")
    ap.add_argument("--frames", type=int, default=100, help="frames per video = ViT positions (cfg-5 uses 200)")
    ap.add_argument("--packing", action="store_true", help="measure `value` with the padding-free text encoder (the engine's default; exact). "
                    "Off here: `value` computes the pad rows like the reference does, the padding-free rate is reported beside it")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--repeats", type=int, default=3, help="timed regions of --steps steps each: `value` is the first, the others give the spread")
    ap.add_argument("--no-overlap", action="store_true", help="single-stream execution (A/B for the stream overlap)")
    ap.add_argument("--no-generate", action="store_true", help="skip the greedy generate() leg (cfg-4, reported as an extra field)")
    ap.add_argument("--grad-comm-dtype", default="fp32", choices=["fp32", "bf16"], help="wire format of the data-parallel gradient all-reduce")
    ap.add_argument("--bucket-mib", type=int, default=48, help="wire bytes per gradient all-reduce (MiB)")
    ap.add_argument("--shard-optimizer", action="store_true", help="data parallel: reduce-scatter the gradient buckets, Adam on the local 1/N "
                    "stripes, all-gather the bf16 shadow weights under the next forward (train.GradSync shard=True).  OPT-IN: its RCCL path has "
                    "never run on a real multi-GPU node, and it needs Trainer.prepare_checkpoint() before a state_dict()")
    ap.add_argument("--replicated-optimizer", action="store_true", help="data parallel: all-reduce + full Adam on every rank (the default)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = local % max(1, torch.cuda.device_count())      # (several ranks on one GPU only in the gloo control-flow check below)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        # which algorithm / protocol RCCL picks for the step's collectives (ring vs tree, LL / LL128 / Simple) decides what the per-link
        # xGMI bandwidth buys: ask the library to log its choices (TUNING subsystem) and summarise them in the JSON line
        if "NCCL_DEBUG" not in os.environ and os.environ.get("V2S_RCCL_LOG", "0") == "1":      # opt-in: logging changes the timed run
            os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,GRAPH,TUNING", NCCL_DEBUG_FILE=f"/tmp/v2s_rccl_{os.getpid()}_rank{rank}.log")
        backend = os.environ.get("V2S_DIST_BACKEND", "nccl")   # "nccl" == RCCL on ROCm; "gloo" lets two ranks share one GPU to
        if backend == "nccl":                                  # exercise the multi-rank control flow where only one GPU exists
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from vidchapters_amd import SyntheticTokenizer, Vid2Seq, synth
    from vidchapters_amd import lib as L
    from vidchapters_amd.train import Trainer

    B, T, Lx, Lo = a.batch, a.frames, a.asr_tokens, a.target_tokens
    tok = SyntheticTokenizer(32100, 100)
    model = Vid2Seq(a.model, num_features=T, tokenizer=tok, vis_drop=a.dropout, enc_drop=a.dropout, dec_drop=a.dropout, init_seed=1234,
                    device=dev).train()
    log(f"model built: {sum(p.numel() for p in model.parameters()) / 1e6:.1f} M parameters")
    model.engine().overlap = not a.no_overlap
    trainer = Trainer(model, lr=3e-4, clip_max_norm=1.0, generative=1.0, denoising=a.denoising, grad_comm_dtype=a.grad_comm_dtype,
                      bucket_bytes=a.bucket_mib << 20, shard_optimizer=bool(a.shard_optimizer) and not a.replicated_optimizer)
    batch = {k: v.to(dev) for k, v in synth.make_batch(B, T, Lx, Lo, len(tok), 1234 + rank, 768, denoising=a.denoising > 0).items()}
    batch["video"] = batch["video"].to(torch.bfloat16)       # features resident in HBM as bf16 (documented in DESIGN.md)
    batch["input_lens"] = (batch["input_ids"] != 0).sum(1).tolist()      # host-side lengths, as a data loader knows them
    batch["output_lens"] = (batch["output_ids"] != 0).sum(1).tolist()
    if "den_input_ids" in batch:
        batch["den_input_lens"] = (batch["den_input_ids"] != 0).sum(1).tolist()
        batch["den_output_lens"] = (batch["den_output_ids"] != 0).sum(1).tolist()
    model.engine().pack = a.packing

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log("batch ready; warmup")
    # the warm-up steps run back to back like the timed ones (no synchronisation in between): the host enqueues 2-3 steps ahead of the
    # device, so the caching allocator needs the activations of several steps at once -- it must reach THAT steady state during the
    # warm-up, not in the first steps of the timed region (one hipMalloc-bound step of 80+ ms in a 10-step region is 5 % of `value`)
    for i in range(a.warmup):
        losses = trainer.step(batch)
    barrier()
    if a.warmup:
        log(f"{a.warmup} warmup steps done, loss {float(losses['loss'].item()):.4f}")
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    # telemetry of the timed region (rank 0's GPU): average shader clock from two stamps of the hardware counters (two 8-block launches,
    # ~3 us each, inside the region) and socket power / temperature sampled from sysfs by a host thread -- what tells a slow BOX or a
    # power-throttled step from a slow kernel (tools/telemetry.py)
    from tools.telemetry import ClockRegion, Sampler
    clk = ClockRegion(dev)
    with Sampler(local) as tele:
        t0 = time.perf_counter()
        evs[0].record()
        clk.begin()
        for i in range(a.steps):
            losses = trainer.step(batch)
            evs[i + 1].record()
        clk.end()
        barrier()
        dt = time.perf_counter() - t0
    step_ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(a.steps))      # HIP events on the step's main stream
    med_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    comm_ms = trainer.sync.exposed_ms() if world > 1 else 0.0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # run-to-run spread: the same bracketed region twice more (`value` is the FIRST region, the one the contract defines; the repeats
    # only say how far a single region moves on this box -- clocks / thermals move it by a percent or two)
    region_ms = [dt / a.steps * 1e3]
    for _ in range(max(0, a.repeats - 1)):
        barrier()
        t0r = time.perf_counter()
        for i in range(a.steps):
            trainer.step(batch)
        barrier()
        dtr = time.perf_counter() - t0r
        if world > 1:
            t = torch.tensor([dtr], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtr = float(t.item())
        region_ms.append(dtr / a.steps * 1e3)
    loss_val = float(losses["loss"].item())
    bad = 0 if math.isfinite(loss_val) else 1
    if world > 1:          # every rank leaves together (a lone exit would leave the others in the next collective)
        t = torch.tensor([float(bad)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        bad = int(t.item())
    if bad:
        # a step that produced NaN / Inf is not the workload any more: no line rather than a number for a broken run
        raise SystemExit(f"bench.py: the loss is {loss_val} on rank {rank} after the timed region -- the step diverged on some rank, nothing is reported")
    log(f"timed region done: {dt / a.steps * 1e3:.1f} ms/step")
    ms_per_step = dt / a.steps * 1e3
    value = world * B * a.steps / dt

    # algorithmic work per step: 3x forward FLOPs (fwd + dgrad + wgrad)
    cfgm = model.cfg
    fps = flops_per_sample(T, Lx, Lo, cfgm.vocab, cfgm.d_model, cfgm.d_ff, cfgm.n_enc, cfgm.n_dec)
    step_tflop = 3.0 * fps * B / 1e12                     # nominal: the dense-padded algorithmic count of SURVEY 8d
    if not a.packing:
        exec_tflop = step_tflop
    else:                                                 # executed: encoder terms at each sample's valid length
        d_, ff_, ne = cfgm.d_model, cfgm.d_ff, cfgm.n_enc
        enc_nom = ne * (Lx * (8 * d_ * d_ + 4 * d_ * ff_) + 4 * Lx * Lx * d_)
        enc_exec = sum(ne * (n * (8 * d_ * d_ + 4 * d_ * ff_) + 4 * n * n * d_) for n in batch["input_lens"]) / B
        exec_tflop = 3.0 * (fps - enc_nom + enc_exec) * B / 1e12
    out = {
        "metric": f"Vid2Seq train-step samples/sec ({T}f\u00d7768 vis, {Lx} ASR tok)",
        "value": round(value, 3), "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"{'cfg-2' if a.model == 't5-base' else 'cfg-5'}: Vid2Seq {a.model} train step (generative pass"
                               f"{' + denoising pass' if a.denoising > 0 else ''}), per-GPU batch {B}, {T} frames x 768, "
                               f"{Lx} ASR tokens, {Lo} target tokens, dropout {a.dropout}, fp32 master weights + fused clip/Adam/renorm"
                               f"{', encoder rows of pad tokens not computed (exact)' if a.packing else ', pad rows computed like the reference (their gradient is exactly zero: the attention backward kernels detect all-zero dO rows and skip them, bit-identical)'}",
                   "global_batch": world * B, "parallelism": f"dp{world}", "weights": "deterministic synthetic init (no checkpoints offline)"},
        "ms_per_step_hipevent_median": round(med_ms, 3), "samples_per_s_hipevent_median": round(world * B / (med_ms / 1e3), 2),
        "timing_note": f"value = steps / wall time between the two barrier+synchronize brackets (max over ranks): a {dt:.2f} s region of {a.steps} steps "
                       f"(the driver passes --steps itself; this script's own default is 60 = 3 s); the median is over per-step HIP-event intervals on rank 0",
        "ms_per_step_hipevent_min_max": [round(step_ms[0], 3), round(step_ms[-1], 3)],
        "repeats": {"regions": len(region_ms), "ms_per_step": [round(x, 3) for x in region_ms],
                    "samples_per_s_min_median_max": [round(world * B / (x / 1e3), 2) for x in (max(region_ms), sorted(region_ms)[len(region_ms) // 2], min(region_ms))],
                    "note": "the bracketed region of --steps steps repeated back to back in this process; `value` is region 0"},
        "loss": round(loss_val, 5),
        "model_tflops_per_step_per_gpu": round(step_tflop, 2),
        "executed_tflops_per_step_per_gpu": round(exec_tflop, 2),
        "achieved_executed_tflops_per_gpu": round(exec_tflop / (ms_per_step / 1e3), 1),
        "frac_of_mfma_peak_whole_step": round(exec_tflop / (ms_per_step / 1e3) / PEAK_BF16_TFLOPS, 4),
    }
    out["clock_power"] = dict(effective_sclk_mhz=clk.mhz(), sclk_max_mhz=2400, **tele.summary(),
                              note="timed region 0 on rank 0's GPU.  effective_sclk = shader cycles clocked / wall time (s_memtime vs the 100 MHz "
                                   "s_memrealtime, per XCD, averaged): the step is power-bound, so a box that sustains a lower clock at the same "
                                   "socket power is slower by that ratio whatever the kernels do (DESIGN.md 8a-r6)")

    if world > 1:
        out["data_parallel"] = {"ranks_seen_by_rccl": dist.get_world_size(), "backend": dist.get_backend(),
                                "grad_comm_dtype": trainer.sync.comm_dtype, "bucket_mib": round(trainer.sync.chunk * (4 if trainer.sync.comm_dtype == "fp32" else 2) / 2 ** 20, 1),
                                "collectives_per_step": trainer.sync.collectives, "wire_mb_per_step": round(trainer.sync.bytes_reduced / 1e6, 1),
                                "optimizer": ("sharded: reduce-scatter + Adam on 1/N stripes + bf16 all-gather" if trainer.sync.shard else "replicated: all-reduce + full Adam on every rank"),
                                "exposed_comm_ms_last_step_rank0": round(comm_ms, 3),
                                "shadow_allgather": ("on the communication stream under the next step's forward (per-group events)" if trainer.sync.shard and trainer.overlap_gather
                                                     else ("on the main stream after Adam" if trainer.sync.shard else "none")),
                                "rccl": rccl_choices(os.environ.get("NCCL_DEBUG_FILE", "")),
                                "note": "exposed = time the main stream waited for the gradient reduction after backward had been enqueued"}
        # the prediction of DESIGN.md section 6, so that the first real multi-GPU line judges itself: wire bytes per rank of a ring all-reduce over
        # xGMI links of ~153 GB/s -- all seven links (direct reduce-scatter + all-gather) or one ring --, all but the last bucket hidden under backward
        wire = trainer.sync.bytes_reduced * 2.0 * (world - 1) / world
        single_gpu_ms = 50.5 if a.model == "t5-base" else None       # this step on one GPU (round-6 boxes 49.1-51.5 ms): weak scaling holds it
        pred = {"wire_gb_per_rank": round(wire / 1e9, 3), "allreduce_ms_all_links": round(wire / (7 * 153e9) * 1e3, 2), "allreduce_ms_one_ring": round(wire / 153e9 * 1e3, 2),
                "exposed_comm_ms_expected": "0.2 - 1.1 (the last bucket: tied embedding + small parameters, ~100 MB); above ~2 ms the buckets are not overlapping",
                "weak_scaling_efficiency_expected": ">= 0.97"}
        if single_gpu_ms:
            pred["ms_per_step_expected"] = f"{single_gpu_ms:.0f} - {single_gpu_ms / 0.97:.1f}"
            pred["verdict"] = ("as predicted" if (comm_ms <= 2.0 and ms_per_step <= single_gpu_ms / 0.97 * 1.03) else
                               "exposed communication above the prediction: buckets not overlapping with backward" if comm_ms > 2.0 else
                               "step slower than predicted with little exposed communication: look at clock_power and the per-rank medians")
        out["data_parallel"]["prediction"] = pred
        out["data_parallel"]["measured_vs_predicted"] = {"exposed_comm_ms": round(comm_ms, 3), "ms_per_step": round(ms_per_step, 3)}
    if rank == 0 and world == 1 and not a.packing and not a.no_generate:
        with _Leg(out, "padding_free", model):
            # the same step with the engine's default padding-free text encoder (pad-token rows are not computed; exact, see
            # DESIGN.md): reported beside `value`, which computes them like the reference does
            eng = model.engine()
            eng.pack = True
            trainer.step(batch)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                trainer.step(batch)
            torch.cuda.synchronize()
            dtp = (time.perf_counter() - t0) / 3
            eng.pack = False
            out["padding_free"] = {"valid_encoder_tokens": int(sum(batch["input_lens"])), "padded_encoder_tokens": int(B * Lx),
                                   "valid_decoder_rows": int(sum(batch["output_lens"])), "padded_decoder_rows": int(B * Lo),
                                   "ms_per_step": round(dtp * 1e3, 3), "samples_per_s": round(B / dtp, 2),
                                   "note": "engine default (Engine.pack, Engine.pack_dec): the text encoder runs on the non-pad tokens only and the "
                                           "decoder on the rows of real targets only; exact because the reference masks those rows as keys everywhere "
                                           "and ignores their labels.  `value` above does NOT use it"}
    if rank == 0 and world == 1 and not a.no_generate:
        with _Leg(out, "captured_step", model):
            # the same step replayed from ONE hipGraph (Trainer.step_graph: batch / dropout salt / Adam scalars read from device memory):
            # host time per step and device time per step; `value` above is the eager path, which is faster on the device on this stack
            gb = {k: v for k, v in batch.items() if torch.is_tensor(v)}
            for _ in range(3):
                trainer.step_graph(gb)
            torch.cuda.synchronize()
            enq, wall = [], []
            for _ in range(5):
                t0 = time.perf_counter(); trainer.step_graph(gb); t1 = time.perf_counter()
                torch.cuda.synchronize(); t2 = time.perf_counter()
                enq.append((t1 - t0) * 1e3); wall.append((t2 - t0) * 1e3)
            eag = []
            for _ in range(4):               # the first eager step after the replays re-creates the eager workspaces: not counted
                t0 = time.perf_counter(); trainer.step(batch); t1 = time.perf_counter(); torch.cuda.synchronize()
                eag.append((t1 - t0) * 1e3)
            eag = sorted(eag[1:])
            out["captured_step"] = {"host_enqueue_ms_per_step": round(sorted(enq)[2], 3), "ms_per_step": round(sorted(wall)[2], 3),
                                    "eager_host_call_ms_per_step": round(eag[1], 3),
                                    "note": "Trainer.step_graph: one hipGraph launch per step (~2500 kernel nodes on three streams).  eager_host_call = time inside the eager Trainer.step() on the host, which includes waiting for room in the launch queue once the host is ~1000 launches ahead of the GPU (tools/cpu_enqueue.py measures 15-23 ms of pure enqueue work)"}
    if rank == 0 and world == 1 and not a.no_roofline:      # N=1 only: the extra step would issue collectives other ranks do not join
        with _Leg(out, "roofline", model):
            eng = model.engine()
            was = eng.overlap
            eng.overlap = False                 # per-launch durations are only meaningful without concurrent kernels
            with L.KernelTimer(by_symbol=True) as kt:
                trainer.step(batch)
            eng.overlap = was
            summ = kt.summary()
            log("roofline leg done")
            # dominant kernel = the kernel SYMBOL with the largest summed duration (GEMM launches are tagged with the variant the
            # library dispatched, so the name and the per-launch average line up with rocprofv3 --kernel-trace --stats)
            tag = max(summ, key=lambda k: summ[k][1])
            n, ms, work = summ[tag]
            ach = work / (ms / 1e3) / 1e12
            out["roofline"] = {"kernel": tag, "bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": None,
                               "effective_sclk_mhz": out["clock_power"].get("effective_sclk_mhz"), "power_w_avg": out["clock_power"].get("power_w_avg"),
                               "launches_per_step": n, "avg_launch_us": round(ms / n * 1e3, 2),
                               "algorithmic_gflop_per_launch": round(work / n / 1e9, 2),
                               "note": "achieved = sum over the step's launches of this kernel of 2*M*N*K (attention: 4*B*H*Nq*Nk*64 fwd, "
                                       "8*... bwd) / their summed HIP-event durations, measured with stream overlap disabled"}
            # the round-5 kernel family (persistent asm-scheduled GEMM) next to the dominant symbol: its launches, time and rate in this same step
            fam = [(k, v) for k, v in summ.items() if k.startswith("gemm_a4p_kernel")]
            if fam:
                fn_, fms, fwork = sum(v[0] for _, v in fam), sum(v[1] for _, v in fam), sum(v[2] for _, v in fam)
                fach = fwork / (fms / 1e3) / 1e12
                out["roofline"]["gemm_a4p_family"] = {"symbols": sorted(k for k, _ in fam), "launches_per_step": fn_, "ms_per_step": round(fms, 3),
                                                      "achieved": round(fach, 1), "unit": "TFLOP/s", "frac": round(fach / PEAK_BF16_TFLOPS, 4),
                                                      "frac_of_sustained_mfma": round(fach / 1650.0, 4),
                                                      "note": "all instantiations of gemm_a4p_kernel in this step; sustained = 1650 TF/s, pure v_mfma on random bf16 operands "
                                                              "on this chip (profiles/r05_mfma_power_ubench.txt)"}
            # HBM traffic of that kernel: from the committed PMC passes (tools/pmc_traffic.sh -> profiles/*.json; rocprofv3 --pmc cannot
            # wrap this whole script on this stack -- it crashes in torch's integer kernels), launch-weighted over the step's shapes
            try:
                for fn in sorted(os.listdir(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles"))):
                    if fn.endswith(".json") and "pmc_traffic" in fn:
                        pj = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", fn)))
                        if pj.get("kernel") == tag:
                            out["roofline"]["traffic"] = round(pj["traffic"])
                            out["roofline"]["traffic_unit"] = "bytes per launch (PMC, offline: profiles/" + fn + ")"
                            out["roofline"]["algorithmic_bytes_per_launch"] = round(pj["algorithmic"])
                            # the PMC passes ran on the launch mix of the day they were taken: say so if this run's mix differs
                            fl = sum(2.0 * sh["M"] * sh["N"] * sh["K"] * sh["launches_per_step"] for sh in pj.get("shapes", [])) / max(1, pj.get("launches_per_step", 1))
                            out["roofline"]["traffic_matches_this_launch_mix"] = bool(pj.get("launches_per_step") == n and abs(fl - work / n) < 0.02 * fl)
            except OSError:
                pass
            out["kernel_breakdown_note"] = ("HIP-event pairs around each launch on the launch stream; at ~2500 launches per step this leg is host-bound, so "
                                            "short kernels carry launch gaps (see profiles/ for the rocprofv3 durations)")
            out["kernel_breakdown_ms_per_step"] = {k: {"launches": v[0], "ms": round(v[1], 3), "tflops": round(v[2] / (v[1] / 1e3) / 1e12, 1)}
                                                   for k, v in sorted(summ.items(), key=lambda kv: -kv[1][1])}

    if rank == 0 and world == 1 and not a.no_generate and a.denoising == 0:
        with _Leg(out, "two_pass_and_generate", model):
            # S2 of SURVEY.md 8d, reported alongside: the dvc.py default two-pass step (generative + denoising pass on the cached
            # video_dict, L~800 / Lo~301 span-corruption shapes from synth.make_batch), same optimizer recipe
            tr2 = Trainer(model, lr=3e-4, clip_max_norm=1.0, generative=1.0, denoising=1.0)
            b2 = {k: v.to(dev) for k, v in synth.make_batch(B, T, Lx, Lo, len(tok), 1234, 768, denoising=True).items()}
            b2["video"] = b2["video"].to(torch.bfloat16)
            tr2.step(b2)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                l2 = tr2.step(b2)
            torch.cuda.synchronize()
            dt2 = (time.perf_counter() - t0) / 3
            out["two_pass_step"] = {"ms_per_step": round(dt2 * 1e3, 3), "samples_per_s": round(B / dt2, 2),
                                    "den_input_tokens": int(b2["den_input_ids"].shape[1]), "den_target_tokens": int(b2["den_output_ids"].shape[1]),
                                    "loss": round(float(l2["loss"].item()), 5)}
            del tr2, b2
            out["generate_greedy"] = generate_leg(model, tok, dev, Lx)
            out["generate_beam4"] = beam_leg(model, tok, dev, Lx)
            out["input_pipeline"] = input_leg(dev, B, Lx, Lo)

    if rank == 0 and "roofline" in out and "generate_greedy" in out:
        # the decode legs' summary inside `roofline` (the driver keeps that object whole): cached decode is HBM-bound, fractions of 8 TB/s
        def dsum(leg):
            r = leg["roofline"]
            return {"ms_per_step": leg["ms_per_decode_step"], "sequences_per_s": leg["sequences_per_s"], "bound": "hbm", "achieved": r["achieved"],
                    "peak": r["peak"], "unit": r["unit"], "frac": r["frac"], "algorithmic_gb_per_step": r["algorithmic_gb_per_step"],
                    "launches": leg.get("launches_per_decode_step")}
        out["roofline"]["decode"] = {"greedy": dsum(out["generate_greedy"]), "beam4": dsum(out["generate_beam4"]),
                                     "note": "cfg-4: greedy B = 64 x 256 steps; the callers' default beam search: 16 entries x 4 beams x 64 steps; "
                                             "launches = library launches of one captured decode step (vidchapters_amd.lib.launch_count)"}

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        with _Leg(out, "cpu_baseline", model):
            log("cpu baseline (oracle on host cores) ...")
            out["cpu_baseline"] = cpu_baseline(model, tok, Lx, Lo, all_cores=a.cpu_all_cores)
            log("cpu baseline done")

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def generate_leg(model, tok, dev, Lx, B=64, new_tokens=256):
    """cfg-4: greedy generate() at B=64, 100 frames + Lx ASR tokens, up to 256 new tokens (extra field, not `value`)."""
    from vidchapters_amd import synth
    model.eval()
    b = synth.make_batch(B, 100, Lx, 8, len(tok), 4321, 768)
    video = b["video"].to(dev).to(torch.bfloat16)
    ids = b["input_ids"].to(dev)
    inp = {"input_ids": ids, "attention_mask": ids != 0}
    eng = model.engine()
    import gc
    gc.collect(); torch.cuda.empty_cache(); torch.cuda.synchronize()      # the training legs' garbage and cached blocks: released here, not inside a timed call
    # The default cross-attention path and, for comparison, the per-layer cross K/V caches (Engine.decode_mem_attn = 0), interleaved
    # twice in this process, best of each: the first graph capture after the training legs releases the allocator's cache
    # (torch.cuda.graph: synchronize + gc + empty_cache -- hundreds of ms, paid by whichever call comes first), so one run each
    # measures the order of the calls, not the paths.
    was = eng.decode_mem_attn
    best, runs = {}, []
    toks = None
    for rep in range(2):
        for mode in (was, 0):
            eng.decode_mem_attn = mode
            eng.greedy(video, inp, max_new_tokens=8)            # warm-up (allocations, LUTs)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out_toks = eng.greedy(video, inp, max_new_tokens=new_tokens, stop_at_eos=False)      # fixed 256 decode steps
            torch.cuda.synchronize()
            runs.append((int(mode), round(time.perf_counter() - t0, 4)))
            best[mode] = min(best.get(mode, 1e9), runs[-1][1])
            if mode == was:
                toks = out_toks
                n_launch = int(getattr(eng, "last_decode_launches", 0))
                path_was = getattr(eng, "last_cross_path", "")
    eng.decode_mem_attn = was
    dt, dt_kv = best[was], best[0]
    steps = toks.shape[1] - 1
    model.train()
    ms_step = dt / max(steps, 1) * 1e3
    return {"batch": B, "decode_steps": int(steps), "seconds": round(dt, 4), "sequences_per_s": round(B / dt, 2),
            "cross_attention": "shared encoder memory (Engine.decode_mem_attn = 1: taken from 40 000 memory positions B x S on)" if path_was == "memory" else "per-layer K/V caches",
            "runs_mode_seconds": runs,
            "kv_cache_path": {"seconds": round(dt_kv, 4), "sequences_per_s": round(B / dt_kv, 2), "ms_per_decode_step": round(dt_kv / max(steps, 1) * 1e3, 3)},
            "launches_per_decode_step": n_launch,
            "ms_per_decode_step": round(ms_step, 3), "roofline": decode_roofline(model, B, 100 + Lx, new_tokens, ms_step, rows=B, valid_keys=100 * B + int((ids != 0).sum()),
                                        on_memory=(path_was == 'memory')),
            "note": "encode + 256 greedy decode steps (EOS stop disabled so that every run decodes the full length), static KV cache"}


def decode_roofline(model, B, S, maxlen, ms_step, rows, valid_keys=None, on_memory=False):
    """HBM roofline of one cached decode step (the step is bandwidth-bound: every byte below is read once per step and nothing is reused
    across steps): the cross-attention stream of every layer over every VALID encoder position of every batch entry (`valid_keys` =
    their count over the batch; padded positions are never fetched; shared by the beams of an entry) -- per-layer K and V ([., 2 * inner])
    on the K/V-cache path, the d_model-wide memory row itself plus the folded-query / partial-sum buffers on the memory path
    (`on_memory`, csrc/v2s_memattn.hip: half the bytes) --, the self-attention cache (average fill maxlen/2), the bf16 decoder weights
    incl. the tied LM head (+ the transposed K projection the memory path reads), and the fp32 logits written and read back.
    `achieved` includes the encoder pass amortised over the steps (ms_per_decode_step is end-to-end / steps)."""
    c = model.cfg
    d, inner, ff, nl, V = c.d_model, c.inner, c.d_ff, c.n_dec, c.vocab
    keys = valid_keys if valid_keys is not None else B * S
    if on_memory:
        pieces = 4 * B                                                         # <= 4 pieces of <= 288 keys per entry up to 1152 keys
        cross = nl * (keys * d * 2 + 2 * rows * (inner // 64) * d * 2 + 2 * pieces * 16 * d * 2)      # memory rows; folded queries and partials written + read
    else:
        cross = nl * keys * 2 * inner * 2
    selfc = nl * rows * (maxlen / 2) * 2 * inner * 2
    weights = nl * (3 * inner * d + inner * d + inner * d + inner * d + 2 * d * ff) * 2 + V * d * 2
    logits = rows * V * 4 * 2
    total = cross + selfc + weights + logits
    ach = total / (ms_step / 1e3) / 1e9
    pmc = None                 # measured HBM bytes of the cross-attention kernel per layer (PMC, offline: B = 64 greedy shapes only)
    try:
        if rows == 64:
            pj = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r04_pmc_traffic_decode_cross_attention.json")))
            pmc = int(pj["kernels"]["mem_attn_kernel" if on_memory else "decode_attn_kernel"]["traffic_bytes"])
    except (OSError, KeyError, ValueError):
        pmc = None
    return {"bound": "hbm", "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4),
            "algorithmic_gb_per_step": round(total / 1e9, 3), "traffic": None,
            "cross_attention_traffic_bytes_per_layer": pmc,
            "cross_attention_traffic_note": "PMC, offline (profiles/r04_pmc_traffic_decode_cross_attention.json; 57 810 valid keys there)" if pmc else None,
            "cross_attention_path": "encoder memory" if on_memory else "per-layer K/V caches",
            "bytes": {"cross": int(cross), "self_kv": int(selfc), "weights": int(weights), "logits": int(logits)}}


def beam_leg(model, tok, dev, Lx, B=16, new_tokens=64, num_beams=4):
    """The reference's default decoding (num_beams=4, vid2seq.py:100-111) at B=16 (64 live beams), 64 new tokens."""
    from vidchapters_amd import synth
    model.eval()
    b = synth.make_batch(B, 100, Lx, 8, len(tok), 4321, 768)
    video = b["video"].to(dev).to(torch.bfloat16)
    ids = b["input_ids"].to(dev)
    inp = {"input_ids": ids, "attention_mask": ids != 0}
    eng = model.engine()
    eng.beam_search(video, inp, num_beams=num_beams, max_new_tokens=4)
    dt = 1e9
    for _ in range(2):                  # best of two (graph-capture housekeeping after other legs lands on whichever call comes first)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        toks = eng.beam_search(video, inp, num_beams=num_beams, max_new_tokens=new_tokens, min_length=new_tokens + 1)   # EOS banned: full length
        torch.cuda.synchronize()
        dt = min(dt, time.perf_counter() - t0)
    launches = int(getattr(eng, "last_decode_launches", 0))
    dt_host = 1e9                       # the same search with the hypothesis bookkeeping on the host (one round trip per step), for comparison
    was = eng.beam_on_device
    eng.beam_on_device = False
    try:
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            toks_host = eng.beam_search(video, inp, num_beams=num_beams, max_new_tokens=new_tokens, min_length=new_tokens + 1)
            torch.cuda.synchronize()
            dt_host = min(dt_host, time.perf_counter() - t0)
    finally:
        eng.beam_on_device = was
    model.train()
    return {"batch": B, "num_beams": num_beams, "scorer": "device (v2s_beam_advance inside the replayed graph)" if was else "host",
            "host_scorer": {"seconds": round(dt_host, 4), "sequences_per_s": round(B / dt_host, 2),
                            "tokens_identical": bool(toks.shape == toks_host.shape and (toks == toks_host).all())}, "max_new_tokens": new_tokens, "returned_length": int(toks.shape[1]), "seconds": round(dt, 4),
            "sequences_per_s": round(B / dt, 2), "ms_per_decode_step": round(dt / new_tokens * 1e3, 3), "launches_per_decode_step": launches,
            "roofline": decode_roofline(model, B, 100 + Lx, new_tokens, dt / new_tokens * 1e3, rows=B * num_beams,
                                        valid_keys=100 * B + int((ids != 0).sum())),
            "note": "encode + beam search, min_length = max length so that every run decodes all steps"}


class _Leg:
    """An optional leg of the line (everything after `value` has been measured): a Python-level failure inside it is recorded under
    `leg_errors` and the line is still printed -- the headline must not depend on, say, a hipGraph capture working on this box."""

    def __init__(self, out, name, model):
        self.out, self.name, self.eng = out, name, model.engine()
        self.pack, self.overlap = self.eng.pack, self.eng.overlap

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        if et is None or not issubclass(et, Exception):
            return False
        self.eng.pack, self.eng.overlap = self.pack, self.overlap
        self.out.setdefault("leg_errors", {})[self.name] = f"{et.__name__}: {ev}"[:500]
        log(f"leg {self.name} FAILED: {et.__name__}: {ev}")
        try:
            torch.cuda.synchronize()
        except Exception:
            pass
        return True


def input_leg(dev, B, Lx, Lo, frames=300):
    """Host -> device batch preparation (vidchapters_amd.data.DeviceBatcher: frame subsampling, padding, pinned H2D, bf16 cast and
    span corruption on the GPU) for one cfg-2 batch from host-resident fp32 features; runs on a side stream in training, so the
    PCIe-inclusive step rate is min(step rate, this rate).  Extra field, never `value`."""
    import numpy as np
    from vidchapters_amd.data import DeviceBatcher
    rng = np.random.RandomState(0)
    samples = []
    for _ in range(B):
        ins = rng.randint(2, 32100, size=Lx).astype(np.int64); ins[-1] = 1
        outs = rng.randint(2, 32200, size=Lo).astype(np.int64); outs[-1] = 1
        samples.append({"video": rng.randn(frames, 768).astype(np.float32), "input_tokens": ins, "output_tokens": outs})
    batcher = DeviceBatcher(dev, max_feats=100, num_text_tokens=32100)
    batcher(samples); batcher.ready.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        batcher(samples); batcher.ready.synchronize()
    dt = (time.perf_counter() - t0) / 5
    return {"ms_per_batch": round(dt * 1e3, 3), "samples_per_s": round(B / dt, 1), "h2d_bytes_per_batch": B * (100 * 768 * 4 + (Lx + Lo) * 8 + Lx + 4),
            "note": f"{frames} source frames/sample -> 100, single host thread incl. numpy span-mask RNG"}


def _cpu_model_string():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(model, tok, Lx, Lo, threads=32, all_cores=False):
    """The CPU oracle (fp32 torch restatement of the reference path, pinned against the reference in the build container) timed on this
    box's host cores, SURVEY 8d protocol: per leg 1 warm-up + 3 timed iterations, median.  Legs: cfg-1 exact (B=2, 100 frames, 256 ASR
    tokens, 256 targets: forward + loss + backward + clip + Adam + renorm), the cfg-2 shapes at B=2 (1000 ASR tokens) -- the number
    `value` is compared with -- and 32 greedy decode steps at B=2.  Thread count: SURVEY 8d says os.cpu_count(), but on the 2 x 64-core
    (256 logical) GPU host torch's CPU kernels collapse when oversubscribed at these matrix sizes: measured B=2 step 3.9 s at 32
    threads, 13.7 s at 128 (round 1) and 891 s at os.cpu_count() = 256 (round 3, profiles/r03_bench_default_with_cpu_all_cores_probe.json
    -- that one probe took 15 minutes).  So the baseline runs at 32 threads (`cores`), the fastest setting found; ``--cpu-all-cores``
    repeats the probe."""
    from oracle import vid2seq_ref as R
    from vidchapters_amd import synth
    ncores = os.cpu_count() or 1
    threads = min(threads, ncores)
    torch.set_num_threads(threads)
    cfg = R.RefConfig(vocab=len(tok))
    P = {k: v.detach().float().cpu().clone().requires_grad_(True) for k, v in model.named_parameters()}

    def timed(fn, n=3, budget_s=None):
        t0 = time.perf_counter(); fn(); warm = time.perf_counter() - t0
        if budget_s is not None and warm > budget_s:          # too slow to repeat within the bench's time budget: the warm-up is the sample
            return warm, 1
        ts = []
        for _ in range(n):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2], n

    legs = {}
    for name, L_in in (("cfg1_exact_B2_L256", 256), ("cfg2_shapes_B2", Lx)):
        b = synth.make_batch(2, 100, L_in, Lo, len(tok), 99, 768)
        state = {}
        dt, _ = timed(lambda: R.train_step(P, state, cfg, b, lr=3e-4, clip=1.0, generative=1.0, denoising=0.0))
        legs[name] = {"samples_per_s": round(2 / dt, 4), "seconds_per_step": round(dt, 3), "threads": threads}
    main, used = legs["cfg2_shapes_B2"], threads
    if all_cores and ncores > threads:                        # the same leg on every logical core of the host (can take many minutes)
        torch.set_num_threads(ncores)
        b = synth.make_batch(2, 100, Lx, Lo, len(tok), 99, 768)
        state = {}
        dt, n = timed(lambda: R.train_step(P, state, cfg, b, lr=3e-4, clip=1.0, generative=1.0, denoising=0.0), n=2, budget_s=12.0)
        legs["cfg2_shapes_B2_all_cores"] = {"samples_per_s": round(2 / dt, 4), "seconds_per_step": round(dt, 3), "threads": ncores, "timed_steps": n}
        if dt < main["seconds_per_step"]:
            main, used = legs["cfg2_shapes_B2_all_cores"], ncores
        torch.set_num_threads(threads)
    b = synth.make_batch(2, 100, Lx, 8, len(tok), 98, 768)
    Pd = {k: v.detach() for k, v in P.items()}
    dtg, _ = timed(lambda: R.greedy_generate(Pd, cfg, b["video"], b["input_ids"], b["input_ids"] != 0, 32), n=3)
    legs["greedy_32_steps_B2"] = {"sequences_per_s": round(2 / dtg, 4), "seconds": round(dtg, 3), "threads": threads}
    return {"value": main["samples_per_s"], "unit": "samples/s", "cores": used, "kind": "port",
            "host": {"cpu_model": _cpu_model_string(), "logical_cores": ncores, "threads_used": used,
                     "thread_sweep_seconds_per_step": {"32": 3.9, "128": 13.7, "256": 891.4},
                     "thread_sweep_note": "B=2 cfg-2 step measured on this host type in rounds 1 and 3 (see the docstring); 32 threads is the fastest setting found"},
            "sample": f"median of 3 timed optimizer steps after 1 warm-up (fwd+bwd+clip+Adam+renorm) at B=2, 100 frames, {Lx} ASR tokens, {Lo} target "
                      f"tokens, fp32 torch CPU oracle, dropout 0, {used} threads; {main['seconds_per_step']} s per step",
            "legs": legs}


if __name__ == "__main__":
    main()

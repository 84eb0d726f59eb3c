#!/usr/bin/env python3
"""Headline benchmark: 10 s audio clips / second for single-step consistency generation
(text states -> latent -> mel -> 16 kHz waveform; FLAN-T5 excluded) on N MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1] (SURVEY.md §8d "Config 2"): batch 32 per GPU, L = 32 text
tokens with per-row valid lengths ~U{6..32}, guidance w = 4, one U-Net query, AudioLDM VAE
decode, HiFi-GAN vocoder, int16 conversion.  A "step" is one such batch.  Clips are
independent, so N GPUs run N replicas of the batch (weak scaling, no data-path collective);
the only collectives are the timing barrier / max-reduce over RCCL.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     -- conv_gemm (the implicit-GEMM MFMA kernel family = >95 % of the step's FLOPs):
                  algorithmic FLOPs per step / summed launch time per step, measured live with
                  HIP events on the launch stream (ctta_prof_*), against the dense bf16 MFMA peak.
  cpu_baseline -- the CPU oracle (fp32 PyTorch-CPU restatement of the reference) timed on this
                  box's host cores for the same pipeline at B=1 (reference recipe, config 1).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Algorithmic FLOPs per clip (2*MAC), SURVEY.md §2a / §8d "Algorithmic work per unit"
GF_UNET_CONV, GF_UNET_LINEAR_L16, GF_UNET_LINEAR_PER_16TOK = 267.9, 163.2, 1.3
GF_UNET_SELF_ATTN, GF_UNET_CROSS_ATTN_L16 = 97.6, 0.59
GF_VAE_CONV, GF_VAE_ATTN = 636.1, 34.4
GF_HIFIGAN = 1027.0
GF_SMALL_N = 0.15 + 0.15 + 0.07          # conv_out (U-Net, VAE) and conv_post run on the direct kernel
PEAK_BF16_TFLOPS = 2500.0                 # dense MFMA peak, MI355X_MICROARCH.md
GB_CLIP_FUSED = 3.15                       # operand bytes per clip, every conv / linear / attention operand once
PEAK_HBM_GBPS = 8000.0                    # HBM3E peak, MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--text-len", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", choices=("both", "gen", "distill", "teacher", "perceptual"), default="both",
                    help="both (default): configs[1] generation line + `distill` (configs[3]) and `teacher` (configs[2]) "
                         "objects; gen / distill / teacher: only that leg (profiling aids)")
    ap.add_argument("--teacher-steps", type=int, default=200, help="Heun steps of the teacher leg (2N-1 U-Net queries)")
    ap.add_argument("--teacher-batch", type=int, default=8)
    ap.add_argument("--distill-batch", type=int, default=9, help="per-GPU micro-batch of the distillation leg (train.sh)")
    ap.add_argument("--no-latency", action="store_true",
                    help="for the rocprofv3 --pmc passes: no hipGraph anywhere (the headline loop times eager launches, the "
                         "single-clip eager/hipGraph latency leg is skipped -- a graph replay under --pmc hung on this pool in "
                         "round 1) and, in --mode distill, no AdamW / EMA timing passes and no wav -> latent leg")
    ap.add_argument("--perceptual-fused-batch", type=int, default=45,
                    help="micro-batch of the configs[4] `fused_micro_batch` leg (train.sh accumulates 15 x 3 samples per GPU)")
    ap.add_argument("--perceptual-batch", type=int, default=4,
                    help="per-GPU micro-batch of the perceptual-loss leg (configs[4] without CLAP)")
    ap.add_argument("--profile-csv", default=None, help="append one line per MFMA launch (tuning aid)")
    return ap.parse_args()


def launcher_command(argv, n, port, script=None):
    """The command line of the child `torch.distributed.run` that `python bench.py --gpus N` (N > 1, no WORLD_SIZE in the
    environment) starts: one rank per GPU on this node, rendezvous on 127.0.0.1 (the container hostname may not resolve),
    the script's own arguments passed through unchanged."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n)),
            "--master-addr", "127.0.0.1", "--master-port", str(int(port)), script or os.path.abspath(__file__)] + list(argv)


def self_launch(args, argv, script=None):
    """`python bench.py --gpus N` by itself: the parent -- which has NOT imported torch, let alone touched the GPU -- starts
    the N ranks as a CHILD process (never exec: a process that initialised HIP must not be replaced, and this one must stay
    to relay), relays the child's stdout so that the one JSON line is the last line the parent prints, and returns its exit
    code.  Under an external torchrun (WORLD_SIZE set) this function is never reached.
    The rendezvous port is found by bind(0) and handed to the child after the socket is closed: another process may take it
    in between (two launches on one node).  A child that fails within a minute WITHOUT having printed a line is therefore
    started again on a fresh port, twice at most, with a note on stderr; any other failure is the parent's exit code."""
    import socket
    import subprocess
    rc = 1
    for attempt in range(3):
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = s_.getsockname()[1]
        cmd = launcher_command(argv, args.gpus, port, script)
        if os.environ.get("CTTA_BENCH_LAUNCH_DRYRUN") == "1":
            print(json.dumps({"launcher_command": cmd}), flush=True)
            return 0
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
        env.setdefault("OMP_NUM_THREADS", "4")
        t0 = time.time()
        child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, bufsize=1, env=env)
        line_json, printed = None, 0
        for line in child.stdout:
            printed += 1
            if line.startswith('{"metric"'):
                line_json = line        # held back: printed last, whatever a rank's C stdio still flushes at exit
            else:
                sys.stdout.write(line)
                sys.stdout.flush()
        rc = child.wait()
        if line_json is not None:
            sys.stdout.write(line_json)
            sys.stdout.flush()
        if rc == 0 or printed > 0 or time.time() - t0 > 60.0 or attempt == 2:
            return rc
        sys.stderr.write("[bench] the rank launcher exited with code %d before any rank printed (rendezvous port %d taken by "
                         "another process?); starting it again on a fresh port\n" % (rc, port))
        sys.stderr.flush()
    return rc


def trace(msg):
    """CTTA_BENCH_TRACE=1: leg boundaries with wall-clock stamps on stderr (every rank) -- tells a slow multi-rank run through a
    host-side backend from a hung one."""
    if os.environ.get("CTTA_BENCH_TRACE", "0") != "0":
        sys.stderr.write("[bench %s rank %s] %s\n" % (time.strftime("%H:%M:%S"), os.environ.get("RANK", "0"), msg))
        sys.stderr.flush()


def flush_c_stdio():
    """RCCL prints its banner (ROCm version / hostname / library path) with C stdio; on a pipe that buffer is only
    flushed at exit, i.e. AFTER the JSON line.  Flushing it here keeps the JSON line the last line on stdout."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()


def emit(result, rank, dev):
    """Every rank drains its stdio, then rank 0 prints the ONE JSON line."""
    from consistencytta_amd import dist_util as du
    flush_c_stdio()
    du.barrier(dev)
    if rank == 0:
        print(json.dumps(result), flush=True)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))     # before `import torch`: the parent never initialises HIP
    if args.mode in ("distill", "perceptual"):
        args.no_cpu_baseline = True
    import torch

    from consistencytta_amd import _native as N
    from consistencytta_amd import dist_util as du
    from consistencytta_amd import modules, spec
    from consistencytta_amd.models import ConsistencyTTA

    world, rank, local_rank = du.env_world()
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d (or unset WORLD_SIZE and let "
                         "bench.py start the ranks itself)" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # CTTA_BENCH_BACKEND=gloo rehearses the multi-rank flow (same collective sequence on every rank) on a box with
    # fewer GPUs than ranks: ranks share devices round-robin and the collectives go through the host
    backend = os.environ.get("CTTA_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    du.init(backend, dev)   # "nccl" is RCCL on ROCm: timing barrier / max-reduce, gradient all-reduce of the distill leg
    du.barrier(dev)         # first collective: the communicator (and its banner) exists from here on
    rccl_ranks = du.count_ranks(dev)      # an all-reduce(SUM) of ones: how many ranks the communicator really joined
    assert rccl_ranks == world, "the process group sums %d ranks, WORLD_SIZE says %d" % (rccl_ranks, world)
    args.rccl_ranks, args.backend = rccl_ranks, backend
    flush_c_stdio()
    if args.mode == "teacher":
        d = teacher_leg(args, dev, world, rank)
        emit(d, rank, dev)
        du.finish()
        return
    if args.mode in ("distill", "perceptual"):   # profiling aid: only that leg, printed as the JSON line
        d = distill_leg(args, dev, world, rank, perceptual=args.mode == "perceptual")
        if rank == 0:
            d.update({"higher_is_better": True, "vs_baseline": None,
                      "data": "synthetic"})
        emit(d, rank, dev)
        du.finish()
        return

    B, L = args.batch, args.text_len
    # ---- models: light U-Net + AudioLDM-s VAE/vocoder architecture, random init (no checkpoints offline)
    vae = modules.AutoencoderKL(ddconfig=spec.VAE_DDCONFIG, embed_dim=8, scale_factor=0.9227914214134216)
    pipe = ConsistencyTTA(unet_config=spec.LIGHT_UNET_CONFIG, vae=vae)
    pipe.to(dev)
    pipe.unet.init_random_(seed=0)
    vae.init_random_(seed=1)
    pipe.eval().requires_grad_(False)

    # ---- synthetic AudioCaps-shaped inputs, resident in HBM
    g = torch.Generator(device="cpu").manual_seed(3 + rank)
    enc = (torch.randn(B, L, 1024, generator=g) * 0.25).to(dev)
    lens = torch.randint(6, L + 1, (B,), generator=g)
    mask = (torch.arange(L)[None, :] < lens[:, None]).to(dev)
    noise = torch.randn(B, 8, 256, 16, generator=torch.Generator(device="cpu").manual_seed(4 + rank)).to(dev)
    scratch = torch.empty(4, dtype=torch.float32, device=dev)
    state = {}

    def step():
        lat = pipe.generate_latent(enc, mask, noise, cfg_scale_input=4.0, cfg_scale_post=1.0, num_steps=1)
        mel = vae.decode_first_stage(lat)
        wav = vae.vocode(mel)
        pcm = state.get("pcm")
        if pcm is None:
            pcm = state["pcm"] = torch.empty(wav.shape, dtype=torch.int16, device=dev)
        N.check(N.lib().ctta_wav_finalize(N.ptr(wav), wav.numel(), N.ptr(scratch), None, N.ptr(pcm), N.stream_ptr()))
        return lat, mel, wav, pcm

    # The step as the product runs it for fixed shapes: ONE hipGraph replay of U-Net query -> VAE decoder -> HiFi-GAN -> int16
    # (ConsistencyTTA.capture_graph: handles never allocate or synchronise inside *_forward, so the ~370 launches of a
    # batch-32 step replay as one submission; bit-identical to the eager launches, asserted here).  CTTA_BENCH_GRAPH=0, or a
    # failed capture, times the eager launches instead; both rates are reported.
    timed, launch_mode, genB = step, "eager launches", None
    # --no-latency (the PMC passes of tools/refresh_profiles.sh) times eager launches only: graph replays under
    # `rocprofv3 --pmc` hung on this pool in round 1, and counters should not mix replayed and eager steps
    # Rule for every leg below: a capture / parity check is LOCAL (no collective inside), then the ranks agree
    # (du.all_agree: one MAX all-reduce of an error flag) and all of them -- or none -- enter the timing collectives.
    note = None
    if os.environ.get("CTTA_BENCH_GRAPH", "1") != "0" and not args.no_latency:
        try:
            genB = pipe.capture_graph(B, L, cfg_scale_input=4.0)
            pg = genB(enc, mask, noise)
            torch.cuda.synchronize()
            assert torch.equal(pg, step()[3]), "hipGraph replay differs from the eager step"
        except Exception as exc:   # a failed capture must not cost the headline line
            note, genB = "graph capture failed on rank %d: %s" % (rank, str(exc)[:120]), None
        if du.all_agree(genB is not None, dev):
            timed, launch_mode = (lambda: genB(enc, mask, noise)), "one hipGraph replay per step"
        else:
            launch_mode, genB = "eager launches (%s)" % (note or "graph capture failed on another rank"), None

    def time_loop(fn):
        for _ in range(args.warmup):
            fn()
        du.barrier(dev)
        t0_ = time.perf_counter()
        for _ in range(args.steps):
            fn()
        du.barrier(dev)
        return du.max_over_ranks(time.perf_counter() - t0_, dev)

    dt = dt_graph1 = time_loop(timed)
    dt_eager = time_loop(step)      # always, on every rank
    # The throughput form of the same step: a 3-deep software pipeline over batches (U-Net of batch i, VAE decoder of batch
    # i - 1, HiFi-GAN of batch i - 2 as three hipGraphs on three streams per step; ConsistencyTTA.capture_pipeline).  Every
    # timed step runs every stage once on a full batch; a batch's waveforms leave two steps after its text states entered.
    placement_gen = None
    if genB is not None and os.environ.get("CTTA_BENCH_PIPELINE", "1") != "0":      # same decision on every rank (agreed above)
        genP, note = None, None
        try:
            genP = pipe.capture_pipeline(B, L, cfg_scale_input=4.0)
            for _ in range(3):
                pp = genP(enc, mask, noise)
            torch.cuda.synchronize()
            assert torch.equal(pp, step()[3]), "the pipelined replay differs from the eager step"
        except Exception as exc:
            note, genP = "pipelined capture failed on rank %d: %s" % (rank, str(exc)[:100]), None
        if du.all_agree(genP is not None, dev):
            dt_pipe = time_loop(lambda: genP(enc, mask, noise))
            placement_gen = genP.placement_ms
            if dt_pipe < dt:
                dt = dt_pipe
                launch_mode = ("three hipGraph replays per step on three streams: U-Net(batch i) | VAE decoder(batch i-1) | "
                               "HiFi-GAN + int16(batch i-2), handed over by device copies at the step boundary")
        else:
            launch_mode += " (%s)" % (note or "pipelined capture failed on another rank")
        del genP
    del timed, genB
    out = step()
    lat, mel, wav, pcm = out
    assert bool(torch.isfinite(wav).all()), "non-finite waveform"
    clips_per_s = world * B * args.steps / dt

    result = {
        "metric": "10s_audio_clips_per_sec_1step_gen", "value": round(clips_per_s, 3), "unit": "clips/s",
        "n_gpus": world, "rccl_ranks": args.rccl_ranks, "collective_backend": "rccl (torch.distributed 'nccl')" if args.backend == "nccl" else args.backend,
        "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "configs[1]: batch-32 single-step consistency inference, U-Net(light, 559M) + "
                               "AudioLDM VAE decoder + HiFi-GAN, 10 s clips, L=32 text tokens, w=4, T5 excluded",
                   "batch_per_gpu": B, "global_batch": B * world, "text_len": L, "latent": [8, 256, 16],
                   "waveform_samples": int(wav.shape[1]), "weights": "random-init (no checkpoints offline)",
                   "parallelism": "replicas x%d (clips sharded, no data-path collective)" % world,
                   "launch": launch_mode},
        "eager_clips_per_s": round(world * B * args.steps / dt_eager, 3),
        "single_graph_clips_per_s": round(world * B * args.steps / dt_graph1, 3),
        "stage_stream_placement_ms": placement_gen,
    }

    minimal = os.environ.get("CTTA_BENCH_MINIMAL", "0") == "1"   # profiling aid: the timed loops only (a kernel table with a known step count)
    if rank == 0 and not minimal:
        # ---- per-stage split (HIP events on the current stream)
        # median of five eager passes (one pass is +-2 ms on the U-Net's ~500 launches; round 3 reported single passes)
        passes = []
        for _ in range(5):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
            l_ = pipe.generate_latent(enc, mask, noise, cfg_scale_input=4.0, cfg_scale_post=1.0, num_steps=1)
            ev[1].record()
            m_ = vae.decode_first_stage(l_)
            ev[2].record()
            vae.vocode(m_)
            ev[3].record()
            torch.cuda.synchronize()
            passes.append([ev[i].elapsed_time(ev[i + 1]) for i in range(3)])
        med = [sorted(p[i] for p in passes)[2] for i in range(3)]
        result["stage_ms"] = {"unet": round(med[0], 3), "vae_decoder": round(med[1], 3), "hifigan": round(med[2], 3)}
        # the same three stages, each as its OWN hipGraph replay (what the headline's single graph is made of; the eager
        # figures above carry the host's launch cadence of ~500 / ~130 / ~190 launches): median of five replays
        try:
            def graphed(fn):
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    out = fn()                      # warm pass on the capture stream
                torch.cuda.current_stream(dev).wait_stream(side)
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=side):
                    out = fn()
                ts = []
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    gr.replay()
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1))
                return out, sorted(ts)[2]
            with torch.no_grad():
                lg, t_u = graphed(lambda: pipe.generate_latent(enc, mask, noise, cfg_scale_input=4.0, cfg_scale_post=1.0, num_steps=1))
                mg, t_v = graphed(lambda: vae.decode_first_stage(lg))
                _, t_h = graphed(lambda: vae.vocode(mg))
            result["stage_ms_graph"] = {"unet": round(t_u, 3), "vae_decoder": round(t_v, 3), "hifigan": round(t_h, 3)}
        except Exception as exc:   # a stage that cannot be captured on this torch build: the eager split stays
            result["stage_ms_graph"] = {"error": str(exc)[:200]}
        # whole-stage fraction of the dense bf16 peak: ALL algorithmic FLOPs of the stage (convs, linears, attention) over
        # the stage's wall time (eager launches, every kernel of the stage included) -- the weakest stage at a glance
        gf_stage = {"unet": GF_UNET_CONV + GF_UNET_LINEAR_L16 + GF_UNET_LINEAR_PER_16TOK * max(0, (L - 16) / 16.0) + GF_UNET_SELF_ATTN + GF_UNET_CROSS_ATTN_L16 * L / 16.0,
                    "vae_decoder": GF_VAE_CONV + GF_VAE_ATTN, "hifigan": GF_HIFIGAN}
        stage_frac = {k: round(gf_stage[k] * 1e9 * B / (result["stage_ms"][k] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4) for k in gf_stage}
        # ---- roofline of the dominant kernel family, measured live with HIP events per launch
        import ctypes
        L_ = N.lib()
        nprof = 2
        L_.ctta_prof_enable(1)
        for _ in range(nprof):
            step()
        torch.cuda.synchronize()
        L_.ctta_prof_enable(0)
        ms, fl, cnt = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        csv = args.profile_csv.encode() if args.profile_csv else None
        # attention first (keeps the CSV complete), then conv_gemm
        N.check(L_.ctta_prof_collect(-1, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(cnt), csv))
        all_ms, all_fl, all_cnt = ms.value / nprof, fl.value / nprof, cnt.value // nprof
        # second pass restricted to conv_gemm launches
        L_.ctta_prof_enable(1)
        for _ in range(nprof):
            step()
        torch.cuda.synchronize()
        L_.ctta_prof_enable(0)
        N.check(L_.ctta_prof_collect(0, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(cnt), None))
        conv_ms, conv_exec_fl, conv_cnt = ms.value / nprof, fl.value / nprof, cnt.value // nprof
        gf_clip = (GF_UNET_CONV + GF_UNET_LINEAR_L16 + GF_UNET_LINEAR_PER_16TOK * max(0, (L - 16) / 16.0)
                   + GF_VAE_CONV + GF_VAE_ATTN + GF_HIFIGAN - GF_SMALL_N)
        algo_flops = gf_clip * 1e9 * B
        achieved = algo_flops / (conv_ms * 1e-3) / 1e12
        result["roofline"] = {
            "kernel": "conv_gemm_kernel family (implicit-GEMM conv / linear / bmm + the fused vocoder ResBlock units + the fused transformer feed-forward, v_mfma_f32_16x16x32_bf16)",
            "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": None, "traffic_detail": pmc_traffic(B),
            "algorithmic_gflop_per_clip": round(gf_clip, 1), "launches_per_step": int(conv_cnt),
            "kernel_ms_per_step": round(conv_ms, 3), "avg_launch_ms": round(conv_ms / max(1, conv_cnt), 4),
            "executed_tflops_incl_padding": round(conv_exec_fl / (conv_ms * 1e-3) / 1e12, 2),
            "share_of_step_time": round(conv_ms / (dt / args.steps * 1e3), 3),
            "mfma_launch_ms_per_step_all_kinds": round(all_ms, 3),
            "stage_frac": stage_frac,
            # the same fractions over each stage replayed alone as a hipGraph (no host gaps): what the kernels of the stage reach
            "stage_frac_graph": ({k: round(gf_stage[k] * 1e9 * B / (result["stage_ms_graph"][k] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4)
                                  for k in gf_stage} if "error" not in result.get("stage_ms_graph", {"error": 1}) else None),
            # ALL algorithmic FLOPs of the clip (attention included) / the HEADLINE step time (every kernel, every gap)
            "frac_end_to_end": round(sum(gf_stage.values()) * 1e9 * B / (dt / args.steps) / 1e12 / PEAK_BF16_TFLOPS, 4),
        }
        td = result["roofline"]["traffic_detail"]
        if td and conv_cnt:   # per launch like `achieved`: HBM-side bytes of the family per step / its launches per step
            result["roofline"]["traffic"] = int((td["read_GB_per_step"] + td["write_GB_per_step"]) * 1e9 / conv_cnt)
            result["roofline"]["traffic_unit"] = ("bytes per conv_gemm launch (PMC FETCH_SIZE x2 + WRITE_SIZE of the family "
                                                  "per step / launches per step) -- the counter bytes are READ FROM THE COMMITTED %s "
                                                  "(two separate rocprofv3 --pmc passes over this command: counters cannot be collected "
                                                  "inside the timed run), the launch count is this run's; algorithmic operand bytes per "
                                                  "launch: %d" % (td.get("source"), int(GB_CLIP_FUSED * 1e9 * B / conv_cnt)))
        # ---- the HBM-bound kernel class of the clip: GroupNorm + SiLU at the two largest layer shapes,
        # priced on the layer-boundary bytes (read x once, write y once; the statistics pass re-reads x)
        def gn_pass(tag, HW, C):
            Lb = N.lib()
            x = torch.randn(B, HW, C, device=dev).to(torch.bfloat16)
            y = torch.empty_like(x)
            ga, be = torch.ones(C, device=dev), torch.zeros(C, device=dev)
            scr = torch.empty(int(Lb.ctta_groupnorm_scratch_floats(B, HW, C, 32)), dtype=torch.float32, device=dev)

            def fn():
                N.check(Lb.ctta_groupnorm(N.ptr(x), N.ptr(y), B, HW, C, 32, N.ptr(ga), N.ptr(be), 1e-5, 1, N.ptr(scr),
                                          N.stream_ptr()))
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms_ = e0.elapsed_time(e1) / 5
            nbytes = 2 * x.numel() * 2
            gbps = nbytes / (ms_ * 1e-3) / 1e9
            return {"kernel": "groupnorm+silu (gn_partial / gn_finalize / gn_apply)", "shape": tag,
                    "algorithmic_GB_per_launch": round(nbytes / 1e9, 3), "ms": round(ms_, 3),
                    "achieved_GBps": round(gbps, 1), "frac": round(gbps / PEAK_HBM_GBPS, 4)}
        result["hbm_kernels"] = {"bound": "hbm", "peak": PEAK_HBM_GBPS, "unit": "GB/s", "passes": [
            gn_pass("VAE decoder last level (%d, 1024x64, 128)" % B, 65536, 128),
            gn_pass("U-Net level 0 (%d, 256x16, 256)" % B, 4096, 256)]}
        # ---- PCIe-inclusive variant (the reference's decode_to_waveform ends on the host)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(2):
            step()[3].cpu()
        torch.cuda.synchronize()
        result["pcie_inclusive_clips_per_s"] = round(2 * B / (time.perf_counter() - t1), 3)

        # ---- the text encoder in front of the clip (FLAN-T5-large shape, random init): cond + uncond batches of
        # encode_text_classifier_free (audio_distilled_model.py:236-248) on the HIP engine
        try:
            from consistencytta_amd import text_encoder
            te = text_encoder.T5EncoderModel(spec.T5_LARGE_CONFIG).to(dev)
            te.init_random_(seed=7)
            ids = torch.randint(2, 32000, (B, L), generator=torch.Generator().manual_seed(8)).to(dev)
            am = mask.to(torch.int64)
            for _ in range(2):
                te(input_ids=ids, attention_mask=am)
                te(input_ids=torch.ones_like(ids), attention_mask=am)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                te(input_ids=ids, attention_mask=am)
                te(input_ids=torch.ones_like(ids), attention_mask=am)
            torch.cuda.synchronize()
            te_ms = (time.perf_counter() - t1) / 5 * 1e3
            result["text_encoder_ms"] = round(te_ms, 3)
            per_rank = result["value"] / world
            result["clips_per_s_incl_text_encoder"] = round(world * B / (B / per_rank + te_ms * 1e-3), 3)
            del te
        except Exception as exc:
            result["text_encoder_ms"] = {"error": str(exc)[:200]}

        # ---- single-clip latency (configs[0] shape: one prompt, L=16), eager launches vs one hipGraph replay
        try:
            if args.no_latency:
                raise RuntimeError("skipped (--no-latency)")
            e1, m1, n1 = enc[:1, :16].contiguous(), mask[:1, :16].contiguous(), noise[:1].contiguous()
            m1[:] = True
            gen1 = pipe.capture_graph(1, 16, cfg_scale_input=4.0)

            def eager1():
                lat_ = pipe.generate_latent(e1, m1, n1, cfg_scale_input=4.0, cfg_scale_post=1.0, num_steps=1)
                w_ = vae.vocode(vae.decode_first_stage(lat_))
                p_ = torch.empty(w_.shape, dtype=torch.int16, device=dev)
                N.check(N.lib().ctta_wav_finalize(N.ptr(w_), w_.numel(), N.ptr(scratch), None, N.ptr(p_), N.stream_ptr()))
                return p_

            lat_ms = {}
            for name, fn in (("eager", eager1), ("hipgraph", lambda: gen1(e1, m1, n1))):
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(10):
                    fn()
                torch.cuda.synchronize()
                lat_ms[name] = round((time.perf_counter() - t1) / 10 * 1e3, 3)
            assert torch.equal(eager1(), gen1(e1, m1, n1))
            result["single_clip_latency_ms"] = lat_ms
            del gen1
        except Exception as exc:   # a failed capture must not cost the headline line
            result["single_clip_latency_ms"] = {"error": str(exc)[:200]}

        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(pipe, vae, enc, mask, noise)
    if args.mode != "gen":
        del pipe, vae
        state.clear()
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        # a leg that fails the same way on every rank (an allocator or collective set-up error) must not cost the headline
        # line; a ONE-sided failure inside a collective cannot be caught here (the other ranks wait in the all-reduce)
        try:
            d = distill_leg(args, dev, world, rank)
        except Exception as exc:
            d = {"error": str(exc)[:300]}
        if rank == 0:
            result["distill"] = d
        gc.collect()
        torch.cuda.empty_cache()
        try:
            t = teacher_leg(args, dev, world, rank)
        except Exception as exc:
            t = {"error": str(exc)[:300]}
        if rank == 0:
            result["teacher"] = t
        gc.collect()
        torch.cuda.empty_cache()
        if world == 1:   # single-GPU extra: a one-sided failure inside a collective would hang the scaling runs
            try:
                pd_ = distill_leg(args, dev, world, rank, perceptual=True)
            except Exception as exc:   # the newest leg must not cost the headline line
                pd_ = {"error": str(exc)[:300]}
            result["perceptual_distill"] = pd_
    emit(result if rank == 0 else None, rank, dev)
    du.finish()


def teacher_leg(args, dev, world, rank):
    """BASELINE.json configs[2] (SURVEY.md §8d "Config 3"): the multi-step Heun teacher of AudioLCM.inference,
    B clips per GPU (2B with CFG), N Heun steps = 2N-1 U-Net queries, w = 3, the loop replayed from one captured
    hipGraph (models._teacher_loop_graphed).  Light U-Net, random-init weights.  Clips are independent: N GPUs
    run N replicas.  Rank 0 also times the eager (uncaptured) loop for comparison."""
    import torch

    from consistencytta_amd import dist_util as du
    from consistencytta_amd import scheduler as sched_mod
    from consistencytta_amd import spec
    from consistencytta_amd.models import AudioLCM

    B, L, nsteps = args.teacher_batch, args.text_len, args.teacher_steps
    m = AudioLCM(text_encoder_name="google/flan-t5-large", scheduler_name="stabilityai/stable-diffusion-2-1",
                 unet_model_config_path="tango_diffusion_light.json", unet_config=spec.LIGHT_UNET_CONFIG, snr_gamma=5.0,
                 use_edm=True, teacher_guidance_scale=-1, num_diffusion_steps=18, vae=None, loss_type="mse")
    m.to(dev)
    m.teacher_unet.init_random_(seed=10)
    m.student_ema_unet.init_random_(seed=11)
    m.eval()
    g = torch.Generator(device="cpu").manual_seed(7 + rank)
    enc = (torch.randn(B, L, 1024, generator=g) * 0.25).to(dev)
    lens = torch.randint(6, L + 1, (B,), generator=g)
    mask = (torch.arange(L)[None, :] < lens[:, None]).to(dev)
    umask = torch.zeros_like(mask)
    umask[:, 0] = True
    P = {"embeds_cf": torch.cat([torch.zeros_like(enc), enc]), "mask_cf": torch.cat([umask, mask]), "embeds": enc,
         "mask": mask}
    noise = torch.randn(B, 8, 256, 16, generator=g).to(dev)
    sch = sched_mod.HeunDiscreteScheduler.from_pretrained("stabilityai/stable-diffusion-2-1", subfolder="scheduler")
    kw = dict(guidance_scale_input=3.0, guidance_scale_post=1.0, num_steps=1, use_edm=True, use_ema=True,
              query_teacher=True, return_all=True, noise=noise)
    m.inference(P, sch, num_teacher_steps=2, graph_teacher=True, **kw)      # warm-up: handles, capture machinery
    du.barrier(dev)
    t0 = time.perf_counter()
    _, lat, _, _ = m.inference(P, sch, num_teacher_steps=nsteps, graph_teacher=True, **kw)
    du.barrier(dev)
    dt = du.max_over_ranks(time.perf_counter() - t0, dev)
    assert bool(torch.isfinite(lat).all()), "non-finite teacher latent"
    evals = 2 * nsteps - 1
    out = {"metric": "heun_teacher_clips_per_sec", "value": round(world * B / dt, 4), "unit": "clips/s",
           "unet_queries_per_s": round(world * evals / dt, 2), "seconds_per_batch": round(dt, 3), "n_gpus": world,
           "config": {"workload": "configs[2]: %d-step Heun teacher (%d CFG U-Net queries at batch %d), w=3, light U-Net, "
                                  "hipGraph-replayed loop (includes the 1-step student query and graph capture)"
                                  % (nsteps, evals, 2 * B), "batch_per_gpu": B, "text_len": L}}
    if rank == 0:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.inference(P, sch, num_teacher_steps=nsteps, **kw)
        torch.cuda.synchronize()
        out["eager_loop_seconds_per_batch"] = round(time.perf_counter() - t0, 3)
    del m
    return out


class _LegSkipped(RuntimeError):
    """A replayed leg that every rank skips TOGETHER: its capture / parity check failed somewhere (du.all_agree) or its first
    step was many times slower than the eager step (MAX over ranks).  Never raised one-sidedly."""


GF_DISTILL_PER_SAMPLE = 4200.0   # 4 teacher + 1 target + 1 student fwd + 1 student bwd (~2 fwd)


def distill_leg(args, dev, world, rank, perceptual=False):
    """perceptual=True: BASELINE.json configs[4] as far as it can be built offline -- the same distillation step with
    the waveform-domain loss path of the CLAP fine-tuning stage (tools/losses.py:294-298): student latent -> VAE
    decode -> HiFi-GAN with allow_grad=True, loss on the waveform (multi-resolution STFT; CLAP itself needs weights
    that are not available), gradient back through vocoder and decoder (frozen) into the student U-Net.

    Otherwise BASELINE.json configs[3] (SURVEY.md §8d "Config 4"): one consistency-distillation optimisation step
    per GPU micro-batch of 9 latents -- 2 CFG teacher queries (batch 18 each) + Heun, target-network
    forward, student forward + backward, SUM all-reduce of the 559 M fp32 gradients over RCCL, fused
    AdamW (lr 1e-5, wd 1e-4), two-shadow EMA (0.95 / 0.999).  All four U-Nets use the light config and
    random-init weights; z0 ~ N(0,1)*0.9, text states as in the generation leg.  Weak scaling: the
    per-GPU batch is fixed, every rank takes the same number of optimizer steps."""
    import torch

    from consistencytta_amd import _native as N
    from consistencytta_amd import dist_util as du
    from consistencytta_amd import spec
    from consistencytta_amd.models import AudioLCM
    from consistencytta_amd.optim import WarmupSchedule

    B, L = (args.perceptual_batch if perceptual else args.distill_batch), args.text_len
    t_build = time.perf_counter()
    vae = clap = None
    if perceptual:
        from consistencytta_amd import clap as clap_mod
        from consistencytta_amd import modules
        vae = modules.AutoencoderKL(ddconfig=spec.VAE_DDCONFIG, embed_dim=8, scale_factor=0.9227914214134216)
        vae.to(dev)
        vae.init_random_(seed=12)
        vae.eval().requires_grad_(False)
        clap = clap_mod.CLAP_Module(enable_fusion=False, amodel="HTSAT-base")     # HTSAT-base + roberta-base, random init
        clap.to(dev)
        clap.model.init_random_(seed=13)
    m = AudioLCM(text_encoder_name="google/flan-t5-large", scheduler_name="stabilityai/stable-diffusion-2-1",
                 unet_model_config_path="tango_diffusion_light.json", unet_config=spec.LIGHT_UNET_CONFIG, snr_gamma=5.0,
                 use_edm=True, teacher_guidance_scale=-1, num_diffusion_steps=18, vae=vae,
                 loss_type="clap" if perceptual else "mse", clap_module=clap, target_ema_decay=0.95, ema_decay=0.999)
    m.to(dev)
    m.teacher_unet.init_random_(seed=10)
    m.student_unet.init_random_(seed=11)
    with torch.no_grad():   # load_state_dict_from_tango starts target / EMA from the student's weights
        for dst in (m.student_target_unet, m.student_ema_unet):
            for p, q in zip(dst.parameters(), m.student_unet.parameters()):
                p.copy_(q)
    m.train()
    opt = m.prepare_training(lr=1e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4, broadcast=True)
    sched = WarmupSchedule(opt, "linear", num_warmup_steps=1000, num_training_steps=100000)
    g = torch.Generator(device="cpu").manual_seed(5 + rank)
    z0 = (torch.randn(B, 8, 256, 16, generator=g) * 0.9).to(dev)
    enc = (torch.randn(B, L, 1024, generator=g) * 0.25).to(dev)
    lens = torch.randint(6, L + 1, (B,), generator=g)
    mask = (torch.arange(L)[None, :] < lens[:, None]).to(dev)
    unc = torch.zeros_like(enc)
    umask = torch.zeros_like(mask)
    umask[:, 0] = True
    P = {"embeds_cf": torch.cat([unc, enc]), "mask_cf": torch.cat([umask, mask]), "embeds": enc, "mask": mask}
    step_kw = {}
    if perceptual:   # CLAPLoss's other two inputs: ground-truth audio (10 s at 16 kHz) and the captions' CLAP text features,
        # the latter from the RoBERTa tower on synthetic token ids (77 positions, ragged lengths, tokenizer files are offline)
        ids = torch.randint(4, 50000, (B, 77), generator=g)
        tl = torch.randint(5, 30, (B,), generator=g)
        tmask = (torch.arange(77)[None, :] < tl[:, None]).long()
        ids = torch.where(tmask == 1, ids, torch.ones_like(ids))
        P["clap_text_features"] = clap.model.get_text_embedding({"input_ids": ids.to(dev), "attention_mask": tmask.to(dev)})
        step_kw["gt_wav"] = (torch.rand(B, 160000, generator=g) * 2 - 1).to(dev) * 0.3
    torch.manual_seed(100 + rank)       # per-rank timestep / guidance / noise streams
    build_s = time.perf_counter() - t_build
    trace("distill%s: models built in %.1f s, eager loop next" % (" (perceptual)" if perceptual else "", build_s))

    # the extra legs time >= 10 steps behind >= 3 warm-up steps whatever K / W the headline uses (the first steps of a
    # training loop pay allocator and clock ramp-up: 119 vs 112 ms per step at 5 / 2 against 10 / 3 on the same box)
    n_steps, n_warm = (max(args.steps, 10), max(args.warmup, 3)) if args.mode not in ("distill", "perceptual") else (args.steps, max(1, args.warmup))
    # (--mode distill / perceptual, the profiling aids, keep exactly K / W: tools/refresh_profiles.sh counts their steps)
    losses = []

    def fixed_draw_loss():
        """The consistency loss of ONE fixed draw (timesteps, noise, guidance scales) with the weights as they are now: the
        per-step losses use fresh random timesteps, so first-vs-last of those is noise; this pair is comparable."""
        if perceptual or args.no_latency:      # (--no-latency = the PMC passes: training steps only in the trace)
            return None
        gfx = torch.Generator().manual_seed(4242 + rank)
        ti = torch.randint(0, 17, (B,), generator=gfx) * 2
        gn = torch.randn(B, 8, 256, 16, generator=gfx).to(dev)
        gsc = torch.rand(B, generator=gfx) * 6
        with torch.no_grad():
            return round(float(m._forward_impl(z0, None, P, False, True, ti, gn, gsc, False)), 6)
    loss_fixed_before = fixed_draw_loss()
    for _ in range(n_warm):
        losses.append(m.train_step(z0, P, opt, sched, **step_kw))
    du.barrier(dev)
    t0 = time.perf_counter()
    for _ in range(n_steps):
        losses.append(m.train_step(z0, P, opt, sched, **step_kw))
    du.barrier(dev)
    dt = du.max_over_ranks(time.perf_counter() - t0, dev)
    trace("distill%s: eager loop done" % (" (perceptual)" if perceptual else ""))
    assert all(v == v for v in losses), "NaN distillation loss"
    # The step as the product runs it for fixed shapes (AudioLCM.capture_train_graph; bit-identical to train_step:
    # tests/test_train_gpu.py, and checked here against an eager forward with the same draws): hipGraph replays of noising,
    # teacher queries, target network, student forward + backward and loss, then AdamW / zero_grad / EMA eager.  Four forms
    # are timed: monolithic (one graph), segmented (one graph per all-reduce bucket -- the form a process group needs, the
    # collectives are issued between the replays), and each of them with the frozen teacher's phase of the NEXT batch as its
    # own graph on a second stream (pipelined).  Headline: one GPU -> the fastest monolithic form; world > 1 -> the fastest of
    # eager / segmented / segmented + pipelined.
    dt_eager, launch_mode = dt, "eager launches (two streams + weight-gradient side stream)"
    dt_seg = dt_graph = dt_pipe = dt_seg_pipe = None
    placements = []
    timed_graph = None
    if not perceptual and os.environ.get("CTTA_BENCH_GRAPH", "1") != "0" and not args.no_latency:
        gdr = torch.Generator().manual_seed(77 + rank)
        kw = dict(time_inds=torch.randint(0, 17, (B,), generator=gdr) * 2,
                  gaussian_noise=torch.randn(B, 8, 256, 16, generator=gdr).to(dev), guidance_scale=torch.rand(B, generator=gdr) * 6)

        def timed_graph(segmented, pipelined=False, z=None, prompt=None, draws=None, steps=None, warm=None):
            """Phase 1 (LOCAL, no collective inside): capture, check one replay against an eager forward with the same draws.
            Then the ranks agree (du.all_agree) -- all of them time the leg or none does; a one-sided capture failure or
            parity assert therefore never leaves the peers alone inside a barrier or a bucket all-reduce.  Phase 2: one
            public step timed on every rank (MAX) -- a transport that cannot keep up between the replays (the gloo
            rehearsal with two ranks on one GPU took 50 s per segmented step) skips the leg collectively -- then warm-up
            and the timed steps.  Returns seconds for `steps` steps, or raises _LegSkipped on EVERY rank.  (One rank alone has
            no peer to leave hanging: there ANY exception of phase 2 becomes a skipped leg instead of costing the line.)"""
            trace("distill: replayed form segmented=%s pipelined=%s batch=%s" % (segmented, pipelined, "fused" if z is not None else "micro"))
            z = z0 if z is None else z
            prompt = P if prompt is None else prompt
            draws = kw if draws is None else draws
            steps = n_steps if steps is None else steps
            warm = n_warm if warm is None else warm
            gs, why = None, None
            try:
                gs = m.capture_train_graph(opt, z, prompt, segmented=segmented, pipeline_teacher=pipelined, **draws)
                with torch.no_grad():
                    loss_e = float(m._forward_impl(z, None, prompt, False, True, draws["time_inds"], draws["gaussian_noise"],
                                                   draws["guidance_scale"], True)[0])
                if pipelined:
                    gs.feed(z, **draws)       # the teacher phase of this batch on the teacher stream ...
                    gs.feed(z, **draws)       # ... becomes the current set; the same batch queued again behind it
                else:
                    gs._refresh(z, draws["time_inds"], draws["gaussian_noise"], draws["guidance_scale"])
                gs.replay()
                loss_g = float(gs.loss.item())
                opt.zero_grad()
                assert loss_g == loss_e, "hipGraph replay of the distillation step differs from the eager step (%r vs %r)" % (loss_g, loss_e)
            except Exception as exc:
                why, gs = "rank %d: %s" % (rank, str(exc)[:160]), None
                opt.zero_grad()
            if not du.all_agree(gs is not None, dev):
                raise _LegSkipped("capture / parity check failed (%s)" % (why or "on another rank"))
            if pipelined:
                placements.append(getattr(gs, "placement_ms", None))
            try:
                du.barrier(dev)
                t1 = time.perf_counter()
                losses.append(gs.step(z, sched))
                du.barrier(dev)
                one = du.max_over_ranks(time.perf_counter() - t1, dev)
                if one > 6.0 * dt_eager / n_steps * max(1.0, z.shape[0] / float(B)):
                    raise _LegSkipped("first replayed step took %.0f ms against %.0f ms eager" % (one * 1e3, dt_eager / n_steps * 1e3))
                for _ in range(warm):
                    losses.append(gs.step(z, sched))
                du.barrier(dev)
                t0 = time.perf_counter()
                for _ in range(steps):
                    losses.append(gs.step(z, sched))
                du.barrier(dev)
                return du.max_over_ranks(time.perf_counter() - t0, dev)
            except _LegSkipped:
                raise
            except Exception as exc:
                if world > 1:      # one-sided: ending the rank (torchrun then stops the job) beats a hang
                    raise
                opt.zero_grad()
                raise _LegSkipped("replayed steps failed: %s" % str(exc)[:160])
        pipe_on = os.environ.get("CTTA_BENCH_PIPELINE", "1") != "0"
        # CTTA_BENCH_DISTILL_FORMS (profiling aid): which replayed forms are timed -- seg, segpipe, graph, pipe; "accum" keeps the
        # accumulation legs.  Default: all.  One form per rocprofv3 run gives a kernel table that belongs to ONE launch form.
        forms = set(os.environ.get("CTTA_BENCH_DISTILL_FORMS", "seg,segpipe,graph,pipe,accum").split(","))   # "eager": none of them
        pipe_txt = ("; the frozen teacher's two CFG queries + Heun step run as their own hipGraph on a second stream for batch "
                    "i + 1 beside the student / target / backward work of batch i (every timed step holds one teacher phase, "
                    "one target forward, one student forward + backward, AdamW, EMA)")
        # _LegSkipped is raised on every rank or on none (decided by a collective), so the except branches below are
        # entered together; any OTHER exception inside phase 2 is one-sided and must end the rank (torchrun then stops
        # the job): failing loudly beats a hang.
        launch_seg = ("8 hipGraph replays per micro-step (forward + loss + the first bucket's blocks | one graph per further "
                      "bucket of the gradient all-reduce, which is issued between replays) + eager AdamW / zero_grad / EMA")
        try:
            if "seg" in forms or world > 1:
                dt_seg = timed_graph(True)
                if world > 1 and dt_seg < dt:
                    dt, launch_mode = dt_seg, launch_seg
        except _LegSkipped as exc:
            launch_mode = "eager launches (segmented replay skipped: %s)" % exc
        if pipe_on and (dt_seg is not None or "seg" not in forms) and ("segpipe" in forms or world > 1):   # what the data-parallel step runs: segmented (bucket all-reduce between replays) AND pipelined
            try:
                dt_seg_pipe = timed_graph(True, True)
                if world > 1 and dt_seg_pipe < dt:
                    dt, launch_mode = dt_seg_pipe, launch_seg + pipe_txt
            except _LegSkipped as exc:
                launch_mode += " (segmented + pipelined replay skipped: %s)" % exc
        if world == 1 and ("graph" in forms or "pipe" in forms):
            try:
                mono_txt = "one hipGraph replay per micro-step (forward + backward + loss) + the optimizer tail (AdamW, zero_grad, EMA: one launch)"
                if "graph" in forms:
                    dt = dt_graph = timed_graph(False)
                    launch_mode = mono_txt
                if pipe_on and "pipe" in forms:
                    try:
                        dt_pipe = timed_graph(False, True)
                        if dt_pipe < dt:
                            dt, launch_mode = dt_pipe, mono_txt + pipe_txt
                    except _LegSkipped as exc:
                        launch_mode += " (pipelined capture skipped: %s)" % exc
            except _LegSkipped as exc:   # a failed capture must not cost the line
                launch_mode = "eager launches (graph capture skipped: %s)" % exc
                dt = dt_eager
    if perceptual and os.environ.get("CTTA_BENCH_PIPELINE", "1") != "0" and not args.no_latency:
        # configs[4]: the teacher phase as one hipGraph on its own stream for batch i + 1; the rest of batch i -- target network,
        # student forward, decode + vocoder + CLAP towers, torch's backward through the loss down to the latent, the engine's
        # backward -- as a second hipGraph (round 6: `main_eager=False`; tests/test_clap_gpu.py checks replay == eager).
        # CTTA_BENCH_PERCEPTUAL_GRAPH=0: only the teacher phase replayed, the rest eager launches (rounds 4-5)
        gs, why = None, None
        try:      # phase 1, local: the teacher-phase capture
            gdr = torch.Generator().manual_seed(78 + rank)
            kw = dict(time_inds=torch.randint(0, 17, (B,), generator=gdr) * 2,
                      gaussian_noise=torch.randn(B, 8, 256, 16, generator=gdr).to(dev), guidance_scale=torch.rand(B, generator=gdr) * 6)
            gs = m.capture_train_graph(opt, z0, P, pipeline_teacher=True, gt_wav=step_kw["gt_wav"],
                                       main_eager=None if os.environ.get("CTTA_BENCH_PERCEPTUAL_GRAPH", "1") == "0" else False, **kw)
        except Exception as exc:
            why, gs = "rank %d: %s" % (rank, str(exc)[:300]), None
            trace("perceptual capture failed: %s" % why)
        if du.all_agree(gs is not None, dev):      # phase 2 on every rank, or on none
            placements.append(getattr(gs, "placement_ms", None))
            for _ in range(n_warm):
                losses.append(gs.step(z0, sched, gt_wav=step_kw["gt_wav"]))
            du.barrier(dev)
            t0 = time.perf_counter()
            for _ in range(n_steps):
                losses.append(gs.step(z0, sched, gt_wav=step_kw["gt_wav"]))
            du.barrier(dev)
            dt_pipe = du.max_over_ranks(time.perf_counter() - t0, dev)
            if not gs.main_eager:
                dt_graph = dt_pipe       # the replayed form of this leg IS the pipelined one (there is no unpipelined capture of it)
            if dt_pipe < dt:
                dt = dt_pipe
                launch_mode = (("one hipGraph replay for target network, student forward, decode, CLAP loss and the backward through all "
                                "of them" if not gs.main_eager else
                                "eager launches (two streams + weight-gradient side stream) for target network, student forward, decode, "
                                "CLAP loss and backward") +
                               "; the frozen teacher's two CFG queries + Heun step as one hipGraph on a second stream for batch i + 1")
        else:
            launch_mode += " (pipelined teacher skipped: %s)" % (why or "capture failed on another rank")
        del gs
    assert all(v == v for v in losses), "NaN distillation loss"
    out = {
        "metric": "distillation_steps_per_sec", "value": round(n_steps / dt, 4), "unit": "optimizer steps/s",
        "samples_per_s": round(world * B * n_steps / dt, 3), "ms_per_step": round(dt / n_steps * 1e3, 3),
        "steps": n_steps, "warmup": n_warm,
        "eager_ms_per_step": round(dt_eager / n_steps * 1e3, 3),
        "segmented_ms_per_step": None if dt_seg is None else round(dt_seg / n_steps * 1e3, 3),
        "segmented_pipelined_ms_per_step": None if dt_seg_pipe is None else round(dt_seg_pipe / n_steps * 1e3, 3),
        "teacher_stream_placement_ms": placements or None,   # (teacher graph || main graph) per candidate stream, see _place_teacher_stream
        "graph_ms_per_step": None if dt_graph is None else round(dt_graph / n_steps * 1e3, 3),
        "pipelined_ms_per_step": None if dt_pipe is None else round(dt_pipe / n_steps * 1e3, 3),
        "n_gpus": world, "scaling": "weak", "dtype": "bf16 (fp32 master weights, gradients, AdamW moments)",
        "config": {"workload": "configs[3]: consistency distillation step, light U-Net x4 (teacher, student, target, EMA), "
                               "2 CFG teacher queries + Heun, SNR-MSE loss, backward, AdamW, EMA 0.95/0.999",
                   "batch_per_gpu": B, "global_batch": B * world, "text_len": L, "latent": [8, 256, 16],
                   "launch": launch_mode, "grad_accum": 1, "gradient_allreduce": ("fp32 SUM over RCCL, one asynchronous collective per finished backward block, blocks merged to >= 16 M "
                                          "elements (64 MiB), overlapped with the rest of the backward") if world > 1 else "none (1 GPU)",
                   "parallelism": "dp%d" % world},
        "build_s": round(build_s, 1),
    }
    if not perceptual:   # (round 4 printed first / last of the per-step losses: random timesteps per step made that pair noise)
        out["fixed_draw_loss_before_after"] = [loss_fixed_before, fixed_draw_loss()]
        out["fixed_draw_loss_note"] = ("consistency loss of one fixed (timestep, noise, guidance) draw before the first and after the "
                                       "last optimizer step of this leg's main loops (%d updates at lr <= 1e-5 * step / 1000)" % len(losses))
    if perceptual:
        out["metric"] = "perceptual_distillation_steps_per_sec"
        out["config"]["workload"] = ("configs[4]: CLAP fine-tuning step -- consistency generation (student / teacher / target "
                                     "U-Nets as in configs[3]), VAE decode + HiFi-GAN with allow_grad=True, 16->48 kHz "
                                     "Kaiser-sinc resampling, CLAP forward (HTSAT-base audio tower on the generated and the "
                                     "ground-truth clip, RoBERTa-base text features) and its input gradient back through "
                                     "vocoder, decoder and U-Net; CLAPLoss(mse 1.0, clap 0.1); AdamW, EMA; random-init CLAP "
                                     "weights (no checkpoint offline)")
        # train.sh:38-46 reaches its 45 samples per optimizer step as 15 accumulated micro-batches of 3 per GPU -- what fits a
        # 40-80 GB card next to four U-Nets, the decoder, the vocoder and two CLAP towers with their activations.  One MI355X
        # holds far more: the same optimizer step as FEWER, LARGER micro-batches (same samples per step, same mathematics:
        # tests/test_train_gpu.py::test_fused_accumulation...), eager launches.  --perceptual-fused-batch sets the micro-batch.
        Bf = int(args.perceptual_fused_batch)
        if Bf > B and not args.no_latency and os.environ.get("CTTA_BENCH_FUSED_ACCUM", "1") != "0":
            zf = Pf = gtf = why = None
            try:
                gf = torch.Generator(device="cpu").manual_seed(56 + rank)
                zf = (torch.randn(Bf, 8, 256, 16, generator=gf) * 0.9).to(dev)
                encf = (torch.randn(Bf, L, 1024, generator=gf) * 0.25).to(dev)
                lensf = torch.randint(6, L + 1, (Bf,), generator=gf)
                maskf = (torch.arange(L)[None, :] < lensf[:, None]).to(dev)
                uncf, umf = torch.zeros_like(encf), torch.zeros_like(maskf)
                umf[:, 0] = True
                idsf = torch.randint(4, 50000, (Bf, 77), generator=gf)
                tlf = torch.randint(5, 30, (Bf,), generator=gf)
                tmf = (torch.arange(77)[None, :] < tlf[:, None]).long()
                idsf = torch.where(tmf == 1, idsf, torch.ones_like(idsf))
                Pf = {"embeds_cf": torch.cat([uncf, encf]), "mask_cf": torch.cat([umf, maskf]), "embeds": encf, "mask": maskf,
                      "clap_text_features": clap.model.get_text_embedding({"input_ids": idsf.to(dev), "attention_mask": tmf.to(dev)})}
                gtf = (torch.rand(Bf, 160000, generator=gf) * 2 - 1).to(dev) * 0.3
            except Exception as exc:
                why, zf = "rank %d: %s" % (rank, str(exc)[:160]), None
            if du.all_agree(zf is not None, dev):
                ok, n_f = True, 3
                try:
                    for _ in range(2):
                        losses.append(m.train_step(zf, Pf, opt, sched, gt_wav=gtf))
                    du.barrier(dev)
                    t0 = time.perf_counter()
                    for _ in range(n_f):
                        losses.append(m.train_step(zf, Pf, opt, sched, gt_wav=gtf))
                    du.barrier(dev)
                    dtf = du.max_over_ranks(time.perf_counter() - t0, dev)
                except Exception as exc:      # (an out-of-memory at this size must not cost the line on one GPU)
                    if world > 1:
                        raise
                    ok, why = False, str(exc)[:200]
                if ok:
                    out["fused_micro_batch"] = {"micro_batch_per_gpu": Bf, "samples_per_s": round(world * Bf * n_f / dtf, 3),
                                                "ms_per_step": round(dtf / n_f * 1e3, 3), "steps": n_f, "launch": "eager launches",
                                                "frac_end_to_end_unet_only": round(GF_DISTILL_PER_SAMPLE * 1e9 * Bf / (dtf / n_f) / 1e12 / PEAK_BF16_TFLOPS, 4),
                                                "note": "%d samples as ONE micro-batch (train.sh accumulates 15 x 3 per GPU); samples_per_s of the "
                                                        "headline above: %.1f" % (Bf, world * B * n_steps / dt)}
                else:
                    out["fused_micro_batch"] = {"skipped": why}
            else:
                out["fused_micro_batch"] = {"skipped": why or "input set-up failed on another rank"}
            del zf, Pf, gtf
        del m, opt, vae, clap
        return out
    # train.sh:33's recipe accumulates 5 micro-batches per optimizer step (SURVEY 8d "grad-accum 1 and 5"): 4 local
    # micro-steps (loss + backward, DDP no_sync) and a 5th that also all-reduces, steps AdamW and updates the EMAs
    acc, n_opt = 5, max(2, args.steps // 5)
    forms_env = set(os.environ.get("CTTA_BENCH_DISTILL_FORMS", "accum").split(","))
    accum_on = "accum" in forms_env                      # profiling aids: "eager" / "pipe" ... alone = no accumulation legs
    if "eager" in forms_env and len(forms_env) == 1:     # ... and "eager" alone ends here: a kernel table of the main loop only
        del m, opt
        return out
    if accum_on:
        m._micro = 0
        for _ in range(acc):
            m.train_step(z0, P, opt, sched, accumulation_steps=acc)
        du.barrier(dev)
        t0 = time.perf_counter()
        for _ in range(acc * n_opt):
            losses.append(m.train_step(z0, P, opt, sched, accumulation_steps=acc))
        du.barrier(dev)
        dt5 = du.max_over_ranks(time.perf_counter() - t0, dev)
        m._micro = 0
        assert all(v == v for v in losses), "NaN distillation loss"
        out["grad_accum_5"] = {"value": round(n_opt / dt5, 4), "unit": "optimizer steps/s", "optimizer_steps": n_opt,
                               "micro_steps_per_s": round(acc * n_opt / dt5, 3), "samples_per_s": round(world * B * acc * n_opt / dt5, 3),
                               "ms_per_optimizer_step": round(dt5 / n_opt * 1e3, 3), "global_batch": B * world * acc}
    # The same optimizer step -- 5 x 9 samples per GPU -- as ONE micro-batch of 45: accumulation exists in train.sh because the
    # reference's GPUs cannot hold more than 9 samples; one MI355X holds the activations of 45 (about 30 of its 288 GB).  Same
    # samples per optimizer step and the same mathematics (tests/test_train_gpu.py::test_fused_accumulation...); 5x larger
    # GEMMs per launch.  Timed with eager launches (every rank: the all-reduce sequence must not depend on a capture).
    if accum_on and os.environ.get("CTTA_BENCH_FUSED_ACCUM", "1") != "0" and not args.no_latency:
        Bf = B * acc
        trace("distill: grad_accum fused micro-batch legs")
        z45 = P45 = kw45 = why = None
        try:      # phase 1, local: the inputs (the arenas of the four U-Nets grow inside the first train_step)
            gf = torch.Generator(device="cpu").manual_seed(55 + rank)
            z45 = (torch.randn(Bf, 8, 256, 16, generator=gf) * 0.9).to(dev)
            enc45 = (torch.randn(Bf, L, 1024, generator=gf) * 0.25).to(dev)
            lens45 = torch.randint(6, L + 1, (Bf,), generator=gf)
            mask45 = (torch.arange(L)[None, :] < lens45[:, None]).to(dev)
            unc45, um45 = torch.zeros_like(enc45), torch.zeros_like(mask45)
            um45[:, 0] = True
            P45 = {"embeds_cf": torch.cat([unc45, enc45]), "mask_cf": torch.cat([um45, mask45]), "embeds": enc45, "mask": mask45}
            kw45 = dict(time_inds=torch.randint(0, 17, (Bf,), generator=gf) * 2,
                        gaussian_noise=torch.randn(Bf, 8, 256, 16, generator=gf).to(dev), guidance_scale=torch.rand(Bf, generator=gf) * 6)
        except Exception as exc:
            why, z45 = "rank %d: %s" % (rank, str(exc)[:160]), None
        if du.all_agree(z45 is not None, dev):      # every rank enters the steps (and their all-reduces), or none does;
            # an exception INSIDE them is one-sided and ends the rank -- torchrun then stops the job: loud, not hung
            for _ in range(2):
                losses.append(m.train_step(z45, P45, opt, sched))
            du.barrier(dev)
            t0 = time.perf_counter()
            for _ in range(n_opt):
                losses.append(m.train_step(z45, P45, opt, sched))
            du.barrier(dev)
            dtf = du.max_over_ranks(time.perf_counter() - t0, dev)
            out["grad_accum_5_fused"] = {"value": round(n_opt / dtf, 4), "unit": "optimizer steps/s", "optimizer_steps": n_opt,
                                         "samples_per_s": round(world * Bf * n_opt / dtf, 3),
                                         "ms_per_optimizer_step": round(dtf / n_opt * 1e3, 3), "global_batch": Bf * world,
                                         "micro_batch_per_gpu": Bf, "launch": "eager launches",
                                         "note": "the 45 samples of one optimizer step as ONE micro-batch (no accumulation loop)"}
            # The training form this machine wants: the same 45-sample micro-batch as hipGraph replays, SEGMENTED (one graph
            # per all-reduce bucket -- what a process group needs) AND with the PIPELINED teacher (batch i + 1's teacher
            # phase beside batch i's student / target / backward work).  This is what world > 1 runs when grad_accum > 1.
            if timed_graph is not None:
                n45 = max(3, n_opt)
                try:
                    dt45 = timed_graph(True, True, z=z45, prompt=P45, draws=kw45, steps=n45, warm=2)
                    out["grad_accum_5_fused_pipelined"] = {
                        "value": round(n45 / dt45, 4), "unit": "optimizer steps/s", "optimizer_steps": n45,
                        "samples_per_s": round(world * Bf * n45 / dt45, 3), "ms_per_optimizer_step": round(dt45 / n45 * 1e3, 3),
                        "global_batch": Bf * world, "micro_batch_per_gpu": Bf,
                        "launch": "segmented hipGraph replays (bucket all-reduce between them) + pipelined teacher graph on its own stream",
                        "frac_end_to_end": round(GF_DISTILL_PER_SAMPLE * 1e9 * Bf / (dt45 / n45) / 1e12 / PEAK_BF16_TFLOPS, 4),
                        "teacher_stream_placement_ms": placements[-1] if placements else None}
                except _LegSkipped as exc:
                    out["grad_accum_5_fused_pipelined"] = {"skipped": str(exc)[:300]}
            # back to the per-GPU batch of 9 for the profiled step below (the handles keep their larger arenas)
            m.train_step(z0, P, opt, sched)
        else:
            out["grad_accum_5_fused"] = {"skipped": why or "input set-up failed on another rank"}
        del z45, P45, kw45
    assert all(v == v for v in losses), "NaN distillation loss"
    # one more step with the in-library launch profiler on rank 0.  EVERY rank takes the step: at world > 1 it issues
    # the gradient all-reduces, and a collective entered by rank 0 alone would pair up with the other ranks' next
    # barrier and hang the job
    import ctypes
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    L_ = N.lib()
    if rank == 0:
        L_.ctta_prof_enable(1)
    # the profiled step keeps every launch on ONE stream: with the student forward on its side stream, or the weight-gradient
    # jobs on the handle's side stream (round 5 left that one on: kernel_ms_per_step 84.8 > ms_per_step 75.4), the bracketed
    # launch times of concurrent kernels overlap and their sum is not a duration any more
    two_stream = os.environ.get("CTTA_TWO_STREAM")
    os.environ["CTTA_TWO_STREAM"] = "0"
    wg_stream = N.get_option("wgrad_stream")
    N.set_option("wgrad_stream", 0)
    torch.cuda.synchronize()
    ev[0].record()
    m.train_step(z0, P, opt, sched)
    ev[1].record()
    torch.cuda.synchronize()
    N.set_option("wgrad_stream", wg_stream)
    if two_stream is None:
        del os.environ["CTTA_TWO_STREAM"]
    else:
        os.environ["CTTA_TWO_STREAM"] = two_stream
    if rank == 0:
        L_.ctta_prof_enable(0)
        ms, fl, cnt = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        csv = (args.profile_csv + ".distill").encode() if args.profile_csv else None
        # ALL MFMA launches (kind -1): the 4.2 TF per sample include the attention contractions, which run in the flash
        # attention forward / backward kernels, not in conv_gemm
        N.check(L_.ctta_prof_collect(-1, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(cnt), csv))
        algo = GF_DISTILL_PER_SAMPLE * 1e9 * B
        wall_ms = ev[0].elapsed_time(ev[1])
        assert ms.value <= wall_ms, ("the bracketed MFMA launches of the single-stream step sum to %.2f ms, more than the step's own "
                                     "%.2f ms: some launches still overlap" % (ms.value, wall_ms))
        out["roofline"] = {
            "kernel": "all MFMA kernels of the step: conv_gemm_kernel (forward, data-gradient and weight-gradient GEMMs) + "
                      "flash attention forward / backward",
            "bound": "mfma", "achieved": round(algo / (ms.value * 1e-3) / 1e12, 2), "peak": PEAK_BF16_TFLOPS,
            "unit": "TFLOP/s", "frac": round(algo / (ms.value * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
            "traffic": None, "traffic_detail": pmc_traffic_distill(B),   # reads unavailable (the FETCH_SIZE pass hangs)
            "algorithmic_gflop_per_sample": GF_DISTILL_PER_SAMPLE, "launches_per_step": int(cnt.value),
            "kernel_ms_per_step": round(ms.value, 3),
            "profiled_step_ms": round(wall_ms, 3),
            "profiled_step": "one eager step with every launch on ONE stream (no student side stream, no weight-gradient side stream)",
            "executed_tflops_incl_padding": round(fl.value / (ms.value * 1e-3) / 1e12, 2),
            "share_of_step_time": round(ms.value / wall_ms, 3),
            # algorithmic FLOPs of the step / the HEADLINE step time (every kernel, every gap, the optimizer tail)
            "frac_end_to_end": round(algo / (dt / n_steps) / 1e12 / PEAK_BF16_TFLOPS, 4),
        }
        td = out["roofline"]["traffic_detail"]
        if td and td.get("read_GB_per_step") is not None and td.get("write_GB_per_step") is not None and cnt.value:
            # the PMC families of tools/pmc_traffic.py that hold the MFMA kernels of the step, per launch like `achieved`
            out["roofline"]["traffic"] = int((td["read_GB_per_step"] + td["write_GB_per_step"]) * 1e9 / cnt.value)
            out["roofline"]["traffic_unit"] = ("bytes per MFMA launch (PMC FETCH_SIZE x2 + WRITE_SIZE of conv_gemm + attention families per step / "
                                               "launches per step) -- counter bytes read from the committed %s, not measured in this run" % td.get("source"))
    if rank == 0 and not args.no_latency:   # the HBM-bound kernel class of the step: fused training-state passes over 559 M fp32
        def timed(fn, reps=5):
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()   # the launches go to torch's current stream (N.stream_ptr()), the one these events see
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            back_to_back = e0.elapsed_time(e1) / reps
            singles = []      # one launch per event pair, device idle before it: the kernel alone, without the host's cadence
            for _ in range(reps):
                a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                a.record()
                fn()
                b_.record()
                torch.cuda.synchronize()
                singles.append(a.elapsed_time(b_))
            return back_to_back, sorted(singles)[reps // 2]
        n_tr, n_all = opt.n, opt.flat.numel()
        passes = [
            ("adamw_kernel", "read p, g, m, v; write p, m, v (fp32)", 28 * n_tr, lambda: opt.step(grad_scale=1.0)),
            ("ema2_kernel (AudioLCM.update_ema: the launch + its O(1) host checks)", "read student, 2 shadows; write 2 shadows (fp32)",
             20 * n_all, m.update_ema),
            ("zero_grad (fill)", "write g (fp32)", 4 * opt.grad.numel(), opt.zero_grad),
            # what the training step runs since round 6 (AudioLCM._optimizer_tail): the three passes above as one
            ("adamw_ema2_zero_kernel (optimizer.step + zero_grad + update_ema as ONE pass: the step's tail)",
             "read p, g, m, v, 2 shadows; write p, m, v, 2 shadows, g = 0 (fp32)", 48 * n_tr + 24 * (n_all - n_tr),
             lambda: m._optimizer_tail(opt, None, 1.0, True)),
        ]
        rows = []
        for name, what, nbytes, fn in passes:
            ms_, ms_one = timed(fn)
            gbps = nbytes / (ms_ * 1e-3) / 1e9
            rows.append({"kernel": name, "streams": what, "algorithmic_GB_per_launch": round(nbytes / 1e9, 3),
                         "ms": round(ms_, 3), "achieved_GBps": round(gbps, 1), "frac": round(gbps / PEAK_HBM_GBPS, 4),
                         "ms_single_launch_median": round(ms_one, 3),
                         "frac_single_launch": round(nbytes / (ms_one * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4)})
        out["hbm_kernels"] = {"bound": "hbm", "peak": PEAK_HBM_GBPS, "unit": "GB/s", "parameters": int(n_all),
                              "passes": rows,
                              "note": "ms = five launches back to back between one event pair (includes the host's per-call work "
                                      "when it exceeds the kernel); ms_single_launch_median = one launch per event pair"}
    if rank == 0 and not args.no_latency:   # the adjacent front half of the real training step (train_utils.py:155-162): wav -> log-mel -> latent
        from consistencytta_amd import audio, modules
        stft = audio.TacotronSTFT(1024, 160, 1024, 64, 16000, 0, 8000).to(dev)
        vae = modules.AutoencoderKL(ddconfig=spec.VAE_DDCONFIG, embed_dim=8, scale_factor=0.9227914214134216)
        vae.to(dev)
        vae.init_random_(seed=12)
        vae.eval().requires_grad_(False)
        wav = (torch.rand(B, 163840, generator=torch.Generator().manual_seed(9)) * 2 - 1).to(dev) * 0.5
        for _ in range(2):
            mel, _ = audio.wav_to_fbank(wav, 1024, stft)
            z = vae.get_first_stage_encoding(vae.encode_first_stage(mel.unsqueeze(1)))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            mel, _ = audio.wav_to_fbank(wav, 1024, stft)
            z = vae.get_first_stage_encoding(vae.encode_first_stage(mel.unsqueeze(1)))
        torch.cuda.synchronize()
        assert bool(torch.isfinite(z).all()) and tuple(z.shape) == (B, 8, 256, 16)
        out["wav_to_latent_ms"] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
        del vae, stft
    del m, opt
    return out


def _profile(stem):
    """Newest committed profiles/<stem>_rNN.json (PMC summaries are named per round)."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", stem + "_r[0-9][0-9].json")))
    return found[-1] if found else os.path.join(ROOT, "profiles", stem + "_r01.json")


def pmc_traffic_distill(batch):
    """Same for the distillation leg: profiles/pmc_traffic_distill_r02.json (conv_gemm family, GB per step at batch 9)."""
    try:
        d = json.load(open(_profile("pmc_traffic_distill")))
        if "batch 9" not in d.get("unit", "") or batch != 9:
            return None
        fams = [d["families"][k] for k in ("conv_gemm_kernel", "attention_kernel", "attn_bwd") if k in d["families"]]
        rd = [f["read_GB"] for f in fams]
        f = {"read_GB": None if any(v is None for v in rd) else round(sum(rd), 3),
             "write_GB": round(sum(f["write_GB"] or 0.0 for f in fams), 3)}
        return {"read_GB_per_step": f["read_GB"], "write_GB_per_step": f["write_GB"],   # read: None when that pass hung
                "source": "profiles/" + os.path.basename(_profile("pmc_traffic_distill"))}
    except Exception:
        return None


def pmc_traffic(batch):
    """HBM-side bytes per step of the conv_gemm family (GB): PMC counters cannot be collected inside the timed run, so
    this is the committed result of the two separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes over this
    same command (tools/pmc_traffic.py; units and the gfx950 x2 read correction per MI355X_MICROARCH.md).  None when
    the file is absent or was taken at another batch size."""
    try:
        d = json.load(open(_profile("pmc_traffic")))
        if "batch 32" not in d.get("unit", "") or batch != 32:
            return None
        f = d["families"]["conv_gemm_kernel"]
        return {"read_GB_per_step": f["read_GB"], "write_GB_per_step": f["write_GB"], "source": "profiles/" + os.path.basename(_profile("pmc_traffic"))}
    except Exception:
        return None


def host_cores():
    """Cores this process may really use: affinity mask, capped by the cgroup CPU quota and at 64
    (B=1 convolutions stop scaling long before that; oversubscribed OpenMP teams crawl)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


def cpu_baseline(pipe, vae, enc, mask, noise):
    """The CPU oracle on this box's host cores: B=1, fp32, the easy_inference recipe
    (1 U-Net query at t=999, w=4, VAE decode, HiFi-GAN), same weights as the GPU leg."""
    import torch
    from consistencytta_amd import spec
    from oracle import heun as oheun
    from oracle import nets as onets

    threads = host_cores()
    torch.set_num_threads(threads)
    usd = {k: v.detach().float().cpu() for k, v in pipe.unet.state_dict().items()}
    vsd = {k: v.detach().float().cpu() for k, v in vae.state_dict().items()}
    _, sig = oheun.set_timesteps(18)
    sigma = torch.tensor([float(sig[0])])
    e, m, nz = enc[:1].cpu(), mask[:1].cpu(), noise[:1].cpu()

    def clip():
        with torch.no_grad():
            z = oheun.scale_model_input(nz * sigma, sigma)
            lat = onets.unet_forward(spec.LIGHT_UNET_CONFIG, usd, z, 999.0, 4.0, e, m)
            mel = onets.vae_decode(spec.VAE_DDCONFIG, vsd, lat, float(vae.scale_factor))
            return onets.mel_to_waveform(spec.HIFIGAN_16K_64, vsd, mel)[2]

    clip()  # warm-up (first touch of the fp32 weights, oneDNN primitive caches)
    n, t0 = 0, time.perf_counter()
    while True:     # bounded sample: about 12 s of CPU work, 3..12 clips
        clip()
        n += 1
        dt = time.perf_counter() - t0
        if n >= 12 or (n >= 3 and dt >= 12.0) or dt >= 40.0:
            break
    return {"value": round(n / dt, 4), "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d clips at B=1 (config 1: 1-step, w=4, L=%d), fp32 PyTorch-CPU oracle, %.1f s"
                      % (n, e.shape[1], dt)}


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- images/s of 256 px, 35-step CFG sampling on the 100M-parameter denoiser (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, or plain
     `python bench.py --gpus N ...`, which re-executes itself under torch.distributed.run on 127.0.0.1)

One "step" = one full pass of the hot path over one batch: DiffusionGenerator.generate_latents of
64 images per GPU (32x32x4 latents, n_iter = 35, class_guidance = 6, DPM-Solver++(2M)): 35 CFG-doubled
denoiser forwards of batch 128 + on-device CFG/solver updates (+ one all-gather of the final latents
when N > 1).  Inputs (weights, initial noise, text embeddings) are resident in HBM before the timed
region.  Weak scaling: 64 images per GPU.  Synthetic, seeded data: there is no network for the
published checkpoint, so weights are the deterministic synthetic 101 M-parameter fill.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      the dominant kernel class (by HIP-event time inside the timed region) against the
                2.5 PFLOP/s dense bf16 MFMA peak
  cpu_baseline  the same module graph on the host cores, timed in this run on a bounded sample of the workload
                (a full 35-step CFG sample of a small batch): the pure-PyTorch restatement (oracle/torch_ref.py:
                the ATen CPU kernels the reference itself would run) as the headline value, and the fp32 C
                restatement (oracle/tld_oracle.c) as a second line -- reported baselines, never part of the
                measured path
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import numpy as np
import torch

PER_GPU_IMAGES = 64
N_ITER = 35
CFG = 6.0
GFLOP_PER_SAMPLE_FWD = 46.163      # SURVEY.md Appendix B (N=256, d=768, L=12), reference op count
GFLOP_BY_IMAGE_SIZE = {32: 46.163, 64: 213.466, 128: 1317.543}   # Appendix B, per sample-forward
MFMA_PEAK_TFLOPS = 2500.0          # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md


def class_flops(cls, M, d, ntok, n_layers, fused_attention=False):
    """Algorithmic flops of one launch of a profiled kernel class (average over the launches of a forward: block 0's
    QKV GEMM and attention run on the un-doubled batch when the CFG halves share it).  fused_attention: the 256-token engine runs the
    self-attention inside the QKV GEMM's epilogue (one kernel per layer), so that class carries both flop counts."""
    att = 4.0 * M * ntok * d * (n_layers - 0.5) / n_layers                  # QK^T and PV: 2 x 2 M ntok d
    if cls == "attention":
        return att
    n, k = {"gemm_qkv": (3 * d, d), "gemm_up": (4 * d, d), "gemm_down": (d, 4 * d)}[cls]
    f = 2.0 * M * n * k
    return f * (n_layers - 0.5) / n_layers + (att if fused_attention else 0.0) if cls == "gemm_qkv" else f


def pmc_traffic(cls, image_size=32, gemm_dtype="bf16"):
    """HBM bytes per launch of a GEMM / attention class from the committed PMC passes (tools/pmc_traffic.sh: separate
    FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE x 2 gfx950 correction) OF THIS WORKLOAD: profiles/rNN_pmc_traffic.json is the C1
    line (256 px, bf16), rNN_pmc_traffic_c3.json the 512 px line, rNN_pmc_traffic_c4_bf16.json / _c4_fp8.json the 1024 px lines.
    None if no pass of the workload is committed."""
    suffix = {(32, "bf16"): "", (64, "bf16"): "_c3", (128, "bf16"): "_c4_bf16", (128, "fp8"): "_c4_fp8"}.get((image_size, gemm_dtype))
    if suffix is None:
        return None, None
    path = next((q for q in (os.path.join(REPO, "profiles", f"r{r:02d}_pmc_traffic{suffix}.json") for r in (6, 5, 4, 3, 2, 1)) if os.path.exists(q)), None)
    if path is None:
        return None, None
    # template arguments <BN, EPI, F8, ...> of gemm256p_kernel: EPI 7 / 5 / 1 = the QKV forms, 6 / 4 / 2 = the up-projection forms, 3 = residual add
    epis = {"gemm_qkv": (", 7>", ", 7,", ", 1>", ", 5>", ", 1,", ", 5,"), "gemm_up": (", 6>", ", 4>", ", 2>", ", 6,", ", 4,", ", 2,", ", 8,", ", 8>"),
            "gemm_down": (", 3>", ", 3,"), "attention": ("attn",)}[cls]       # 4 / 6 / 8 = up-projection fused with dwconv + GELU
    best = None
    for name, v in json.load(open(path)).items():
        if (("gemm256p_kernel" in name and cls != "attention") or (cls == "attention" and "attn" in name and "gemm256p" not in name)) and any(e in name for e in epis):
            if best is None or v["hbm_bytes_per_launch"] * v["launches"] > best["hbm_bytes_per_launch"] * best["launches"]:
                best = v
    if best is None:
        return None, None
    return best["hbm_bytes_per_launch"], f"profiles/{os.path.basename(path)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, same workload)"


def model_flops(cfg, ntok, share_l0=True):
    """(reference op count, executed by the engine) in FLOP per sample-forward.  The reference count is SURVEY.md
    Appendix B's; the engine does not execute the cross-attention Q GEMM (folded into per-row vectors) and computes
    block 0 up to its self-attention once per CFG pair."""
    d, L, hid = cfg.embed_dim, cfg.n_layers, cfg.mlp_multiplier * cfg.embed_dim
    qkv = 2.0 * ntok * d * 3 * d
    attn = 4.0 * ntok * ntok * d
    crossq = 2.0 * ntok * d * d
    mlp = 2 * 2.0 * ntok * d * hid
    dw = 2.0 * 9 * ntok * hid
    per_layer_exec = qkv + attn + mlp + dw
    executed = L * per_layer_exec - (0.5 * (qkv + attn) if share_l0 else 0.0)
    return L * (per_layer_exec + crossq), executed


def physical_cores():
    """Distinct (package, core) pairs among the CPUs this process may run on (SMT siblings counted once)."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
        seen = set()
        for c in cpus:
            base = f"/sys/devices/system/cpu/cpu{c}/topology/"
            with open(base + "physical_package_id") as f1, open(base + "core_id") as f2:
                seen.add((f1.read().strip(), f2.read().strip()))
        return max(1, len(seen))
    except OSError:
        return max(1, (os.cpu_count() or 2) // 2)


def cpu_baseline(cfg, sd, budget_s=20.0, S=32):
    """The same workload on the host cores, HARD-BOUNDED to about `budget_s` seconds per line.

    value: the pure-PyTorch restatement of the module graph (oracle/torch_ref.py -- conv2d / linear / layer_norm /
    gelu / scaled_dot_product_attention on ATen's CPU kernels, i.e. what the reference executes on this host) running
    one 35-step CFG sample of a batch of 8 images -- all 35 denoise steps when they fit the budget, else the first k
    steps scaled to 35 (said in `sample`).  Threads: the fastest of {all, half, quarter} of the physical cores on one
    calibration step (oversubscribing SMT siblings made a 256-thread run ~300x slower on the first try).
    c_port: the fp32 C restatement with OpenMP (oracle/tld_oracle.c), 3 steps of the same batch, scaled."""
    from dataclasses import asdict
    from oracle.oracle import OracleDenoiser, num_threads, set_num_threads
    from oracle.torch_ref import TorchRefDenoiser
    from transformer_latent_diffusion_amd import schedule
    phys = physical_cores()
    levels = schedule.noise_schedule(N_ITER, 1)
    rng = np.random.default_rng(11)
    tm = TorchRefDenoiser(asdict(cfg), sd)
    b = {32: 8, 64: 2}.get(S, 1)               # the same bounded budget at every resolution: fewer images where a forward is 5x / 29x the work
    gflop_fwd = GFLOP_BY_IMAGE_SIZE[S]
    x = torch.from_numpy(rng.standard_normal((b, 4, S, S)).astype(np.float32))
    lab = torch.from_numpy((rng.standard_normal((b, 768)) * 0.5).astype(np.float32))

    def one_step():
        t0 = time.perf_counter()
        tm.sample(x, lab, levels, CFG, True, max_forwards=1)
        return time.perf_counter() - t0

    best_t, best_n = None, None
    for n in sorted({max(1, phys), max(1, phys // 2), max(1, phys // 4)}, reverse=True):
        torch.set_num_threads(n)
        one_step()                                                                  # warm (thread pool, page-in)
        t = one_step()
        if best_t is None or t < best_t:
            best_t, best_n = t, n
        elif t > 1.1 * best_t:                                                      # fewer threads is slower: stop probing
            break
    torch.set_num_threads(best_n)
    steps = int(max(1, min(N_ITER, budget_s / best_t)))
    t0 = time.perf_counter()
    out = tm.sample(x, lab, levels, CFG, True, max_forwards=None if steps == N_ITER else steps)
    dt = time.perf_counter() - t0
    assert torch.isfinite(out).all()
    per_step = dt / steps
    what = (f"one full {N_ITER}-step CFG-{CFG:g} DPM-Solver++(2M) sample" if steps == N_ITER
            else f"the first {steps} of {N_ITER} CFG denoise steps (scaled to {N_ITER})")
    res = {
        "value": b / (per_step * N_ITER), "unit": "images/s", "cores": best_n, "kind": "port",
        "sample": f"{what} of {b} images (forwards of batch {2 * b}) in {dt:.1f} s: pure-PyTorch fp32 restatement of the "
                  f"reference module graph on ATen CPU kernels, {best_n} threads ({phys} physical cores on the host)",
        "ms_per_denoise_step": per_step * 1e3,
        "gflops": 2 * b * gflop_fwd / per_step,
    }
    if S != 32:
        return res                                                        # (the C restatement's second line: C1 only -- it is 6x slower still)
    try:
        set_num_threads(phys)
        ora = OracleDenoiser(cfg, sd)
        c_steps = 3                                                       # second line only: a bounded slice, scaled
        xn, ln = x.numpy(), lab.numpy()
        ora.sample(xn, ln, levels[:2], CFG, True, 0.0, 0.0)               # warm (page in, pack)
        t0 = time.perf_counter()
        ora.sample(xn, ln, levels[:c_steps], CFG, True, 0.0, 0.0)
        dc = (time.perf_counter() - t0) / c_steps * N_ITER
        res["c_port"] = {"value": b / dc, "unit": "images/s", "cores": num_threads(),
                         "sample": f"{c_steps} of {N_ITER} CFG denoise steps on {b} images, scaled to {N_ITER}: fp32 C "
                                   f"restatement (oracle/tld_oracle.c) with OpenMP"}
    except Exception as exc:          # the second line is optional
        res["c_port"] = {"error": str(exc)[:200]}
    return res


class PowerPoller:
    """Socket power and shader clock of the bench device from hwmon sysfs while one step of the workload runs (outside the timed region, see main): the part holds a power cap
    by lowering its clock, so the MFMA peak the kernels can reach is the guide's 2.5 PFLOP/s scaled by sclk / 2400 MHz.
    Best effort: absent or unreadable files give no 'power' object in the JSON line."""

    def __init__(self, dev_index, period=0.25):
        import glob
        import threading
        self.files = None
        want = None
        try:
            pr = torch.cuda.get_device_properties(dev_index)
            want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}."
        except Exception:
            pass
        cands = []
        for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            f = {k: os.path.join(d, k) for k in ("power1_average", "power1_input", "freq1_input", "power1_cap") if os.path.exists(os.path.join(d, k))}
            if ("power1_average" in f or "power1_input" in f) and "freq1_input" in f:
                pci = os.path.basename(os.path.realpath(os.path.join(d, "..", "..")))
                cands.append((pci, f))
        match = [c for c in cands if want and c[0].startswith(want)]
        self.cands = match if match else cands          # (no PCI match: keep all, the busiest one is ours -- one GPU is visible to this process)
        self.rows = [[] for _ in self.cands]
        self.stop = False
        self.period = period
        self.thread = threading.Thread(target=self._run, daemon=True) if self.cands else None

    @staticmethod
    def _read(path):
        try:
            with open(path) as h:
                return int(h.read().strip())
        except Exception:
            return None

    def _run(self):
        while not self.stop:
            for i, (_, f) in enumerate(self.cands):
                p = self._read(f.get("power1_average", f.get("power1_input")))
                c = self._read(f["freq1_input"])
                if p is not None and c is not None:
                    self.rows[i].append((p / 1e6, c / 1e6))
            time.sleep(self.period)

    def start(self):
        if self.thread:
            self.thread.start()

    def finish(self):
        if not self.thread:
            return None
        self.stop = True
        self.thread.join(timeout=1.0)
        best = max(range(len(self.cands)), key=lambda i: sum(r[0] for r in self.rows[i]) / max(len(self.rows[i]), 1))
        rows = self.rows[best]
        if len(rows) < 2:
            return None
        pw = sorted(r[0] for r in rows)
        ck = sorted(r[1] for r in rows)
        cap = self._read(self.cands[best][1]["power1_cap"]) if "power1_cap" in self.cands[best][1] else None
        return {"avg_w": sum(pw) / len(pw), "median_w": pw[len(pw) // 2], "cap_w": cap / 1e6 if cap else None,
                "sclk_mhz_median": ck[len(ck) // 2], "sclk_mhz_min": ck[0], "sclk_mhz_max": ck[-1], "samples": len(rows),
                "source": f"hwmon sysfs of {self.cands[best][0]}, sampled every {self.period:.2f} s over ~1 s of extra (untimed) steps of the same workload"}


def respawn_under_torchrun(n):
    """`python bench.py --gpus N` (N > 1) without a launcher: re-execute under torch.distributed.run, one rank per GPU."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush(); sys.stderr.flush()
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--images-per-gpu", type=int, default=PER_GPU_IMAGES)
    ap.add_argument("--image-size", type=int, default=32, choices=(32, 64, 128),
                    help="latent size: 32 = C1/C2 (headline), 64 = C3 shape, 128 = C4 shape (bf16)")
    ap.add_argument("--gemm-dtype", default="bf16", choices=("bf16", "fp8"),
                    help="operand type of the QKV / MLP GEMMs: bf16 (headline) or fp8 (MX-fp8, BASELINE config C4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--with-vae", action="store_true",
                    help="after the timed region, also decode each rank's latents with the native VAE (SDXL-VAE geometry, random-init "
                         "weights) and report it beside the headline line ('with_vae'); never part of 'value'")
    ap.add_argument("--no-profile", action="store_true", help="skip HIP-event timing of the GEMM classes")
    ap.add_argument("--extra-classes", default="", help="comma-separated non-MFMA kernel classes (cross_row, dwconv_gelu, layernorm, embed, tail, update) to time on the "
                                                        "untimed profiling pass as well; reported under 'other_classes' (A/B tooling)")
    ap.add_argument("--no-power", action="store_true", help="do not poll hwmon power / clock during the timed region")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for plumbing tests)")
    ap.add_argument("--same-device", action="store_true",
                    help="plumbing test on a 1-GPU box: every rank uses cuda:0 (needs --backend gloo)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        if "RANK" in os.environ or world != 1:
            raise SystemExit(f"--gpus {args.gpus} but the launcher set WORLD_SIZE={world}")
        respawn_under_torchrun(args.gpus)          # does not return
    assert torch.cuda.is_available(), "bench.py measures the HIP engine; no HIP device is visible"
    dev = torch.device("cuda", 0 if args.same_device else local_rank)
    torch.cuda.set_device(dev)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from dataclasses import asdict
    from transformer_latent_diffusion_amd import Denoiser, DiffusionGenerator, config_100m
    from transformer_latent_diffusion_amd.sharded import generate_latents_sharded
    from transformer_latent_diffusion_amd.weights import synth_state_dict

    cfg = config_100m(args.image_size)
    S = args.image_size
    gflop_fwd = GFLOP_BY_IMAGE_SIZE[S]
    ntok = (S // 2) ** 2
    sd = synth_state_dict(cfg, 5)
    model = Denoiser(**asdict(cfg)).to(dev).set_gemm_dtype(args.gemm_dtype)
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    B = args.images_per_gpu
    model.reserve(2 * B)
    gen = DiffusionGenerator(model, None, dev, torch.float32)

    total = B * world
    g = torch.Generator().manual_seed(11)
    x_T = torch.randn(total, 4, S, S, generator=g).to(dev)                # resident before timing
    labels = (torch.randn(total, 768, generator=torch.Generator().manual_seed(12)) * 0.5).to(dev)

    def one_step():
        return generate_latents_sharded(gen, labels, n_iter=N_ITER, num_imgs=total, class_guidance=CFG,
                                        img_size=S, sharp_f=0.0, bright_f=0.0, exponent=1, seeds=x_T)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    gemm_classes = ("gemm_qkv", "gemm_up", "gemm_down", "attention")
    extra_classes = tuple(c for c in args.extra_classes.split(",") if c)
    for _ in range(args.warmup):
        out = one_step()
    fence()
    # choose the dominant GEMM class on one untimed pass (all three classes under HIP events), then keep only
    # that class's events inside the timed region (each event pair costs ~1 us of stream time)
    warm_prof = {}
    dom_cls = "gemm_down"
    if not args.no_profile:
        model.set_profile(gemm_classes + extra_classes)
        out = one_step()
        fence()
        other_prof = {c: model.get_profile(c) for c in extra_classes}
        warm_prof = {c: model.get_profile(c) for c in gemm_classes}
        warm_prof = {c: v for c, v in warm_prof.items() if v[1] > 0}       # (no attention launches: it runs inside the QKV kernel)
        dom_cls = max(warm_prof, key=lambda c: warm_prof[c][0])
        model.set_profile((dom_cls,))
        model.reserve_profile(dom_cls, warm_prof[dom_cls][1] * args.steps)    # no hipEventCreate inside the timed region
    fence()
    # per-step boundaries as events on the launch stream (no synchronisation inside the timed region): the median step time
    #  is reported beside the contract's wall-clock mean
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        marks[i].record()
        out = one_step()
    marks[args.steps].record()
    fence()
    dt = time.perf_counter() - t0
    # socket power and shader clock of the same workload, on a few MORE steps after the timed region: a hwmon read stalls the device for milliseconds (four reads per
    # second inside the timed region cost 2.5 % of the images/s, a hundred 4-7 %, and at 50 reads per second the reading itself drops from 1350 to 1135 W: same-box
    # A/B in profiles/r06_power_probe.txt), so the timed steps run unobserved and the observed ones are sampled five times a second
    power = None
    if rank == 0 and world == 1 and not args.no_power:       # (one rank only: a step of the sharded path ends in a collective)
        poller = PowerPoller(dev.index, period=0.2)
        poller.start()
        t_obs = time.perf_counter()
        while time.perf_counter() - t_obs < 0.9:
            one_step()
            torch.cuda.synchronize(dev)
        power = poller.finish()
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    assert torch.isfinite(out).all()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    vae_info = None
    if args.with_vae:
        from transformer_latent_diffusion_amd.vae import AutoencoderKLDecoder, VaeDecoderConfig
        per_sample = (8 * S) ** 2 * 256 * 2                     # largest activation of the decoder, bytes per sample
        vae = AutoencoderKLDecoder(VaeDecoderConfig(), max_batch=max(1, min(B, int(3.9 * 2 ** 30 // per_sample)))).to(dev)
        lat = (out[rank * B:(rank + 1) * B] * 8).float()        # scale_factor 8 (tld/diffusion.py:91,180)
        img = vae.decode(lat)[0]
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            img = vae.decode(lat)[0]
        fence()
        vdt = (time.perf_counter() - t1) / args.steps
        assert torch.isfinite(img).all()
        vae_cpu = None
        if world == 1 and not args.no_cpu_baseline:             # cpu_baseline leg of the VAE stage: fp32 torch restatement, one image
            from oracle.vae_ref import TorchRefVaeDecoder
            ref = TorchRefVaeDecoder(VaeDecoderConfig(), vae.state_dict())
            zc = lat[:1].cpu()
            ref.decode(zc)
            t2 = time.perf_counter()
            ref.decode(zc)
            vae_cpu = {"value": 1.0 / (time.perf_counter() - t2), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
                       "sample": "1 image, fp32 torch restatement of AutoencoderKL.decode (oracle/vae_ref.py)"}
        vae_info = {"vae_ms_per_batch": vdt * 1e3, "vae_images_per_sec_per_gpu": B / vdt, "vae_cpu_baseline": vae_cpu,
                    "end_to_end_images_per_sec": total / (dt / args.steps + vdt),
                    "note": "denoise + native VAE decode (tools/vae_bench.py, DESIGN.md 7.1) back to back on each rank's shard; "
                            "SDXL-VAE geometry with random-init weights (no checkpoint offline); not part of 'value'"}

    prof = {}
    if not args.no_profile:
        prof = dict(warm_prof)                       # untimed pass: all classes (context for the JSON)
        prof[dom_cls] = model.get_profile(dom_cls)   # timed region: the dominant class, live
        model.set_profile(())

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = total * args.steps / dt
        line = {
            "metric": f"images/sec ({8 * S}px, 35-step CFG sampling), denoiser only",
            "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.gemm_dtype == "bf16" else "fp8-gemm/bf16", "data": "synthetic",
            "config": {"workload": f"{ {32: 'C1', 64: 'C3', 128: 'C4'}[S]}{' (MX-fp8 QKV/MLP GEMMs)' if args.gemm_dtype == 'fp8' else ''}: 100M-param denoiser (d=768, L=12), {S}x{S}x4 latents, 35 steps + CFG 6, "
                                   f"DPM-Solver++(2M), {B} images/GPU (model batch {2 * B})",
                       "images_per_gpu": B, "global_batch": total, "n_iter": N_ITER, "class_guidance": CFG,
                       "parallelism": f"dp{world} (sample-sharded, one all-gather)" if world > 1 else "single GPU"},
            "ms_per_denoise_step": ms_per_step / N_ITER,
            "ms_per_step_median": median_ms, "ms_per_step_min": step_ms[0], "ms_per_step_max": step_ms[-1],
            "value_at_median_step": total / (median_ms * 1e-3),
        }
        # whole-step MFMA fractions.  "algorithmic" = the REFERENCE's op count per image / time:
        # the fraction of the 774 img/s ceiling.  "executed" = the flops the engine's kernels actually perform (no
        # cross-attention Q GEMM, block 0 computed once per CFG pair up to its attention): the hardware utilisation.
        ref_f, exe_f = model_flops(cfg, ntok)
        line["algorithmic_tflops_reference_op_count"] = value * 2 * N_ITER * gflop_fwd / 1e3
        line["frac_of_bf16_mfma_peak_algorithmic"] = value * 2 * N_ITER * gflop_fwd / 1e3 / (MFMA_PEAK_TFLOPS * world)
        line["executed_tflops"] = value * 2 * N_ITER * exe_f / 1e12
        line["frac_of_bf16_mfma_peak_executed"] = value * 2 * N_ITER * exe_f / 1e12 / (MFMA_PEAK_TFLOPS * world)
        if prof:
            M = 2 * B * ntok
            dom = dom_cls
            ms, n = prof[dom]
            avg_s = ms / max(n, 1) / 1e3
            fused_att = "attention" not in prof
            fl = lambda c: class_flops(c, M, cfg.embed_dim, ntok, cfg.n_layers, fused_att)
            ach = fl(dom) / avg_s / 1e12
            tot_f = sum(fl(c) * prof[c][1] for c in prof)
            tot_t = sum(prof[c][0] for c in prof) / 1e3
            traffic, traffic_src = pmc_traffic(dom, S, args.gemm_dtype)
            # dense MFMA peak of the dominant class's operand type (MI355X_MICROARCH.md): bf16 2.5 PF, MX-fp8 5 PF
            peak = 2 * MFMA_PEAK_TFLOPS if (args.gemm_dtype == "fp8" and dom != "attention") else MFMA_PEAK_TFLOPS
            line["roofline"] = {
                "bound": "mfma", "kernel": "attn kernel" if dom == "attention" else f"gemm256p_kernel<{dom}>", "achieved": ach,
                "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_ms": ms / max(n, 1), "launches": n, "flops_per_launch": fl(dom),
                "all_mfma_classes": {c: {"avg_ms": prof[c][0] / max(prof[c][1], 1), "launches": prof[c][1],
                                         "tflops": fl(c) / (prof[c][0] / max(prof[c][1], 1) / 1e3) / 1e12} for c in prof},
                "note": "dominant class timed with HIP events inside the timed region; the other classes on one untimed pass; "
                        "flops_per_launch is the average over a forward's launches (block 0 runs on the un-doubled batch)"
                        + ("; gemm_qkv = QKV projection + the whole self-attention in one kernel per layer (EPI_QKV_ATTN): its flops are the sum" if fused_att else ""),
                "mfma_aggregate_tflops": tot_f / tot_t / 1e12,
            }
            if power and power.get("sclk_mhz_median"):
                # the same kernel against the MFMA rate of the clock the part actually held under its power cap (peak is quoted at 2400 MHz)
                line["roofline"]["peak_at_sustained_clock"] = peak * power["sclk_mhz_median"] / 2400.0
                line["roofline"]["frac_at_sustained_clock"] = ach / (peak * power["sclk_mhz_median"] / 2400.0)
        if power:
            line["power"] = power
        if prof and extra_classes:
            line["other_classes"] = {c: {"avg_ms": v[0] / max(v[1], 1), "launches": v[1]} for c, v in other_prof.items()}
        if vae_info:
            line["with_vae"] = vae_info
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, sd, S=S)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

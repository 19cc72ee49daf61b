"""Training step of the denoiser on the gfx950 engine -- the host side of ``tld_train_*``.

Mirrors the body of the reference's training loop (tld/train.py:118-175) for one batch:

    noise_level ~ Beta(beta_a, beta_b);  x_noisy = noise_level * noise + (1 - noise_level) * x       (:120-130)  -> make_batch / mix_noise
    label[rand < 0.15] = 0                                                                           (:135-138)  -> make_batch / drop_labels
    pred = model(x_noisy, noise_level.view(-1, 1), label); loss = MSELoss(pred, x); backward         (:160-168)  -> Trainer.forward_backward
    gradient all-reduce of accelerate's DDP wrapper                                                  (:114,168)  -> Trainer.optimizer_step (RCCL)
    optimizer.step()  (Adam, lr) ; update_ema(ema_model, model, alpha)                               (:87,169-172,55-58) -> Trainer.optimizer_step

Parameters, gradients, Adam moments and the EMA copy are flat fp32 torch tensors on the device (PyTorch owns the memory and the
collective; the arithmetic is in libtld_hip.so).  There is no CPU path: constructing a ``Trainer`` without a HIP device raises.
"""
from __future__ import annotations

import ctypes as C
import os
import warnings
from collections import OrderedDict
from dataclasses import asdict, dataclass
from typing import Dict, Mapping, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .configs import DenoiserConfig
from .weights import state_dict_spec, synth_state_dict

LABEL_DROPOUT = 0.15                 # tld/train.py:135


@dataclass
class TrainConfig:
    """tld/configs.py:58-72 (the fields the step itself reads: lr, alpha, beta_a, beta_b, batch_size)."""
    batch_size: int = 128
    lr: float = 3e-4
    n_epoch: int = 100
    alpha: float = 0.999
    from_scratch: bool = True
    beta_a: float = 1
    beta_b: float = 2.5
    save_and_eval_every_iters: int = 1000
    run_id: str = ""
    model_name: str = ""
    compile: bool = True
    save_model: bool = True
    use_wandb: bool = True


def param_layout(cfg) -> "OrderedDict[str, Tuple[int, Tuple[int, ...]]]":
    """{key: (offset, shape)} of the flat parameter vector: ``Denoiser.named_parameters()`` order = the state_dict order without the two
    registered buffers (``angular_speeds``, ``precomputed_pos_enc``).  The engine reports the same table (``tld_train_param_layout``)."""
    out: "OrderedDict[str, Tuple[int, Tuple[int, ...]]]" = OrderedDict()
    off = 0
    for k, (shape, kind) in state_dict_spec(cfg).items():
        if kind in ("angular", "arange"):
            continue
        out[k] = (off, tuple(shape))
        off += int(np.prod(shape))
    return out


def mix_noise(x: torch.Tensor, noise_level: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """tld/train.py:123-130: ``noise_level`` is float64 there (np.random.beta), so the mix is formed in float64 and cast to float32."""
    nl = noise_level.to(torch.float64).view(-1, 1, 1, 1)
    return (nl * noise + (1 - nl) * x).float()


def drop_labels(y: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """tld/train.py:135-138 (on a copy: the reference mutates the batch tensor the loader yields)."""
    out = y.clone()
    out[mask] = 0
    return out


def allreduce_mean_(flat: torch.Tensor, group=None) -> float:
    """DDP's gradient averaging as one collective on the flat vector: SUM all-reduce in place; returns the factor (1 / world size)
    the optimizer kernel applies (tld/train.py:114,168: accelerate's DDP wrapper).  No-op without an initialised process group."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        return 1.0 / dist.get_world_size(group)
    return 1.0


def update_ema_(ema: torch.Tensor, params: torch.Tensor, alpha: float = 0.999) -> None:
    """tld/train.py:55-58 on flat vectors (host-side statement of what the fused kernel does; used by tests)."""
    ema.mul_(alpha).add_(params, alpha=1 - alpha)


class Trainer:
    """One model replica + optimizer state on one device.  With ``torch.distributed`` initialised (backend ``nccl`` = RCCL) every rank
    holds a replica, gradients are summed with one all-reduce of the flat vector per step and divided by the world size."""

    def __init__(self, denoiser_cfg: DenoiserConfig, train_cfg: Optional[TrainConfig] = None, device="cuda",
                 state_dict: Optional[Mapping[str, torch.Tensor]] = None, init_seed: int = 0, max_batch: Optional[int] = None,
                 betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8, keep_ema: bool = True, process_group=None,
                 overlap_allreduce: bool = True, use_graph: Optional[bool] = None):
        self.cfg = denoiser_cfg
        self.tc = train_cfg if train_cfg is not None else TrainConfig()
        dev = torch.device(device)
        if dev.type != "cuda" or not torch.cuda.is_available():
            raise RuntimeError("Trainer needs a HIP device ('cuda'); the training engine has no CPU path")
        # ONE device for the engine, the flat tensors and the stream: "cuda" means the caller's CURRENT device (torch.cuda.set_device(local_rank)
        # under torch.distributed.run), exactly as Denoiser resolves it -- not device 0
        self.device = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        if float(denoiser_cfg.dropout) != 0.0:
            # the reference trains with model.train(): dropout_p in SDPA and nn.Dropout in the MLP (tld/transformer_blocks.py:43,105).  The engine's
            # training forward has no dropout; training a different model silently is worse than refusing (every published config has dropout = 0)
            raise NotImplementedError(f"DenoiserConfig.dropout = {denoiser_cfg.dropout}: the training engine implements dropout = 0 only")
        self.betas, self.eps = betas, eps
        self.group = process_group
        self.layout = param_layout(denoiser_cfg)
        self.numel = sum(int(np.prod(s)) for _, s in self.layout.values())
        c = asdict(denoiser_cfg)
        L = _lib.lib()
        self.max_batch = int(max_batch if max_batch is not None else self.tc.batch_size)
        cc = _lib.TldConfig(c["image_size"], c["noise_embed_dims"], c["patch_size"], c["embed_dim"], c["n_layers"], c["text_emb_size"],
                            c["n_channels"], c["mlp_multiplier"], self.max_batch, self.device.index)
        h = C.c_void_p()
        _lib.check(L.tld_train_create(C.byref(cc), C.byref(h)), "tld_train_create")
        self._h = h
        try:
            self._check_layout()
            z = lambda: torch.zeros(self.numel, dtype=torch.float32, device=self.device)
            self.params, self.grads, self.exp_avg, self.exp_avg_sq = z(), z(), z(), z()
            self.ema = z() if keep_ema else None
            self.step = 0
            sd = state_dict if state_dict is not None else {k: torch.from_numpy(np.array(v)) for k, v in synth_state_dict(denoiser_cfg, init_seed).items()}
            self.load_state_dict(sd)
            _lib.check(L.tld_train_bind(self._h, C.c_void_p(self.params.data_ptr()), C.c_void_p(self.grads.data_ptr())), "tld_train_bind")
        except Exception:
            L.tld_train_destroy(self._h)
            self._h = None
            raise
        self._loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        # HIP-graph replay of forward + backward: the step is ~1 900 small launches with static shapes; captured once (on the second full-batch
        # call, after an eager warm-up) and replayed from fixed input buffers, it takes the host out of the loop.  Adam stays outside (its step
        # count is a kernel argument).  Round 5: ON by default for a single replica (TLD_TRAIN_GRAPH=0 / use_graph=False: eager launches) -- on a
        # slow host the eager step's wall time was 44 ms against 31 ms of device time; with more than one rank the eager path stays the default,
        # because the per-block gradient all-reduces are launched from a host callback in the middle of the backward (not capturable).
        # precedence: the explicit constructor argument, then TLD_TRAIN_GRAPH, then the default (one replica: on).  The world size is read
        # here; a process group initialised later does not flip the choice, so a graph step with more than one rank (which gives up the
        # overlapped per-block all-reduce for full batches) is announced once, at the first such step (forward_backward).
        env = os.environ.get("TLD_TRAIN_GRAPH")
        self.use_graph = bool(use_graph) if use_graph is not None else (bool(int(env)) if env is not None else self._world() == 1)
        self._graph_world_warned = False
        self.overlap_allreduce = overlap_allreduce
        self._comm_stream = None
        self._pending = []                # async all-reduce handles of the gradient slices of the step in flight
        self._reduced = False             # True between an overlapped forward_backward and the optimizer_step that consumes its reduced gradients
        self._cb_error = None             # first exception raised inside the gradient-ready callback (ctypes would swallow it)
        self.global_step = 0              # tld/train.py:104,174 -- the loop counter the checkpoint carries; Adam's own count is self.step
        self._slices = []                 # (offset, numel) in the order they became ready (tests)
        self._graph = None
        self._graph_calls = 0
        self._static = None
        # gradient accumulation over micro-batches (accelerator.accumulate(), tld/train.py:160)
        self._acc = None                  # flat fp32 sum of the gradients of the micro-batches folded so far
        self._acc_n = 0
        self._micro_scale = 1.0           # 1 / (micro-batches of the step): applied by the optimizer kernel together with 1 / world
        # host -> device staging of train_step's batch: pinned double buffers + a copy stream, so that the host never blocks on a pageable copy
        # that is stream-ordered behind the previous step's kernels (that stall made the eager step's wall time device + enqueue time)
        self._stage = [None, None]
        self._stage_i = 0
        self._copy_stream = None

    def _check_layout(self):
        L = _lib.lib()
        n = L.tld_train_tensor_count(self._h)
        if n != len(self.layout) or L.tld_train_param_count(self._h) != self.numel:
            raise RuntimeError("parameter layout of the engine and of weights.state_dict_spec disagree")
        buf = C.create_string_buffer(256)
        off, num = C.c_int64(), C.c_int64()
        for i, (k, (o, s)) in enumerate(self.layout.items()):
            _lib.check(L.tld_train_param_layout(self._h, i, buf, 256, C.byref(off), C.byref(num)), "tld_train_param_layout")
            if buf.value.decode() != k or off.value != o or num.value != int(np.prod(s)):
                raise RuntimeError(f"parameter layout mismatch at {i}: engine {buf.value.decode()}@{off.value}+{num.value}, host {k}@{o}")

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None:
                _lib.lib().tld_train_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---- state ----------------------------------------------------------------------------------------------------------------
    def _unflatten(self, flat: torch.Tensor) -> "OrderedDict[str, torch.Tensor]":
        out = OrderedDict()
        for k, (o, s) in self.layout.items():
            out[k] = flat[o:o + int(np.prod(s))].view(*s)
        return out

    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        """Views into the live flat parameter vector, reference keys (+ the two registered buffers)."""
        sd = self._unflatten(self.params)
        sd["fourier_feats.0.angular_speeds"] = self._angular
        return sd

    def ema_state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        """What the reference saves as ``model_ema`` (tld/train.py:150-156) and the inference ``Denoiser`` loads."""
        if self.ema is None:
            raise RuntimeError("this Trainer keeps no EMA copy (keep_ema=False)")
        sd = OrderedDict((k, v.clone()) for k, v in self._unflatten(self.ema).items())
        sd["fourier_feats.0.angular_speeds"] = self._angular.clone()
        sd["denoiser_trans_block.precomputed_pos_enc"] = torch.arange(self.layout["denoiser_trans_block.pos_embed.weight"][1][0])
        return sd

    def grad_dict(self) -> "OrderedDict[str, torch.Tensor]":
        """Views into the flat gradient vector.  In overlap mode with more than one rank these are the cross-rank SUM (not the mean) once the
        slice reductions have been waited for, which this does."""
        self.wait_gradients()
        return self._unflatten(self.grads)

    def load_state_dict(self, sd: Mapping[str, torch.Tensor]) -> "Trainer":
        missing = [k for k in self.layout if k not in sd]
        if missing:
            raise RuntimeError(f"missing keys in state_dict: {missing[:4]}{' ...' if len(missing) > 4 else ''}")
        host = torch.empty(self.numel, dtype=torch.float32)
        for k, (o, s) in self.layout.items():
            t = (sd[k] if isinstance(sd[k], torch.Tensor) else torch.as_tensor(np.asarray(sd[k]))).detach().to(torch.float32).cpu()
            if tuple(t.shape) != s:
                raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(t.shape)}, model {s}")
            host[o:o + t.numel()] = t.reshape(-1)
        self.params.copy_(host)
        if self.ema is not None:
            self.ema.copy_(host)                                  # ema_model = copy.deepcopy(model)   (tld/train.py:110-111)
        ang = sd.get("fourier_feats.0.angular_speeds")
        if ang is None:
            ang = torch.from_numpy(np.array(synth_state_dict(self.cfg, 0)["fourier_feats.0.angular_speeds"]))
        self._angular = (ang if isinstance(ang, torch.Tensor) else torch.as_tensor(np.asarray(ang))).detach().to(torch.float32).cpu().contiguous()
        a = self._angular.numpy()
        _lib.check(_lib.lib().tld_train_set_angular_speeds(self._h, a.ctypes.data_as(C.POINTER(C.c_float)), a.size), "tld_train_set_angular_speeds")
        _lib.check(_lib.lib().tld_train_bind(self._h, C.c_void_p(self.params.data_ptr()), C.c_void_p(self.grads.data_ptr())), "tld_train_bind")
        if getattr(self, "_acc_n", 0):
            self.reset_accumulation()                             # gradients folded against the old weights mean nothing for the new ones
        return self

    # ---- the step -------------------------------------------------------------------------------------------------------------
    def make_batch(self, x: torch.Tensor, y: torch.Tensor, np_rng: Optional[np.random.Generator] = None,
                   generator: Optional[torch.Generator] = None):
        """tld/train.py:118-138 for one (latents, text embeddings) batch: returns (x_noisy, noise_level fp32, label)."""
        rng = np_rng if np_rng is not None else np.random.default_rng()
        noise_level = torch.tensor(rng.beta(self.tc.beta_a, self.tc.beta_b, len(x)))          # float64 (:120-122)
        noise = torch.randn(x.shape, generator=generator, dtype=x.dtype)
        mask = torch.rand(y.size(0), generator=generator) < LABEL_DROPOUT
        return mix_noise(x, noise_level, noise), noise_level.float(), drop_labels(y, mask)

    def forward_backward(self, x_noisy: torch.Tensor, noise_level: torch.Tensor, label: torch.Tensor, target: torch.Tensor, *,
                         last_micro_batch: bool = True):
        """zero_grad + forward + MSE + backward (tld/train.py:163-168).  Returns (loss [1] on the device, pred); the gradients are in
        ``self.grads`` (flat) / ``grad_dict()``.

        Gradient accumulation (``accelerator.accumulate()``, tld/train.py:160, with ``gradient_accumulation_steps`` = n > 1): call n - 1 times
        with ``last_micro_batch=False`` and once with the default; ``optimizer_step()`` then steps on the MEAN of the n micro-batch gradients
        (equal micro-batch sizes: the gradient of the mean loss).  Like DDP's ``no_sync`` the ranks exchange nothing until the last micro-batch:
        the accumulated sum is reduced by ONE all-reduce in ``optimizer_step``."""
        dev = self.device
        t = lambda a: a.detach().to(dev, torch.float32).contiguous()
        xn, nl, lab, tgt = t(x_noisy), t(noise_level).view(-1), t(label), t(target)
        B = xn.shape[0]
        if not (nl.numel() == B and lab.shape[0] == B and tgt.shape == xn.shape):
            raise ValueError("inconsistent batch shapes")
        if B > self.max_batch:
            raise ValueError(f"batch {B} exceeds max_batch {self.max_batch}")
        accumulating = (not last_micro_batch) or self._acc_n > 0
        if self.use_graph and B == self.max_batch and self._world() > 1 and self.overlap_allreduce and not self._graph_world_warned:
            self._graph_world_warned = True
            warnings.warn("Trainer: HIP-graph replay is on with more than one rank -- full batches reduce their gradients with ONE all-reduce after the "
                          "backward instead of the per-block slices under it (use_graph=False / TLD_TRAIN_GRAPH=0 restores the overlap)", RuntimeWarning)
        overlap = self.overlap_allreduce and self._world() > 1 and not (self.use_graph and B == self.max_batch) and not accumulating
        self.wait_gradients()             # a previous call's slice reductions may still be running on the communication stream: they read / write self.grads
        self._slices = []
        self._cb_error = None
        self._reduced = False             # (a previous overlapped call that was never stepped on must not vouch for THIS call's gradients)
        cb = None
        if overlap:
            import torch.distributed as dist
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=dev)
            compute = torch.cuda.current_stream(dev)

            def ready(_user, off, n):      # called by the engine, on this thread, right after the kernels finishing grads[off : off + n] are enqueued
                if self._cb_error is not None:
                    return                 # a slice already failed: the step is void, launch nothing more
                try:                       # ctypes swallows what a callback raises: keep it and re-raise when the engine call returns
                    ev = torch.cuda.Event()
                    ev.record(compute)
                    self._comm_stream.wait_event(ev)
                    with torch.cuda.stream(self._comm_stream):
                        self._pending.append(dist.all_reduce(self.grads[off:off + n], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                    self._slices.append((int(off), int(n)))
                except BaseException as e:  # noqa: BLE001 -- re-raised below
                    self._cb_error = e
            cb = _lib.GRAD_READY_FN(ready)

        def launch(xn_, nl_, lab_, tgt_, pred_, refresh=False):
            stream = torch.cuda.current_stream(dev).cuda_stream
            with torch.cuda.device(dev):
                if cb is not None:
                    _lib.check(_lib.lib().tld_train_forward_backward_cb(self._h, C.c_void_p(xn_.data_ptr()), C.c_void_p(nl_.data_ptr()), C.c_void_p(lab_.data_ptr()),
                                                                        C.c_void_p(tgt_.data_ptr()), B, C.c_void_p(self._loss.data_ptr()), C.c_void_p(pred_.data_ptr()),
                                                                        C.c_void_p(stream), cb, None), "tld_train_forward_backward_cb")
                    if self._cb_error is not None:
                        err, self._cb_error = self._cb_error, None
                        self._abandon_pending()
                        raise RuntimeError("gradient all-reduce of a slice failed during the backward; the step's gradients are not reduced") from err
                    self._reduced = True
                    return
                if refresh:     # graph capture: the bf16 / transposed operand copies are rebuilt INSIDE the captured region, whatever the engine's
                    # weights_fresh flag says at capture time -- every replay then follows the optimizer's latest parameters
                    _lib.check(_lib.lib().tld_train_refresh_weights(self._h, C.c_void_p(stream)), "tld_train_refresh_weights")
                _lib.check(_lib.lib().tld_train_forward_backward(self._h, C.c_void_p(xn_.data_ptr()), C.c_void_p(nl_.data_ptr()), C.c_void_p(lab_.data_ptr()),
                                                                 C.c_void_p(tgt_.data_ptr()), B, C.c_void_p(self._loss.data_ptr()), C.c_void_p(pred_.data_ptr()),
                                                                 C.c_void_p(stream)), "tld_train_forward_backward")
        if self.use_graph and B == self.max_batch:
            self._graph_calls += 1
            if self._graph_calls >= 2:
                if self._graph is None:      # second call: fixed buffers, capture (the first call ran eagerly: one-time attribute / cache set-up is done)
                    self._static = tuple(torch.empty_like(a) for a in (xn, nl, lab, tgt, xn))
                    for dst, src in zip(self._static[:4], (xn, nl, lab, tgt)):
                        dst.copy_(src)
                    torch.cuda.synchronize(dev)
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        launch(*self._static, refresh=True)
                    self._graph = g
                else:
                    for dst, src in zip(self._static[:4], (xn, nl, lab, tgt)):
                        dst.copy_(src)
                self._graph.replay()
                return self._fold_micro_batch(last_micro_batch), self._static[4].clone()         # (a copy: the static buffer is overwritten by the next replay)
        pred = torch.empty_like(xn)
        launch(xn, nl, lab, tgt, pred)
        return self._fold_micro_batch(last_micro_batch), pred

    def _fold_micro_batch(self, last: bool) -> torch.Tensor:
        """Gradient accumulation bookkeeping after a forward_backward; returns the loss tensor to hand out (a copy while accumulating: the engine
        overwrites its loss cell on the next micro-batch).  ``global_step`` advances HERE, once per micro-batch: the reference counts loader
        iterations (``global_step += 1`` after every pass through ``accelerator.accumulate()``, tld/train.py:162-174), so checkpoints and the
        ``save_and_eval_every_iters`` cadence keep its meaning with accumulation on.  One stated deviation remains for n > 1 micro-batches: the
        reference calls ``update_ema`` in every iteration (on unchanged weights between optimizer steps), this trainer once per optimizer step."""
        self.global_step += 1
        if last and self._acc_n == 0:
            self._micro_scale = 1.0
            return self._loss
        if self._acc is None:
            self._acc = torch.zeros_like(self.grads)
        if last:                                  # grads <- sum of all micro-batch gradients; the optimizer kernel divides by their number
            self.grads.add_(self._acc)
            self._micro_scale = 1.0 / (self._acc_n + 1)
            self._acc.zero_(); self._acc_n = 0
        else:
            self._acc.add_(self.grads)
            self._acc_n += 1
        return self._loss.clone()

    def reset_accumulation(self) -> None:
        """Abandon a gradient accumulation in progress (an exception in a later micro-batch, a skipped bad batch): the folded gradients are
        dropped and the next forward_backward starts a fresh step.  Also called when weights or a checkpoint are loaded."""
        if self._acc is not None:
            self._acc.zero_()
        self._acc_n = 0
        self._micro_scale = 1.0

    def _world(self) -> int:
        import torch.distributed as dist
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def wait_gradients(self) -> None:
        """Make the current stream wait for every gradient-slice all-reduce still in flight (overlap mode).  Until then ``self.grads`` is a mix of
        local and summed slices; after it, it is the cross-rank SUM (the optimizer kernel applies 1 / world).  Called by ``grad_dict``,
        ``optimizer_step`` and at the start of the next ``forward_backward``; a no-op when nothing is pending."""
        pending, self._pending = self._pending, []
        for w in pending:
            w.wait()

    def _abandon_pending(self) -> None:
        pending, self._pending = self._pending, []
        for w in pending:
            try:
                w.wait()
            except Exception:      # noqa: BLE001 -- the step is already being abandoned with the first error
                pass
        self._reduced = False

    def _slices_tile_vector(self) -> bool:
        """True when the recorded slices cover [0, numel) exactly once (they arrive last block first; order does not matter here)."""
        pos = 0
        for off, n in sorted(self._slices):
            if off != pos:
                return False
            pos += n
        return pos == self.numel

    def optimizer_step(self) -> None:
        """DDP gradient mean + Adam + EMA (tld/train.py:168-172).  The gradient sum over the ranks is either already in flight (per-block slices
        started during the backward, ``overlap_allreduce``) or one all-reduce of the flat vector here."""
        if self._acc_n:
            raise RuntimeError(f"optimizer_step in the middle of a gradient accumulation ({self._acc_n} micro-batches folded, none marked last); "
                               "finish it with a forward_backward(last_micro_batch=True) or drop it with reset_accumulation()")
        if self._reduced:
            self._reduced = False
            self.wait_gradients()         # the current stream waits for the slices' reductions
            if not self._slices_tile_vector():
                raise RuntimeError(f"overlapped gradient all-reduce covered {sum(n for _, n in self._slices)} of {self.numel} elements: "
                                   "refusing to step on partly reduced gradients")
            scale = 1.0 / self._world()
        else:
            scale = allreduce_mean_(self.grads, self.group)
        scale *= self._micro_scale
        self._micro_scale = 1.0
        self.step += 1
        stream = torch.cuda.current_stream(self.device).cuda_stream
        p = lambda a: C.c_void_p(a.data_ptr()) if a is not None else None
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().tld_train_adam_ema(self._h, p(self.params), p(self.grads), p(self.exp_avg), p(self.exp_avg_sq), p(self.ema), self.numel,
                                                     float(self.tc.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps), self.step,
                                                     float(self.tc.alpha), float(scale), C.c_void_p(stream)), "tld_train_adam_ema")

    def train_step(self, x: torch.Tensor, y: torch.Tensor, np_rng: Optional[np.random.Generator] = None,
                   generator: Optional[torch.Generator] = None) -> torch.Tensor:
        """One iteration of the reference's inner loop on the batch (x, y) as the loader yields it (x already divided by the VAE scale
        factor, :119).  Returns the loss tensor (device, not synchronised)."""
        x_noisy, noise_level, label = self.make_batch(x, y, np_rng, generator)
        staged = self._stage_batch((x_noisy, noise_level, label, x))
        loss, _ = self.forward_backward(*staged)
        self._release_stage()
        self.optimizer_step()
        return loss

    def _stage_batch(self, arrs):
        """Host tensors -> device through pinned double buffers on a copy stream (see __init__); device tensors pass through."""
        if not all(isinstance(a, torch.Tensor) and a.device.type == "cpu" for a in arrs):
            return arrs
        dev = self.device
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=dev)
        i = self._stage_i = self._stage_i ^ 1
        shapes = tuple(tuple(a.shape) for a in arrs)
        st = self._stage[i]
        compute = torch.cuda.current_stream(dev)
        if st is None or st["shapes"] != shapes:
            # the device buffers are allocated ON the copy stream: the caching allocator recycles blocks per stream, and a block just released by the
            # compute stream (say the previous step's prediction copy, its kernel still queued) must not come back here and be written by the copy
            # stream ahead of that kernel -- the first version of this did exactly that and trained on a corrupted batch
            with torch.cuda.stream(self._copy_stream):
                devb = [torch.empty(a.shape, dtype=torch.float32, device=dev) for a in arrs]
            for t_ in devb:
                t_.record_stream(compute)
            st = {"shapes": shapes, "host": [torch.empty(a.shape, dtype=torch.float32).pin_memory() for a in arrs], "dev": devb, "copied": None, "free": None}
            self._stage[i] = st
        if st["copied"] is not None:
            st["copied"].synchronize()            # the copy that last read these pinned buffers (two steps ago: long done)
        for h, a in zip(st["host"], arrs):
            h.copy_(a)
        if st["free"] is not None:
            self._copy_stream.wait_event(st["free"])      # the step that last read these device buffers has finished with them
        with torch.cuda.stream(self._copy_stream):
            for d, h in zip(st["dev"], st["host"]):
                d.copy_(h, non_blocking=True)
            st["copied"] = torch.cuda.Event()
            st["copied"].record(self._copy_stream)
        compute.wait_event(st["copied"])
        return tuple(st["dev"])

    def _release_stage(self):
        st = self._stage[self._stage_i]
        if st is not None:
            st["free"] = torch.cuda.Event()
            st["free"].record(torch.cuda.current_stream(self.device))

    # ---- checkpoint / resume ----------------------------------------------------------------------------------------------------
    def optimizer_state_dict(self) -> Dict[str, object]:
        """``torch.optim.Adam(model.parameters(), lr).state_dict()`` as the reference saves it (tld/train.py:86,152): per-parameter
        ``{step, exp_avg, exp_avg_sq}`` in ``named_parameters()`` order + one param group.  (Before the first step torch's ``state`` is empty.)"""
        state = {}
        if self.step > 0:
            for i, (k, (o, s)) in enumerate(self.layout.items()):
                n = int(np.prod(s))
                state[i] = {"step": torch.tensor(float(self.step)), "exp_avg": self.exp_avg[o:o + n].view(*s).clone(),
                            "exp_avg_sq": self.exp_avg_sq[o:o + n].view(*s).clone()}
        group = {"lr": self.tc.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "params": list(range(len(self.layout)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, osd: Mapping[str, object]) -> None:
        """Inverse of ``optimizer_state_dict`` (``optimizer.load_state_dict`` of tld/train.py:100): also accepts a reference checkpoint's Adam state."""
        state = osd.get("state", {})
        self.exp_avg.zero_(); self.exp_avg_sq.zero_()
        steps = set()
        for i, (k, (o, s)) in enumerate(self.layout.items()):
            st = state.get(i, state.get(str(i)))
            if st is None:
                continue
            n = int(np.prod(s))
            for name, flat in (("exp_avg", self.exp_avg), ("exp_avg_sq", self.exp_avg_sq)):
                t = torch.as_tensor(st[name]).detach().to(torch.float32)
                if tuple(t.shape) != s:
                    raise RuntimeError(f"optimizer state of parameter {i} ({k}): shape {tuple(t.shape)}, expected {s}")
                flat[o:o + n].copy_(t.reshape(-1))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise RuntimeError(f"per-parameter Adam step counts differ ({sorted(steps)}): the fused optimizer kernel keeps one count")
        self.step = steps.pop() if steps else 0
        groups = osd.get("param_groups") or []
        if groups:
            self.tc.lr = float(groups[0].get("lr", self.tc.lr))
            self.betas = tuple(groups[0].get("betas", self.betas)); self.eps = float(groups[0].get("eps", self.eps))

    def checkpoint(self) -> Dict[str, object]:
        """The dict the reference saves (tld/train.py:150-156): EMA weights, ``optimizer.state_dict()``, global step."""
        return {"model_ema": self.ema_state_dict(), "opt_state": self.optimizer_state_dict(), "global_step": self.global_step}

    def load_checkpoint(self, ckpt) -> "Trainer":
        """Resume as the reference does with ``from_scratch=False`` (tld/train.py:92-104): the EMA weights go into the live model (and the EMA
        copy restarts from them), the optimizer state and the step count are restored.  ``ckpt``: the dict, or a path to a torch-saved one."""
        if isinstance(ckpt, (str, os.PathLike)):
            ckpt = torch.load(ckpt, map_location="cpu", weights_only=False)
        sd = {k.replace("_orig_mod.", "").replace("module.", ""): v for k, v in ckpt["model_ema"].items()}
        self.load_state_dict(sd)
        self.load_optimizer_state_dict(ckpt["opt_state"])
        # the loop counter and Adam's bias-correction count are different things: a checkpoint saved before the first optimizer step of a
        # resumed run (empty Adam state) keeps Adam at step 0 -- fresh moments with step = global_step would be under-corrected
        self.global_step = int(ckpt.get("global_step", self.step))
        return self

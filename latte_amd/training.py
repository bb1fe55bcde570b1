"""Training on the engine: the body of the reference's optimisation loop (train.py:197-236) behind one object.

    trainer = LatteTrainer(model, diffusion, max_batch=5)          # model: latte_amd.Latte, diffusion: create_diffusion("")
    out = trainer.train_step(x_start, y=labels)                     # q_sample + forward + losses + backward + clip + AdamW + EMA
    trainer.ema_state_dict() / model.state_dict()                  # reference checkpoint format (train.py:257-262)

What runs where: the forward with saved activations, the loss terms and their gradient, the backward (MFMA GEMMs for the
input / weight gradients, the attention backward, LayerNorm-modulate / GELU / gate backward kernels), gradient norm + clipping,
AdamW and the EMA update are engine kernels behind ``latte_trainer_*`` (include/latte_amd.h).  PyTorch holds the flat fp32
buffers (the model's ``nn.Parameter``s become views of the parameter buffer, as DistributedDataParallel's buckets do) and
averages the gradient buffer across ranks with ONE RCCL all-reduce per step when ``torch.distributed`` is initialised
(train.py:125 wraps the model in DDP; the engine itself never communicates).  There is no CPU / autograd fallback.
"""
import torch

from . import _lib
from ._lib import LatteError, check, load_library, ptr, stream_ptr
from .diffusion import SpacedDiffusion, _LOSS
from .models import Latte

FROZEN = ("pos_embed", "temp_embed")   # nn.Parameter(requires_grad=False), latte.py:246-247


def _world(group=None):
    import torch.distributed as dist
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def average_gradients(flat_grads, group=None, async_op=False):
    """DistributedDataParallel's gradient averaging (train.py:125) as one all-reduce over a (slice of a) flat gradient buffer:
    RCCL over xGMI when the process group's backend is nccl, gloo in the CPU tests.  No-op without an initialised process group.
    async_op: returns the work handle (None when there is nothing to do); the caller divides by the world size after wait()."""
    import torch.distributed as dist
    world = _world(group)
    if world <= 1:
        return None if async_op else flat_grads
    if async_op:
        return dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=group, async_op=True)
    dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=group)
    flat_grads.div_(world)
    return flat_grads


class LatteTrainer:
    """One data-parallel replica of train.py's model / ema / AdamW triple.

    lr, betas, eps, weight_decay: ``torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0)`` (train.py:127);
    clip_max_norm / start_clip_iter: train.py:228-231 (the gradient norm is always computed, clipping starts at that step);
    ema_decay: utils.update_ema's default; class_dropout_prob: LabelEmbedder's (latte.py:130), applied here because the engine
    takes the labels AFTER token_drop.
    compute_dtype: MFMA operand type of the forward / backward GEMMs and attention products; masters, gradients, AdamW state and
    EMA are fp32 either way.  "f16" runs the backward loss-scaled (``loss_scale``, a power of two, default 2**14: the per-token
    gradients of 1e-7 ... 1e-4 would underflow f16 otherwise; scaling and unscaling by a power of two is exact) and has the 10
    mantissa bits of the TF32 matmuls the reference trains with (train.py:12-14) -- the default: every gradient tensor within
    3.8e-4 relative L2 of the reference's fp32 gradients (tests/test_training_step.py); "bf16" needs no scaling, has 7 bits
    (3.2e-3) and is 1.6 % faster.
    Overflow handling (round 4): a non-finite gradient norm skips the update and, with ``dynamic_loss_scale`` (default: on for
    f16, off for bf16), halves the loss scale; 2000 applied updates in a row double it again -- all inside the engine's optimiser
    step, no host synchronisation (``scaler_state()`` reads the counters back for logging).  ``train_steps`` is the TRAINING-step
    counter of train.py:195-236 (clipping starts at ``start_clip_iter``, checkpoints are numbered by it, a continued run sets it to
    the checkpoint's step); AdamW's bias correction uses the engine's own count of applied updates, which starts at 0 with the
    fresh moments -- as ``torch.optim.AdamW`` does in the reference."""

    def __init__(self, model, diffusion, max_batch, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip_max_norm=0.1,
                 start_clip_iter=20000, ema_decay=0.9999, class_dropout_prob=0.1, compute_dtype="f16", process_group=None,
                 loss_scale=None, dynamic_loss_scale=None):
        if not isinstance(model, Latte):
            raise LatteError("LatteTrainer needs a latte_amd.Latte model")
        if not isinstance(diffusion, SpacedDiffusion):
            raise LatteError("LatteTrainer needs a latte_amd SpacedDiffusion (create_diffusion)")
        if model.extras not in (1, 2):
            raise LatteError("T2V training are Not supported at this moment!")             # train.py:213-214
        if diffusion.loss_type not in ("mse", "rescaled_mse"):
            raise LatteError("the engine trains the MSE loss types (create_diffusion's default and rescale_learned_sigmas)")
        if compute_dtype not in _lib.DTYPES:
            raise LatteError(f"compute_dtype must be one of {sorted(_lib.DTYPES)}")
        _lib.require_gpu()
        self.model, self.diffusion = model, diffusion
        self.max_batch = int(max_batch)
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), tuple(betas), float(eps), float(weight_decay)
        self.clip_max_norm, self.start_clip_iter, self.ema_decay = float(clip_max_norm), int(start_clip_iter), float(ema_decay)
        self.class_dropout_prob = float(class_dropout_prob)
        self.process_group = process_group
        self.always_staged = False     # test hook: take the staged (bucketed) backward path without a process group
        self.train_steps = 0
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise LatteError("move the model to the GPU first (model.to('cuda'))")
        self.device = dev
        lib = load_library()
        cfg = model.engine_config(compute_dtype)
        h = _lib.c_void()
        with torch.cuda.device(dev):
            check(lib.latte_trainer_create(cfg, self.max_batch, h))
        self._h = h
        if loss_scale is not None:
            check(lib.latte_trainer_set_option(h, b"loss_scale", float(loss_scale)))
        if dynamic_loss_scale is not None:
            check(lib.latte_trainer_set_option(h, b"dynamic_loss_scale", float(bool(dynamic_loss_scale))))
        self.compute_dtype = compute_dtype
        n = lib.latte_trainer_num_params(h)
        self.layout = [(lib.latte_trainer_param_key(h, i).decode(), int(lib.latte_trainer_param_offset(h, i)),
                        int(lib.latte_trainer_param_numel(h, i))) for i in range(n)]
        total = int(lib.latte_trainer_total_numel(h))
        named = dict(model.named_parameters())
        want = [k for k in named if k not in FROZEN]
        if [k for k, _, _ in self.layout] != want:
            raise LatteError("parameter order of the engine and of the model differ")
        self.params = torch.zeros(total, device=dev)
        self.grads = torch.zeros(total, device=dev)
        self.exp_avg = torch.zeros(total, device=dev)
        self.exp_avg_sq = torch.zeros(total, device=dev)
        self.ema = torch.zeros(total, device=dev)
        with torch.no_grad():
            for k, off, numel in self.layout:
                p = named[k]
                self.params[off:off + numel].copy_(p.detach().reshape(-1).float())
                p.data = self.params[off:off + numel].view(p.shape)          # the module's parameters are views of the flat buffer
            self.ema.copy_(self.params)                                        # update_ema(ema, model, decay=0), train.py:165
        with torch.cuda.device(dev):
            check(lib.latte_trainer_bind(h, ptr(self.params), ptr(self.grads), ptr(self.exp_avg), ptr(self.exp_avg_sq), ptr(self.ema)))
            check(lib.latte_trainer_set_frozen(h, ptr(named["pos_embed"].detach().float().contiguous()),
                                               ptr(named["temp_embed"].detach().float().contiguous()), 1, stream_ptr()))
            check(lib.latte_trainer_sync_weights(h, stream_ptr()))
        self._norm = torch.zeros(2, device=dev)
        if hasattr(model, "mark_weights_dirty"):
            model.mark_weights_dirty()

    def set_option(self, name, value):
        """Engine options of the trainer (latte_trainer_set_option): "loss_scale", "dynamic_loss_scale", "loss_scale_growth_interval",
        "fuse_gelu" (0: separate GELU passes, 1: inside the fc1 / fc2-gradient GEMMs -- the default), "fuse_small" (0: the
        separate finalize / column-sum / adaLN / gate-backward launches of rounds 2 - 6, 1: folded -- the default, csrc/train_fin.hip;
        not while a step is in flight)."""
        check(load_library().latte_trainer_set_option(self._h, name.encode(), float(value)))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                load_library().latte_trainer_destroy(self._h)
        except Exception:
            pass

    # ------------------------------------------------------------------ views
    def grad_dict(self):
        """{reference key: gradient view} of the last forward_backward."""
        named = dict(self.model.named_parameters())
        return {k: self.grads[off:off + numel].view(named[k].shape) for k, off, numel in self.layout}

    def model_state_dict(self):
        """The ``"model"`` entry of train.py's checkpoints (:257-262), built from the flat fp32 master buffer through the layout --
        not through the module's parameter views: a later ``model.cuda()`` / ``.to()`` / ``.float()`` re-materialises the
        ``nn.Parameter``s and silently ends the aliasing, after which ``model.state_dict()`` would be stale."""
        sd = {k: v.detach().clone() for k, v in self.model.state_dict().items()}
        for k, off, numel in self.layout:
            sd[k] = self.params[off:off + numel].view(sd[k].shape).clone()
        return sd

    def check_aliasing(self):
        """True while every trained ``nn.Parameter`` of the module still is a view of the flat master buffer."""
        named = dict(self.model.named_parameters())
        base = self.params.data_ptr()
        return all(named[k].data_ptr() == base + 4 * off for k, off, _ in self.layout)

    def ema_state_dict(self):
        """The ``"ema"`` entry of train.py's checkpoints (:257-262): every key of model.state_dict()."""
        sd = {k: v.detach().clone() for k, v in self.model.state_dict().items()}
        for k, off, numel in self.layout:
            sd[k] = self.ema[off:off + numel].view(sd[k].shape).clone()
        return sd

    def load_state_dict(self, sd, ema_sd=None):
        with torch.no_grad():
            for k, off, numel in self.layout:
                self.params[off:off + numel].copy_(sd[k].reshape(-1).float())
                self.ema[off:off + numel].copy_((ema_sd or sd)[k].reshape(-1).float())
        with torch.cuda.device(self.device):
            check(load_library().latte_trainer_sync_weights(self._h, stream_ptr()))

    # ------------------------------------------------------------------ one iteration of train.py:197-236
    def forward_backward(self, x_start, t, noise, y=None, drop_mask=None, return_model_out=False, overlap_all_reduce=False):
        """q_sample + forward + training_losses + backward; gradients of ``terms['loss'].mean()`` land in ``self.grads``
        (overlap_all_reduce: already averaged over the process group, bucket by bucket under the backward).
        -> dict(loss, mse, vb [, model_out])."""
        self._reduced = False
        d = self.diffusion
        x0 = x_start.to(device=self.device, dtype=torch.float32).contiguous()
        B = x0.shape[0]
        if B > self.max_batch:
            raise LatteError(f"batch {B} exceeds max_batch {self.max_batch}")
        nz = noise.to(device=self.device, dtype=torch.float32).contiguous()
        if nz.shape != x0.shape:
            raise AssertionError("noise.shape == x_start.shape")
        t64 = t.to(device=self.device, dtype=torch.int64).contiguous()
        yy = None
        if self.model.extras == 2:
            if y is None:
                raise LatteError("class-conditional model: labels required")
            yy = y.to(device=self.device, dtype=torch.int64)
            if drop_mask is not None:                                              # LabelEmbedder.token_drop, latte.py:138-148
                yy = torch.where(drop_mask.to(self.device), torch.full_like(yy, self.model.num_classes), yy)
            if int(yy.min()) < 0 or int(yy.max()) > self.model.num_classes:
                raise IndexError("label out of range")
            yy = yy.contiguous()
        terms = torch.empty(3, B, device=self.device)
        mo = torch.empty(B, x0.shape[1], self.model.out_channels, x0.shape[3], x0.shape[4], device=self.device) if return_model_out else None
        lib = load_library()
        args = (self._h, d._h, _LOSS[d.loss_type], ptr(x0), ptr(nz), ptr(t64), ptr(yy) if yy is not None else None, B, ptr(terms),
                ptr(mo) if mo is not None else None, stream_ptr())
        with torch.cuda.device(self.device):
            if not overlap_all_reduce:
                check(lib.latte_trainer_forward_backward(*args))
            else:
                # bucketed data parallelism: as soon as a stage's gradient slice is final its all-reduce is enqueued (RCCL runs it
                # on its own stream behind the kernels already launched) while the next stages' kernels follow on this stream
                check(lib.latte_trainer_begin(*args))
                handles = []
                for k in range(lib.latte_trainer_num_stages(self._h)):
                    check(lib.latte_trainer_backward_stage(self._h, k, stream_ptr()))
                    off, n = _lib.c_i64(), _lib.c_i64()
                    check(lib.latte_trainer_stage_range(self._h, k, off, n))
                    h = average_gradients(self.grads[off.value:off.value + n.value], self.process_group, async_op=True)
                    if h is not None:
                        handles.append(h)
                for h in handles:
                    h.wait()
                if handles:
                    self.grads.div_(_world(self.process_group))
                self._reduced = True
        out = {"loss": terms[0], "mse": terms[1]}
        if d.learn_sigma:
            out["vb"] = terms[2]
        if mo is not None:
            out["model_out"] = mo
        return out

    def all_reduce_gradients(self):
        """DDP's gradient averaging (train.py:125) as ONE collective over the flat gradient buffer: RCCL over xGMI when the
        process group's backend is nccl, gloo in the CPU tests."""
        if not getattr(self, "_reduced", False):
            average_gradients(self.grads, self.process_group)
            self._reduced = True

    def optimizer_step(self):
        """clip_grad_norm_ + AdamW + update_ema; -> gradient norm (0-d tensor, device)."""
        self.train_steps += 1
        clip = int(self.train_steps - 1 >= self.start_clip_iter)                  # train.py:228-231
        with torch.cuda.device(self.device):
            # step = 0: AdamW's bias correction counts the engine's APPLIED updates (fresh moments start at 1), not train_steps
            check(load_library().latte_trainer_optimizer_step(self._h, self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                                                              0, self.clip_max_norm, clip, self.ema_decay,
                                                              ptr(self._norm), stream_ptr()))
        if hasattr(self.model, "mark_weights_dirty"):
            self.model.mark_weights_dirty()
        return self._norm[0]

    def scaler_state(self):
        """Loss-scaling / update counters of the engine (synchronises): dict(loss_scale, good_steps, applied_updates,
        skipped_updates, last_skipped, dynamic, growth_interval, max_scale)."""
        import ctypes
        out = (ctypes.c_double * 8)()
        check(load_library().latte_trainer_scaler_state(self._h, out))
        keys = ("loss_scale", "good_steps", "applied_updates", "skipped_updates", "last_skipped", "dynamic", "growth_interval", "max_scale")
        return dict(zip(keys, [float(v) for v in out]))

    def train_step(self, x_start, y=None, t=None, noise=None, drop_mask=None):
        """train.py:197-236 for one micro-batch (gradient_accumulation_steps = 1)."""
        B = x_start.shape[0]
        if t is None:
            t = torch.randint(0, self.diffusion.num_timesteps, (B,), device=self.device)       # train.py:223
        if noise is None:
            noise = torch.randn_like(x_start, dtype=torch.float32, device=self.device)          # gd:733-734
        if drop_mask is None and self.model.extras == 2 and self.class_dropout_prob > 0:
            drop_mask = torch.rand(B, device=self.device) < self.class_dropout_prob              # latte.py:142-143
        out = self.forward_backward(x_start, t, noise, y, drop_mask, overlap_all_reduce=_world(self.process_group) > 1 or self.always_staged)
        self.all_reduce_gradients()
        out["grad_norm"] = self.optimizer_step()
        return out

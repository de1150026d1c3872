"""TasuEngine -- the MI355X replacement of the DeepSpeed engine the reference trains with
(``deepspeed.initialize`` at Multitask/finetune_deepspeed.py:147-149; ``model.backward(loss)`` / ``model.step()``
at Multitask/utils/deepspeed_utils.py:235-236; config Multitask/conf/ds_config.json).

Semantics reproduced: every rank computes the mean CE over ITS OWN non-ignored tokens; gradients are averaged
over ranks (DeepSpeed's post-divide by world size); AdamW (decoupled decay, bias correction, fp32 state) on the
trainable parameters only; WarmupCosineLR stepped after every optimizer step.  ZeRO-2 partitioning is dropped:
with 54.5 M trainable parameters it would save 0.6 GB per GPU of 288 GB, and replicated DP with one all-reduce
is the same update.

Data parallelism: one process per GPU, ``torch.distributed`` (backend "nccl" == RCCL over xGMI); ONE flat fp32
gradient bucket (the projector's flat buffer, 218 MB) is all-reduced on a side HIP stream that is event-chained
behind the wgrad kernels; ``step()`` makes the compute stream wait for it and folds the 1/world_size into the
fused AdamW kernel.  No data-path collective exists besides this one.
"""
import math

import torch
import torch.distributed as dist


def warmup_cosine_ratio(it, warmup_num_steps=200, total_num_steps=15000, warmup_min_ratio=0.0, cos_min_ratio=1e-4,
                        warmup_type="log"):
    """DeepSpeed WarmupCosineLR ratio at scheduler iteration ``it`` (``last_batch_iteration``).  DeepSpeed is not part
    of the reference tree (unpinned dependency): restated from its published formula, PARITY UNPINNED."""
    warmup_num_steps = max(2, warmup_num_steps)
    if it < 0:
        return 0.0
    if it < warmup_num_steps:
        r = math.log(it + 1) / math.log(warmup_num_steps) if warmup_type == "log" else it / warmup_num_steps
        return warmup_min_ratio + (1.0 - warmup_min_ratio) * r
    real_last = it - warmup_num_steps + 1
    real_total = total_num_steps - warmup_num_steps
    return max(0.0, cos_min_ratio + (1 - cos_min_ratio) * 0.5 * (1 + math.cos(math.pi * real_last / real_total)))


class TasuEngine:
    def __init__(self, module, ds_config, process_group=None):
        """module: ps_slm_amd.ps_slm.slam_model_asr (exposes .core = TasuModel).  ds_config: dict from
        ps_slm_amd.config.load_ds_config."""
        self.module = module
        self.core = module.core
        self.cfg = ds_config
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if self.world > 1 else 0
        self.global_steps = 0          # optimizer steps taken
        self.sched_iter = -1           # WarmupCosineLR.last_batch_iteration (scheduler steps AFTER the optimizer)
        self.micro_steps = 0
        dev = self.core.device
        self.lr_dev = torch.zeros(1, dtype=torch.float32, device=dev)
        self._lr_host = torch.zeros(1, dtype=torch.float32).pin_memory() if dev.type == "cuda" else torch.zeros(1)
        self.comm_stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self._pending = None
        self._last_state = None
        # gradient accumulation (ds_config "gradient_accumulation_steps", DeepSpeed semantics: every micro-step's gradient
        # enters with weight 1/k, the optimizer and the scheduler step on every k-th call of step()); k = 1 in the shipped
        # config, in which case no accumulation buffer exists and nothing below costs anything
        self.ga = max(1, int(ds_config.get("gradient_accumulation_steps", 1)))
        self._g_acc = torch.zeros_like(self.core.proj.g) if self.ga > 1 else None

    # ---- nn.Module-like surface the reference's train() uses (deepspeed_utils.py:136-246)
    def train(self):
        self.module.train()
        return self

    def eval(self):
        self.module.eval()
        return self

    def __call__(self, **batch):
        out = self.module(**batch)
        self._last_state = self.module.last_state
        return out

    def parameters(self):
        return self.module.parameters()

    def get_lr(self):
        return [self.cfg["lr"] * warmup_cosine_ratio(self.sched_iter, self.cfg["warmup_num_steps"],
                                                     self.cfg["total_num_steps"], self.cfg["warmup_min_ratio"],
                                                     self.cfg["cos_min_ratio"], self.cfg["warmup_type"])]

    # ---- backward + gradient exchange
    def backward(self, loss=None):
        """Hand-scheduled backward of the last forward, then the DP all-reduce of the flat gradient bucket on the
        side stream (asynchronous w.r.t. the compute stream until step())."""
        st = self._last_state
        if st is None:
            raise RuntimeError("backward() called before a forward pass")
        self.core.run_backward(st)
        self.micro_steps += 1
        if self.ga > 1:
            self._g_acc.add_(self.core.proj.g, alpha=1.0 / self.ga)
            if self.micro_steps % self.ga != 0:
                return                                  # not a boundary: no exchange yet
        if self.world > 1:
            g = self._g_acc if self.ga > 1 else self.core.proj.g
            if self.comm_stream is not None:
                self.comm_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self.comm_stream):
                    self._pending = dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            else:
                self._pending = dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

    def is_gradient_accumulation_boundary(self):
        return self.micro_steps % self.ga == 0

    def step(self):
        if self.ga > 1 and not self.is_gradient_accumulation_boundary():
            return                                      # DeepSpeed: step() between boundaries is a no-op
        if self._pending is not None:
            self._pending.wait()
            if self.comm_stream is not None:
                torch.cuda.current_stream().wait_stream(self.comm_stream)
            self._pending = None
        c, pr = self.cfg, self.core.proj
        self.global_steps += 1
        self._lr_host[0] = self.get_lr()[0]
        self.lr_dev.copy_(self._lr_host, non_blocking=True)
        g = self._g_acc if self.ga > 1 else pr.g
        self.core.ops.adamw(pr.p, g, pr.m, pr.v, pr.pb, self.lr_dev, c["betas"][0], c["betas"][1], c["eps"],
                            c["weight_decay"], self.global_steps, 1.0 / self.world)
        if self.ga > 1:
            self._g_acc.zero_()
        pr.refresh_working_copies(self.core.ops)
        self.sched_iter += 1           # lr_scheduler.step() follows optimizer.step() in the DeepSpeed engine

    # ---- uneven-data join (replaces the per-step gloo monitored_barrier of deepspeed_utils.py:102-123,191)
    def all_have_data(self, has_batch: bool) -> bool:
        if self.world == 1:
            return has_batch
        flag = torch.tensor([1 if has_batch else 0], dtype=torch.int32, device=self.core.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.pg)
        return bool(flag.item())

    def reduce_scalars(self, *vals):
        """SUM-all-reduce of logged scalars (deepspeed_utils.py:321-322, :476-477)."""
        t = torch.tensor(list(vals), dtype=torch.float32, device=self.core.device)
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)
        return t.tolist()

    # ---- checkpoint (projector tensors only, reference key names: checkpoint_handler.py:169-182, ps-slm.py:163-170)
    def save_checkpoint(self, path):
        if self.rank == 0:
            torch.save({k: v.cpu() for k, v in self.core.projector_state_dict().items()}, path)
        if self.world > 1:
            dist.barrier(group=self.pg)

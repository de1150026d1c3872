"""TasuEngine -- the MI355X replacement of the DeepSpeed engine the reference trains with
(``deepspeed.initialize`` at Multitask/finetune_deepspeed.py:147-149; ``model.backward(loss)`` / ``model.step()``
at Multitask/utils/deepspeed_utils.py:235-236; config Multitask/conf/ds_config.json).

Semantics reproduced: every rank computes the mean CE over ITS OWN non-ignored tokens; gradients are averaged
over ranks (DeepSpeed's post-divide by world size); AdamW (decoupled decay, bias correction, fp32 state) on the
trainable parameters only; WarmupCosineLR stepped after every optimizer step.  ZeRO-2 partitioning is dropped:
with 54.5 M trainable parameters it would save 0.6 GB per GPU of 288 GB, and replicated DP with all-reduce
is the same update.

Data parallelism (``overlap_comm`` of ds_config.json:15-21, rebuilt for this step): one process per GPU.  On the GPU the
collective is RCCL over xGMI through the C-ABI (``tasu_comm_init`` / ``tasu_allreduce_f32``, include/tasu_hip.h: RcclExchange
below); ``torch.distributed`` is the launcher's rendezvous (it carries RCCL's unique id to the ranks) and the control plane
(logged scalars, checkpoint barrier) -- and, over gloo, the collective of the CPU test double and of the two-ranks-on-one-GPU
test, which RCCL cannot host.  The flat fp32 gradient bucket (218 MB) is exchanged in the
ranges the backward completes them in (TasuModel.grad_ranges): the Linear2 / bias tail, then N row blocks of the
Linear1 weight gradient (94 % of the bytes), then the LayerNorm parameters.  Each range's all-reduce is issued on a side
HIP stream behind an event recorded right after its wgrad kernel, so it runs under the remaining backward kernels
(later row blocks, the dxn GEMM, the LayerNorm-parameter reduction); ``step()`` then walks the ranges in the same order:
wait for range i, fused AdamW on range i (1/world folded in) -- which runs while range i+1 is still on the wire.  The
projector is the FIRST layer of the network, so its gradients are the LAST thing backward produces: what cannot be
hidden is the all-reduce of the final row block.  No other data-path collective exists.
"""
import ctypes
import math

import torch
import torch.distributed as dist

from .streams import side_stream


class RcclExchange:
    """The data-path collective on the GPU: one RCCL communicator per process behind the C-ABI (csrc/comm.hip).  The unique id
    travels over the process group the launcher set up (control plane); the all-reduces themselves never touch
    torch.distributed."""

    def __init__(self, lib, pg, device):
        self.lib, self.comm = lib, None
        rank, world = dist.get_rank(pg), dist.get_world_size(pg)
        src = dist.get_global_rank(pg, 0) if pg is not None else 0
        # Every step of the bootstrap is agreed on by ALL ranks before the next one starts: a rank that cannot bind RCCL (or
        # rank 0 failing to draw the unique id) must not leave the others blocked in a broadcast or inside ncclCommInitRank.
        note = ctypes.create_string_buffer(640)
        bound = lib.tasu_comm_library(note, len(note)) == 0
        ident = (ctypes.c_uint8 * 128)()
        ok = bound and (rank != 0 or lib.tasu_comm_unique_id(ident) == 0)
        box = [(bool(ok), bytes(ident), note.value.decode(errors="replace"))]
        if world > 1:
            gathered = [None] * world
            dist.all_gather_object(gathered, box[0], group=pg)
            bad = [(r, g[2]) for r, g in enumerate(gathered) if not g[0]]
            if bad:
                raise RuntimeError("RCCL bootstrap failed on rank(s) " + "; ".join(f"{r}: {why or 'tasu_comm_unique_id failed'}" for r, why in bad)
                                   + " (TASU_RCCL_PATH selects the library; every rank raises this error)")
            box = [gathered[0]]
        elif not ok:
            raise RuntimeError(f"RCCL bootstrap failed: {box[0][2] or 'tasu_comm_unique_id failed'}")
        self.library = box[0][2] if rank == 0 else note.value.decode(errors="replace")
        ident = (ctypes.c_uint8 * 128).from_buffer_copy(box[0][1])
        torch.cuda.set_device(device)
        comm = ctypes.c_void_p()
        rc = lib.tasu_comm_init(ident, rank, world, ctypes.byref(comm))
        if world > 1:                                  # agree again: a rank whose init failed has already left the collective
            flags = [None] * world
            dist.all_gather_object(flags, int(rc), group=pg)
            if any(flags):
                raise RuntimeError(f"tasu_comm_init failed on rank(s) {[r for r, f in enumerate(flags) if f]} of {world}")
        elif rc:
            raise RuntimeError(f"tasu_comm_init failed (rank {rank} of {world})")
        self.comm = comm
        n = ctypes.c_int(0)
        if lib.tasu_comm_count(comm, ctypes.byref(n)) or n.value != world:
            raise RuntimeError(f"RCCL sees {n.value} ranks in the communicator, the launcher {world}")
        self.ranks = n.value
        _OPEN.add(self)

    def all_reduce_sum(self, t, stream):
        """In-place SUM over the ranks of a contiguous fp32 tensor, asynchronous on ``stream``."""
        if self.lib.tasu_allreduce_f32(self.comm, t.data_ptr(), t.numel(), stream.cuda_stream):
            raise RuntimeError("tasu_allreduce_f32 failed")

    def all_reduce_min_i32(self, t, stream):
        if self.lib.tasu_allreduce_min_i32(self.comm, t.data_ptr(), t.numel(), stream.cuda_stream):
            raise RuntimeError("tasu_allreduce_min_i32 failed")

    def close(self):
        """Destroys the communicator (TasuEngine.destroy(), or at interpreter exit BEFORE torch.distributed tears its own
        process group down: a live RCCL communicator at exit can hang or warn)."""
        _OPEN.discard(self)
        if self.comm:
            torch.cuda.synchronize()
            self.lib.tasu_comm_destroy(self.comm)
            self.comm = None


_OPEN = set()


def _close_all():
    for x in list(_OPEN):
        try:
            x.close()
        except Exception:          # interpreter shutdown: nothing left to report to
            pass


import atexit  # noqa: E402

atexit.register(_close_all)


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
    def __init__(self, module, ds_config, process_group=None, w1_chunks=None, force_exchange=False):
        """module: ps_slm_amd.ps_slm.slam_model_asr (exposes .core = TasuModel).  ds_config: dict from
        ps_slm_amd.config.load_ds_config.  ``w1_chunks``: row blocks the Linear1 wgrad is exchanged in (default 4 when
        gradients are exchanged, 1 otherwise).  ``force_exchange``: run the collectives even at world size 1 (tests)."""
        self.module = module
        self.core = module.core
        self.cfg = ds_config
        tc = getattr(module, "train_config", None)
        if tc is not None and not tc.get("freeze_encoder", False) and not getattr(module, "gt_emb", True):
            # the reference then leaves the SenseVoice encoder's parameters trainable (ps-slm.py:31-40); this engine has no
            # backward through the encoder -- say so instead of silently training less than was asked for
            raise NotImplementedError("train_config.freeze_encoder=false with the audio branch: training the SenseVoice encoder is not "
                                      "built (Multitask/scripts/finetune_deespeed_sensevoice.sh:44 ships freeze_encoder=true)")
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if self.world > 1 else 0
        self.exchange = self.world > 1 or (force_exchange and dist.is_available() and dist.is_initialized())
        self.w1_chunks = int(w1_chunks) if w1_chunks else (4 if self.exchange else 1)
        self.global_steps = 0          # optimizer steps taken
        self.sched_iter = -1           # WarmupCosineLR.last_batch_iteration (scheduler steps AFTER the optimizer)
        self.micro_steps = 0
        dev = self.core.device
        # (a stream on its own hardware queue: ps_slm_amd/streams.py; only when gradients are exchanged)
        self.comm_stream = side_stream(dev, "gradient exchange", owner=self) if self.exchange else None
        # the collective: RCCL through the C-ABI when the ranks were launched with the nccl backend (one GPU per rank); a gloo
        # group (CPU double; two ranks sharing one GPU in tests) keeps torch.distributed's all_reduce
        self.rccl = None
        if self.exchange and dev.type == "cuda" and dist.get_backend(process_group) == "nccl":
            self.rccl = RcclExchange(self.core.ops.lib, process_group, dev)
        self._pending = []             # [(lo, hi, work)] in issue order
        self._last_state = None
        self.time_exchange = False     # bench.py: record event pairs around every wait of step()
        self.exposed_events = []
        self.trace_exchange = False    # tests: timed events per exchanged range (issue point on the compute stream, start / end of
        self.exchange_trace = []       # its all-reduce on the side stream): [(lo, hi, issued, started, ended)]
        # gradient accumulation (ds_config "gradient_accumulation_steps"; 1 in the shipped config, in which case no
        # accumulation buffer exists and nothing below costs anything).  Reference semantics: the loop divides the loss by k
        # (deepspeed_utils.py:210) and DeepSpeed's backward scales by 1/k again, so each micro-batch gradient enters with
        # weight 1/k^2; the optimizer and the scheduler step on every k-th call of step().
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

    def prefetch(self, **batch) -> bool:
        """The next batch's frozen encoder pass under this batch's decoder step (slam_model_asr.prefetch); call it between
        ``engine(**batch)`` and ``engine.backward()`` with the batch the next ``engine(**...)`` call will get."""
        fn = getattr(self.module, "prefetch", None)
        return bool(fn(**batch)) if fn is not None else False

    def parameters(self):
        return self.module.parameters()

    def get_lr(self):
        return [self.cfg["lr"] * warmup_cosine_ratio(self.sched_iter, self.cfg["warmup_num_steps"],
                                                     self.cfg["total_num_steps"], self.cfg["warmup_min_ratio"],
                                                     self.cfg["cos_min_ratio"], self.cfg["warmup_type"])]

    # ---- backward + gradient exchange
    def _issue(self, g, lo, hi):
        """All-reduce of bucket range [lo, hi) on the side stream, ordered behind everything the compute stream has been
        given so far (= the kernels that wrote the range)."""
        if self.comm_stream is not None:
            ev = torch.cuda.Event(enable_timing=self.trace_exchange)
            ev.record()
            self.comm_stream.wait_event(ev)
            if self.rccl is not None:
                if self.trace_exchange:
                    t0 = torch.cuda.Event(enable_timing=True)
                    t0.record(self.comm_stream)
                self.rccl.all_reduce_sum(g[lo:hi], self.comm_stream)
                work = torch.cuda.Event(enable_timing=self.trace_exchange)   # "this range has been reduced": step() waits for it
                work.record(self.comm_stream)
                if self.trace_exchange:
                    self.exchange_trace.append((lo, hi, ev, t0, work))
            else:
                with torch.cuda.stream(self.comm_stream):
                    work = dist.all_reduce(g[lo:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        else:
            work = dist.all_reduce(g[lo:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        self._pending.append((lo, hi, work))

    def backward(self, loss=None):
        """Hand-scheduled backward of the last forward; with N > 1 the bucket ranges are all-reduced on the side stream as
        the wgrad kernels complete them (asynchronous w.r.t. the compute stream until step())."""
        st = self._last_state
        if st is None:
            raise RuntimeError("backward() called before a forward pass")
        self.micro_steps += 1
        st.backward_ran = True                         # (outputs.loss.backward() of the same step would be a second pass: _HipStep refuses)
        if self.ga > 1:
            self.core.run_backward(st)
            self._g_acc.add_(self.core.proj.g, alpha=1.0 / (self.ga * self.ga))
            if self.micro_steps % self.ga == 0 and self.exchange:
                lo0 = self.core.lora.base if (self.core.freeze_projector and self.core.lora is not None) else 0
                self._issue(self._g_acc, lo0, self._g_acc.numel())
            return
        if self.exchange:
            g = self.core.proj.g
            self.core.run_backward(st, on_ready=lambda lo, hi: self._issue(g, lo, hi), w1_chunks=self.w1_chunks)
        else:
            self.core.run_backward(st)

    def is_gradient_accumulation_boundary(self):
        return self.micro_steps % self.ga == 0

    def step(self):
        if self.ga > 1 and not self.is_gradient_accumulation_boundary():
            return                                      # DeepSpeed: step() between boundaries is a no-op
        c, pr = self.cfg, self.core.proj
        self.global_steps += 1
        lr = self.get_lr()[0]
        g = self._g_acc if self.ga > 1 else pr.g
        lo0 = self.core.lora.base if (self.core.freeze_projector and self.core.lora is not None) else 0   # frozen projector: adapters only
        ranges = self._pending if self._pending else [(lo0, pr.numel, None)]
        for lo, hi, work in ranges:
            if work is not None:
                timed = self.time_exchange and self.comm_stream is not None
                if timed:
                    e0 = torch.cuda.Event(enable_timing=True)
                    e0.record()
                if isinstance(work, torch.cuda.Event):
                    torch.cuda.current_stream().wait_event(work)    # RCCL: the compute STREAM waits for this range only
                else:
                    work.wait()                         # gloo: host wait
                    if self.comm_stream is not None:
                        torch.cuda.current_stream().wait_stream(self.comm_stream)
                if timed:
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record()
                    self.exposed_events.append((e0, e1))
            self.core.ops.adamw(pr.p[lo:hi], g[lo:hi], pr.m[lo:hi], pr.v[lo:hi], pr.pb[lo:hi], lr, c["betas"][0], c["betas"][1],
                                c["eps"], c["weight_decay"], self.global_steps, 1.0 / self.world)
        self._pending = []
        if self.ga > 1:
            self._g_acc.zero_()
        self.core.refresh_working_copies()
        self.sched_iter += 1           # lr_scheduler.step() follows optimizer.step() in the DeepSpeed engine

    def exposed_ms(self):
        """Sum of the intervals the compute stream spent waiting for gradient ranges since ``exposed_events`` was cleared
        (needs ``time_exchange``; call after a device synchronize)."""
        return sum(a.elapsed_time(b) for a, b in self.exposed_events)

    def destroy(self):
        """Releases the RCCL communicator; call before ``dist.destroy_process_group()`` (the entrypoints and bench.py do)."""
        if self.rccl is not None:
            self.rccl.close()
            self.rccl = None
        if self.comm_stream is not None:
            from .streams import release
            release(self.comm_stream, owner=self)       # (the engine may be kept alive by a caller's reference: do not wait for gc)

    def comm_info(self):
        """What the data-path collective runs on: {"backend", "ranks" (RCCL's own ncclCommCount), "library"}."""
        if self.rccl is not None:
            return {"backend": "rccl (C-ABI)", "ranks": self.rccl.ranks, "library": self.rccl.library}
        return {"backend": dist.get_backend(self.pg) if self.exchange else None, "ranks": self.world, "library": None}

    # ---- uneven-data join (replaces the per-step gloo monitored_barrier of deepspeed_utils.py:102-123,191)
    def all_have_data(self, has_batch: bool) -> bool:
        if self.world == 1:
            return has_batch
        flag = torch.tensor([1 if has_batch else 0], dtype=torch.int32, device=self.core.device)
        if self.rccl is not None:
            self.rccl.all_reduce_min_i32(flag, torch.cuda.current_stream())
        else:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.pg)
        return bool(flag.item())

    def reduce_scalars(self, *vals):
        """SUM-all-reduce of logged scalars (deepspeed_utils.py:321-322, :476-477)."""
        t = torch.tensor(list(vals), dtype=torch.float32, device=self.core.device)
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)
        return t.tolist()

    # ---- checkpoint (the trainable tensors -- projector, + LoRA adapters with use_peft -- under the reference's key names:
    #      checkpoint_handler.py:169-182, ps-slm.py:163-170)
    def save_checkpoint(self, path):
        if self.rank == 0:
            torch.save({k: v.cpu() for k, v in self.module.state_dict().items()}, path)          # the trainable tensors
        if self.world > 1:
            dist.barrier(group=self.pg)

"""Training entrypoint with the surface of the reference's ``Multitask/finetune_deepspeed.py`` (hydra-style
``++section.key=value`` overrides, ``--local_rank``, RANK/LOCAL_RANK/WORLD_SIZE from the launcher env, ``deepspeed_config``
json) and the loop body of ``Multitask/utils/deepspeed_utils.py:train`` (:136-391), with the DeepSpeed engine replaced by
``TasuEngine`` (RCCL all-reduce + fused AdamW + WarmupCosineLR).

    python -m torch.distributed.run --nproc-per-node 8 -m ps_slm_amd.finetune_deepspeed \
        ++model_config.llm_path=synthetic:qwen2.5-1.5b ++model_config.encoder_projector=linear-silu \
        ++train_config.freeze_llm=true ++train_config.gt_emb=true ++train_config.gt_emb_noise=true \
        ++train_config.ctc_posterior=true ++dataset_config.file=synthetic ++deepspeed_config=conf/ds_config.json

The dataset plugin (``dataset_config.file = "<file>.py:<func>"``, Multitask/utils/dataset_utils.py:28-57) is honoured;
``dataset_config.file=synthetic`` yields fixed-length synthetic batches in the collator's schema (the reference's
jsonl/fbank loader is SURVEY section 8f "next").
"""
import contextlib
import importlib.machinery
import importlib.util
import logging
import os
import random
import sys
import time
from pathlib import Path

import torch
import torch.distributed as dist

from .config import DEFAULT_DS_CONFIG, load_ds_config, parse_args
from .engine import TasuEngine
from .streams import ensure_hw_queues

logger = logging.getLogger(__name__)


def load_module_from_py_file(py_file):
    """Multitask/utils/dataset_utils.py:14-25."""
    name = Path(py_file).name
    loader = importlib.machinery.SourceFileLoader(name, py_file)
    spec = importlib.util.spec_from_loader(name, loader)
    module = importlib.util.module_from_spec(spec)
    loader.exec_module(module)
    return module


def get_custom_model_factory(model_config):
    """Multitask/utils/model_utils.py:9-33 (same ValueError / FileNotFoundError / AttributeError behaviour)."""
    path = model_config.get("file", None)
    if path is None:
        raise ValueError("must set correct model path")
    module_path, func_name = path.split(":") if ":" in path else (path, "model_factory")
    if not module_path.endswith(".py"):
        raise ValueError(f"Dataset file {module_path} is not a .py file.")
    p = Path(module_path)
    if not p.is_file():
        alt = Path(__file__).resolve().parent.parent / module_path          # relative to the repository root
        if not alt.is_file():
            raise FileNotFoundError(f"Dataset py file {p.as_posix()} does not exist or is not a file.")
        p = alt
    module = load_module_from_py_file(p.as_posix())
    try:
        return getattr(module, func_name)
    except AttributeError:
        logger.info("It seems like the given method name (%s) is not present in the model .py file (%s).", func_name, p)
        raise


def _id_word(i):
    """id -> lower-case word (base-26 digits a..z), e.g. 27 -> 'bb'."""
    w = ""
    i = int(i)
    while True:
        w = chr(ord("a") + i % 26) + w
        i //= 26
        if i == 0:
            return w


class SyntheticDataset:
    """Pre-batched iterable with a ``collator`` (the contract of MultiTaskDynamicBatchDataset,
    Multitask/dataset/speech_dataset_large.py:307-330) that yields the SURVEY 8d synthetic utterances."""

    def __init__(self, geo, batch_size, steps, rank, inference=False):
        self.geo, self.B, self.steps, self.rank, self.inference = geo, batch_size, steps, rank, inference

    def __len__(self):
        return self.steps * self.B

    def __iter__(self):
        from .synthetic import synthetic_text_batch
        for i in range(self.steps):
            yield synthetic_text_batch(self.geo, self.B, seed=1234 + self.rank + 7919 * i, noise=False)

    def collator(self, raw):
        b = dict(input_ids=raw["input_ids"], attention_mask=raw["attention_mask"], input_features=raw["input_features"],
                 input_feature_length=raw["input_feature_length"], GT=[" ".join(map(str, p)) for p in raw["post_ids"]])
        if self.inference:
            b["keys"] = [f"utt{i}" for i in range(len(b["GT"]))]
            # generate() cleans the transcripts with the reference's regex (letters and .,!? only, ps-slm.py:592-596), so
            # the synthetic transcripts are letter-words (one per pseudo-posterior row), not digit strings
            b["targets"] = [" ".join(_id_word(i) for i in p) for p in raw["post_ids"]]
            n = 25
            b["input_ids"], b["attention_mask"] = b["input_ids"][:, :n], b["attention_mask"][:, :n]
        else:
            b["labels"] = raw["labels"]
        return b


def get_dataset(dataset_config, tokenizer, split, geo, rank, steps=20, batch_size=16):
    f = dataset_config.get("file", "synthetic")
    if f == "synthetic" or f is None:
        return SyntheticDataset(geo, batch_size, steps, rank, inference=(split == "test"))
    module_path, func_name = f.split(":") if ":" in f else (f, "get_custom_dataset")
    if not module_path.endswith(".py"):
        raise ValueError(f"Dataset file {module_path} is not a .py file.")
    p = Path(module_path)
    if not p.is_file():
        alt = Path(__file__).resolve().parent.parent / module_path          # relative to the repository root
        if not alt.is_file():
            raise FileNotFoundError(f"Dataset py file {module_path} does not exist or is not a file.")
        p = alt
    module = load_module_from_py_file(p.as_posix())
    return getattr(module, func_name)(dataset_config, tokenizer, split)


def evaluation(engine, train_config, eval_dataset, rank, world):
    """Multitask/utils/deepspeed_utils.py:394-498: forward only, loss and accuracy summed over the batches and over ranks,
    ``eval_epoch_loss = (sum / num_steps) / world`` -- including the reference's step count under dynamic batching
    (``tot_step + 1`` = number of batches + 1, :479-481).  The reference also arg-maxes and batch-decodes every eval
    batch into a list it never reads (:457-463); that dead host work is not reproduced."""
    import math
    engine.eval()
    dev = engine.core.device
    tot = torch.zeros(2, dtype=torch.float32, device=dev)
    n = 0
    for raw in eval_dataset:
        batch = eval_dataset.collator(raw)
        outputs, acc = engine(**batch)
        tot[0] += outputs.loss.detach().float()
        tot[1] += acc if isinstance(acc, torch.Tensor) else float(acc)
        n += 1
    if world > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    num_steps = n if train_config.batching_strategy != "dynamic" else n + 1
    loss, acc = (tot / max(num_steps, 1) / world).tolist()
    ppl = math.exp(loss)
    if rank == 0:
        logger.info(" eval_ppl=%s eval_epoch_loss=%s eval_epoch_acc=%s", ppl, loss, acc)
    engine.train()
    return ppl, loss, acc


class BatchReader:
    """The reference feeds its loop from DataLoader worker PROCESSES (Multitask/finetune_deepspeed.py:185-208, num_workers 4 in
    finetune_deespeed_sensevoice.sh:92).  Here ONE reader THREAD does that work -- jsonl line -> audio container (tmpfs / page
    cache read, FLAC decode in libtasu_hip.so: ctypes and numpy release the GIL) -> front end (two HIP launches per utterance on
    the reader's OWN stream) -> tokenisation -> collate -- and hands finished batches over a bounded queue, in dataset order.
    Forked workers are not an option (a process that has initialised HIP must not fork) and are not needed: the features are
    made on the device, so what the host moves per 30-s utterance is 0.96 MB of PCM.  The consumer waits for the batch's event
    on its own stream before touching device tensors.  ``depth`` batches of lookahead bound the memory."""

    _END = object()

    def __init__(self, dataset, device, depth=3):
        import queue
        import threading
        self.dataset, self.device = dataset, torch.device(device)
        self.q = queue.Queue(maxsize=depth)
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self.thread = threading.Thread(target=self._run, name="tasu-batch-reader", daemon=True)
        self.stopped = False
        self.thread.start()

    def _run(self):
        try:
            if self.stream is not None:
                torch.cuda.set_device(self.device)
            ctx = torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()
            with ctx:
                for raw in self.dataset:
                    if self.stopped:
                        return
                    batch = self.dataset.collator(raw)
                    ev = None
                    if self.stream is not None and isinstance(batch.get("input_features"), torch.Tensor) and batch["input_features"].is_cuda:
                        ev = torch.cuda.Event()
                        ev.record(self.stream)
                    self.q.put((raw, batch, ev))
            self.q.put((self._END, None, None))
        except BaseException as e:                       # hand the failure to the training thread instead of dying silently
            self.q.put((e, None, None))

    def __iter__(self):
        while True:
            raw, batch, ev = self.q.get()
            if raw is self._END:
                return
            if isinstance(raw, BaseException):
                raise raw
            if ev is not None:
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)
                batch["input_features"].record_stream(cur)   # (allocated on the reader's stream, consumed on this one)
            yield raw, batch

    def close(self):
        self.stopped = True
        while self.thread.is_alive():                    # unblock a producer waiting on a full queue
            try:
                self.q.get_nowait()
            except Exception:
                self.thread.join(timeout=0.05)


class MetricsLog:
    """``log_config.use_wandb``: the records of the reference's ``wandb.log`` calls under the same keys and step numbering --
    ``train_inner/train_inner_{loss,accuracy}`` every ``log_interval`` steps (Multitask/utils/deepspeed_utils.py:213-230),
    ``valid/{val_epoch_loss,val_perplexity,best_val_loss,val_accuracy,val_best_accuracy}`` after every validation pass (:283-293),
    ``train/{train_perplexity,train_epoch_loss,train_epoch_acc}`` per epoch (:334-351) -- on rank 0.  They go to wandb itself when
    the package is importable (``wandb.init`` with the reference's arguments, Multitask/finetune_deepspeed.py:118-124) and ALWAYS
    to ``<wandb_dir>/metrics.jsonl``, one JSON object per call (this image has no wandb).  Deviation (stated): the loop reads loss
    and accuracy from the device only every ``log_interval`` steps (the reference's ``total_loss += loss`` is a host sync per
    step), so the per-epoch figures average the logged steps."""

    def __init__(self, log_config, rank, run_config=None):
        self.on = bool(getattr(log_config, "use_wandb", False)) and rank == 0
        self.wandb, self.f = None, None
        if not self.on:
            return
        os.makedirs(log_config.wandb_dir, exist_ok=True)
        self.path = os.path.join(log_config.wandb_dir, "metrics.jsonl")
        self.f = open(self.path, "a")
        try:
            import wandb
            wandb.init(dir=log_config.wandb_dir, entity=log_config.wandb_entity_name, project=log_config.wandb_project_name,
                       name=log_config.wandb_exp_name, config=run_config)
            self.wandb = wandb
        except ImportError:
            logger.info("log_config.use_wandb: wandb is not installed, the records go to %s only", self.path)

    def log(self, record, step=None):
        if not self.on:
            return
        import json
        self.f.write(json.dumps(dict(record, **({"_step": int(step)} if step is not None else {}))) + "\n")
        self.f.flush()
        if self.wandb is not None:
            self.wandb.log(record, step=step)

    def finish(self):
        if self.f is not None:
            self.f.close()
            self.f = None
        if self.wandb is not None:
            self.wandb.finish()


def _inline_batches(dataset):
    for raw in dataset:
        yield raw, dataset.collator(raw)


def train(engine, dataset, train_config, log_config, rank, world, eval_dataset=None, metrics=None):
    """Loop body of Multitask/utils/deepspeed_utils.py:190-246 (uneven-data join, forward, backward, step, logging) and
    the validation / save-on-improvement block behind it (:248-290).

    Deviation (stated): the loop runs AHEAD of the step -- batch i + 1 is fetched and collated before the forward of batch i, and
    with the reader thread (BatchReader, the default) up to ``depth`` + 2 batches are in flight.  The prompt draws of those batches
    therefore come before the draws a validation pass at step i makes; ``main`` gives each split its own ``random.Random`` so that
    the two never share a stream (a fixed seed reproduces a run, with or without validation), but the draws are not those of the
    reference's loop, whose DataLoader workers each own a forked copy of the global generator."""
    results = {}
    total_loss, total_acc, steps, utts = 0.0, 0.0, 0, 0
    best_val_loss, best_val_acc = float("inf"), 0.0
    val_loss, val_ppl, val_acc = [], [], []
    t0 = time.perf_counter()
    # num_workers_dataloader > 0 (the reference's recipe asks for 4 worker processes): one reader thread, see BatchReader;
    # 0 = read and collate on the training thread, like a DataLoader without workers
    threaded = int(getattr(train_config, "num_workers_dataloader", 0) or 0) > 0 and hasattr(dataset, "collator")
    epoch_rates = []
    metrics = metrics if metrics is not None else MetricsLog(log_config, rank)
    ga = max(1, int(getattr(engine, "ga", 1)))
    dynamic = train_config.batching_strategy == "dynamic"
    for epoch in range(train_config.num_epochs):
        engine.train()
        ep_loss, ep_acc, ep_n = 0.0, 0.0, 0
        reader = BatchReader(dataset, engine.core.device) if threaded else None
        it = iter(reader) if reader is not None else _inline_batches(dataset)
        epoch_step = 0                                         # the reference's `step + 1` (per epoch, :190, :248)
        e_t0, e_utts = time.perf_counter(), 0
        nxt, nxt_batch = next(it, (None, None))
        while True:
            raw, batch = nxt, nxt_batch
            if not engine.all_have_data(raw is not None):      # replaces deepspeed_join's gloo monitored_barrier
                break
            nxt, nxt_batch = next(it, (None, None))            # one batch of lookahead: its frozen encoder pass runs on a side
                                                               # stream under this batch's decoder step
            outputs, acc = engine(**batch)
            val_now = eval_dataset is not None and train_config.run_validation and \
                (epoch_step + 1) % train_config.validation_interval == 0
            if nxt_batch is not None and not val_now:          # (a validation pass would run in between and discard the prefetched
                engine.prefetch(**nxt_batch)                   #  encoder output: the pass would run twice for that batch)
            loss = outputs.loss
            engine.backward(loss)
            engine.step()
            steps += 1
            epoch_step += 1
            utts += batch["input_ids"].shape[0]
            e_utts += batch["input_ids"].shape[0]
            log_now = steps % max(1, log_config.log_interval) == 0
            if log_now and val_now:
                # loss / acc are views into the step's result buffer, which the validation forwards below overwrite:
                # read them first
                loss, acc = float(loss.detach()), float(acc)
            if val_now:
                ppl, el, ea = evaluation(engine, train_config, eval_dataset, rank, world)
                if train_config.save_model and (el < best_val_loss or ea > best_val_acc) and \
                        not str(train_config.output_dir).startswith("PATH/"):
                    # checkpoint_handler.py:169-182 naming: <output_dir>/<model_name>_epoch_E_step_S/
                    d = os.path.join(train_config.output_dir, f"{train_config.model_name}_epoch_{epoch + 1}_step_{epoch_step}")
                    if rank == 0:
                        os.makedirs(d, exist_ok=True)
                    engine.save_checkpoint(os.path.join(d, "pytorch_model.bin"))
                best_val_loss, best_val_acc = min(best_val_loss, el), max(best_val_acc, ea)
                val_loss.append(el)
                val_ppl.append(ppl)
                val_acc.append(ea)
                metrics.log({"valid/val_epoch_loss": el, "valid/val_perplexity": ppl, "valid/best_val_loss": best_val_loss,
                             "valid/val_accuracy": ea, "valid/val_best_accuracy": best_val_acc})
            if log_now:
                l, a = float(loss.detach() if torch.is_tensor(loss) else loss), float(acc)                 # the only host sync of the loop, every log_interval steps
                total_loss, total_acc = total_loss + l, total_acc + a
                ep_loss, ep_acc, ep_n = ep_loss + l / ga, ep_acc + a / ga, ep_n + 1
                # (the reference's step index: `step + 1` under dynamic batching, else epoch * total_length + step: :213-230)
                metrics.log({"train_inner/train_inner_loss": l / ga, "train_inner/train_inner_accuracy": a / ga},
                            step=epoch_step if dynamic else steps - 1)
                if rank == 0:
                    logger.info("epoch %d step %d loss %.4f acc %.4f lr %.3e  %.1f utt/s", epoch + 1, steps, l, a,
                                engine.get_lr()[0], world * utts / (time.perf_counter() - t0))
        if reader is not None:
            reader.close()
        if engine.core.device.type == "cuda":
            torch.cuda.synchronize()
        epoch_rates.append(world * e_utts / max(time.perf_counter() - e_t0, 1e-9))
        if ep_n:
            import math
            el_, ea_ = engine.reduce_scalars(ep_loss / ep_n, ep_acc / ep_n)
            el_, ea_ = el_ / world, ea_ / world
            metrics.log({"train/train_perplexity": math.exp(min(el_, 80.0)), "train/train_epoch_loss": el_, "train/train_epoch_acc": ea_})
    results["epoch_utterances_per_s"] = epoch_rates            # (per epoch, device drained: the first epoch carries the warm-up)
    n_logged = max(1, steps // max(1, log_config.log_interval))
    sl, sa = engine.reduce_scalars(total_loss / n_logged, total_acc / n_logged)
    results["avg_train_loss"], results["avg_train_acc"] = sl / world, sa / world
    results["steps"], results["utterances_per_s"] = steps, world * utts / max(time.perf_counter() - t0, 1e-9)
    if val_loss:
        results["avg_eval_prep"] = sum(val_ppl) / len(val_ppl)
        results["avg_eval_loss"] = sum(val_loss) / len(val_loss)
        results["avg_eval_acc"] = sum(val_acc) / len(val_acc)
    return results


def main(argv=None):
    ensure_hw_queues()                   # (before the first HIP call: ps_slm_amd/streams.py)
    cfg = parse_args(sys.argv[1:] if argv is None else argv)
    train_config, model_config, log_config, dataset_config = cfg.train_config, cfg.model_config, cfg.log_config, cfg.dataset_config
    logging.basicConfig(level=logging.INFO, format="[%(asctime)s][%(name)s][%(levelname)s] - %(message)s")
    torch.manual_seed(train_config.seed)
    random.seed(train_config.seed)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
    model_factory = get_custom_model_factory(model_config)
    model, tokenizer = model_factory(train_config, model_config, ckpt_path=cfg.ckpt_path, metric=cfg.metric,
                                     device=f"cuda:{local_rank}", keep_logits=False)
    ds_cfg = load_ds_config(cfg.deepspeed_config) if cfg.deepspeed_config else load_ds_config(DEFAULT_DS_CONFIG)
    engine = TasuEngine(model, ds_cfg)
    # hipGraph replay of the forward/backward launch sequences pays when batch shapes repeat (fixed-length data, no CPS
    # drops); with dynamic batching or CPS drops every batch has its own shape, so it is off unless asked for
    model.core.use_graphs = str(cfg.get("use_graphs", False)).lower() in ("1", "true", "yes")
    if model.core.use_graphs and str(cfg.get("graph_buckets", "16,8,256")).lower() not in ("", "none", "0"):
        # shapes of real batches rarely repeat exactly: pad them to buckets (token columns, posterior rows, labelled rows) so that
        # the LRU of captured step graphs gets hits (TasuModel.shape_buckets)
        model.core.shape_buckets = tuple(int(v) for v in str(cfg.get("graph_buckets", "16,8,256")).split(","))
    dataset = get_dataset(dataset_config, tokenizer, "train", model.core.geo, rank,
                          steps=int(cfg.get("synthetic_steps", 20)), batch_size=int(cfg.get("synthetic_batch", 16)))
    eval_dataset = None
    if train_config.run_validation:
        eval_dataset = get_dataset(dataset_config, tokenizer, "val", model.core.geo, rank,
                                   steps=int(cfg.get("synthetic_eval_steps", 2)), batch_size=int(cfg.get("synthetic_batch", 16)))
    # every split draws its prompts from its OWN generator: the reader thread (training split) and a validation pass (main thread)
    # would otherwise interleave draws on the global stream in a timing-dependent order
    for k, ds in enumerate((dataset, eval_dataset)):
        inner = getattr(ds, "dp", ds)                  # (MultiTaskDynamicBatchDataset wraps the sample dataset)
        if inner is not None and hasattr(inner, "rng"):
            inner.rng = random.Random(int(train_config.seed) * 1000003 + 7919 * rank + k)
    metrics = MetricsLog(log_config, rank, run_config={"train_config": vars(train_config), "model_config": vars(model_config),
                                                         "log_config": vars(log_config)})
    try:
        results = train(engine, dataset, train_config, log_config, rank, world, eval_dataset, metrics=metrics)
    finally:
        metrics.finish()
    if rank == 0:
        for k, v in results.items():
            logger.info("Key: %s, Value: %s", k, v)
        if train_config.save_model and train_config.output_dir and not train_config.output_dir.startswith("PATH/"):
            os.makedirs(train_config.output_dir, exist_ok=True)
    if train_config.save_model and train_config.output_dir and not train_config.output_dir.startswith("PATH/"):
        engine.save_checkpoint(os.path.join(train_config.output_dir, "pytorch_model.bin"))
    engine.destroy()                                    # the RCCL communicator goes before the process group
    if world > 1:
        dist.destroy_process_group()
    return results


if __name__ == "__main__":
    main()

"""The engine's gradient exchange on hardware: a 1-rank ``nccl`` (= RCCL) process group on the MI355X, so that the side-stream
event chaining, the per-range RCCL all-reduces, the stream waits of step() and the chunked AdamW all execute on the GPU
(the N > 1 semantics -- rank-averaged gradients, identical replicas -- are covered over gloo in tests/test_engine_cpu.py;
a one-GPU box cannot host a second RCCL rank)."""
import os

import pytest
import torch
import torch.distributed as dist

from ps_slm_amd.config import DEFAULT_DS_CONFIG, ModelConfig, TrainConfig, load_ds_config
from ps_slm_amd.engine import TasuEngine
from ps_slm_amd.ps_slm import model_factory
from ps_slm_amd.synthetic import synthetic_text_batch
from conftest import free_port

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_group():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    yield
    dist.destroy_process_group()


def build(force, chunks, graphs):
    tc = TrainConfig(freeze_llm=True, freeze_encoder=True, gt_emb=True, gt_emb_noise=False, ctc_posterior=True, do_psd=True,
                     use_fp16=True)
    mc = ModelConfig(llm_path="synthetic:mid", encoder_projector="linear-silu", llm_dim=256)
    model, _ = model_factory(tc, mc, device="cuda:0", init_seed=1234, keep_logits=False)
    model.core.use_graphs = graphs
    cfg = load_ds_config(DEFAULT_DS_CONFIG)
    cfg["lr"] = 1e-3
    eng = TasuEngine(model, cfg, force_exchange=force, w1_chunks=chunks)
    eng.sched_iter = 10                                   # past DeepSpeed's two zero-lr steps
    return model, eng


def run_steps(model, eng, n=4):
    raw = synthetic_text_batch(model.core.geo, 3, seed=5, prompt_len=9, n_audio=21, target_len=17, speech_pos=4, feat_frames=8,
                               noise=False, ragged=True)
    call = dict(input_ids=raw["input_ids"], attention_mask=raw["attention_mask"], labels=raw["labels"], input_features=None,
                input_feature_length=None, GT=[" ".join(map(str, p)) for p in raw["post_ids"]])
    losses = []
    for _ in range(n):
        out, _ = eng(**call)
        eng.backward(out.loss)
        eng.step()
        losses.append(out.loss.clone())
    torch.cuda.synchronize()
    return torch.stack(losses).cpu(), model.core.proj.p.clone(), model.core.proj.m.clone()


@pytest.mark.parametrize("graphs", [False, True])
def test_chunked_rccl_exchange_is_bit_identical_to_the_single_bucket_step(nccl_group, graphs):
    """Four optimizer steps with the exchange forced on (6 RCCL all-reduces per step: tail, 4 row blocks of the Linear1
    wgrad, LayerNorm parameters; AdamW per range) against the plain single-bucket step: losses, parameters and Adam moments
    must be bit-identical (a 1-rank sum is the identity, AdamW is elementwise, the row-block GEMMs keep every element's
    K order).  With graphs the decoder backward is replayed as a hipGraph and the projector tail stays eager."""
    m0, e0 = build(False, None, graphs)
    want = run_steps(m0, e0)
    m1, e1 = build(True, 4, graphs)
    assert e1.exchange and e1.w1_chunks == 4 and e1.comm_stream is not None
    assert e1.rccl is not None and e1.rccl.comm                 # the collective is RCCL through the C-ABI (tasu_allreduce_f32)
    e1.time_exchange = True
    got = run_steps(m1, e1)
    assert len(e1.exposed_events) == 4 * 6 and e1.exposed_ms() >= 0.0
    for a, b in zip(want, got):
        assert torch.equal(a, b)
    assert float(want[0][-1]) < float(want[0][0])         # and the steps do train


def test_side_stream_all_reduce_runs_under_the_remaining_wgrad_blocks(nccl_group):
    """The overlap the engine is built for, at the FULL projector (LayerNorm 25055 -> 2048 -> 1536: a 218 MB bucket, four
    51-MB row blocks of the Linear1 weight gradient) and a 16 x 104-row batch: the RCCL all-reduce of range i is issued on the
    side stream behind an event recorded after wgrad block i, while the compute stream goes on with block i + 1 -- persistent
    512-thread / 128-KiB-LDS GEMM workgroups on every CU.  Event timestamps must show every row block's all-reduce STARTING
    before the next block's GEMM has ended (the GEMM workgroups do not starve the collective's kernel), and the step must equal
    the plain single-bucket step bit for bit.  (One rank: RCCL still launches its kernel; N > 1 is the driver's to run.)"""
    def build_full(force):
        tc = TrainConfig(freeze_llm=True, freeze_encoder=True, gt_emb=True, gt_emb_noise=False, ctc_posterior=True, do_psd=True,
                         use_fp16=True)
        mc = ModelConfig(llm_path="synthetic:qwen2.5-1.5b", encoder_projector="linear-silu", llm_dim=1536, encoder_dim=25055)
        model, _ = model_factory(tc, mc, device="cuda:0", init_seed=77, keep_logits=False)
        cfg = load_ds_config(DEFAULT_DS_CONFIG)
        cfg["lr"] = 1e-3
        eng = TasuEngine(model, cfg, force_exchange=force, w1_chunks=4 if force else None)
        eng.sched_iter = 10
        return model, eng
    m0, e0 = build_full(False)
    geo = m0.core.geo
    assert geo.ctc_vocab == 25055 and m0.core.proj.numel > 54_000_000
    raw = synthetic_text_batch(geo, 16, seed=9, noise=False)
    call = dict(input_ids=raw["input_ids"], attention_mask=raw["attention_mask"], labels=raw["labels"], input_features=None,
                input_feature_length=None, GT=[" ".join(map(str, p)) for p in raw["post_ids"]])

    def steps(eng, model, n=3):
        for _ in range(n):
            out, _ = eng(**call)
            eng.backward(out.loss)
            eng.step()
        torch.cuda.synchronize()
        return model.core.proj.p.clone()
    want = steps(e0, m0)
    del m0, e0
    torch.cuda.empty_cache()
    m1, e1 = build_full(True)
    e1.trace_exchange = True
    got = steps(e1, m1)
    assert torch.equal(want, got)
    tr = e1.exchange_trace[-6:]                                   # the last step's six ranges, in issue order
    assert len(tr) == 6
    blocks = tr[1:5]                                              # the four row blocks of the Linear1 weight gradient
    assert all(hi - lo > 10_000_000 for lo, hi, *_ in blocks)     # ~12.8 M floats = 51 MB each
    for (lo, hi, issued, started, ended), nxt in zip(blocks[:-1], blocks[1:]):
        # nxt[2] is recorded on the compute stream right after the NEXT block's GEMM: the all-reduce of this block must have
        # started before that point, i.e. it ran under (or before) the next block's GEMM, not after it
        assert started.elapsed_time(nxt[2]) > 0.0, (lo, hi)
        assert issued.elapsed_time(started) < 5.0                 # and it started within milliseconds of becoming ready


def test_rccl_entry_points_of_the_c_abi():
    """tasu_comm_unique_id / tasu_comm_init / tasu_allreduce_f32 / tasu_allreduce_min_i32 / tasu_comm_destroy called directly (one
    rank: the sum over one rank is the identity, bit for bit), on a side stream with event chaining like the engine's."""
    import ctypes
    from ps_slm_amd import _lib
    lib = _lib.load()
    assert lib.tasu_comm_available() == 1
    ident = (ctypes.c_uint8 * 128)()
    assert lib.tasu_comm_unique_id(ident) == 0 and any(ident)
    torch.cuda.set_device(0)
    comm = ctypes.c_void_p()
    assert lib.tasu_comm_init(ident, 0, 1, ctypes.byref(comm)) == 0 and comm.value
    assert lib.tasu_comm_init(ident, 1, 1, ctypes.byref(ctypes.c_void_p())) != 0        # rank out of range
    x = torch.randn(1 << 20, device="cuda")
    want = x.clone()
    side = torch.cuda.Stream()
    ev = torch.cuda.Event()
    ev.record()
    side.wait_event(ev)
    assert lib.tasu_allreduce_f32(comm, x.data_ptr(), x.numel(), side.cuda_stream) == 0
    torch.cuda.current_stream().wait_stream(side)
    flag = torch.tensor([1], dtype=torch.int32, device="cuda")
    assert lib.tasu_allreduce_min_i32(comm, flag.data_ptr(), 1, torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    assert torch.equal(x, want) and int(flag) == 1
    assert lib.tasu_allreduce_f32(comm, None, 4, None) != 0
    assert lib.tasu_comm_destroy(comm) == 0


def _two_rank_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model, eng = build(False, None, graphs=True)
        assert eng.exchange and eng.world == 2 and eng.w1_chunks == 4
        raw = synthetic_text_batch(model.core.geo, 3, seed=50 + rank, prompt_len=9, n_audio=21, target_len=17, speech_pos=4,
                                   feat_frames=8, noise=False, ragged=True)
        call = dict(input_ids=raw["input_ids"], attention_mask=raw["attention_mask"], labels=raw["labels"], input_features=None,
                    input_feature_length=None, GT=[" ".join(map(str, p)) for p in raw["post_ids"]])
        eng.time_exchange = True
        losses, p1 = [], None
        for i in range(4):                                   # eager, captured, replayed, replayed (decoder backward as a graph)
            out, _ = eng(**call)
            eng.backward(out.loss)
            eng.step()
            losses.append(float(out.loss.detach()))
            if i == 0:
                p1 = model.core.proj.p.cpu().clone()         # the replica after the FIRST rank-averaged update
        torch.cuda.synchronize()
        ret[rank] = dict(losses=losses, p=model.core.proj.p.cpu().clone(), p1=p1, waits=len(eng.exposed_events))
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_keep_identical_replicas():
    """world size 2 on the GPU: two processes share the MI355X and exchange through gloo (RCCL refuses two ranks on one
    device), so that the N > 1 engine path -- hooks between the wgrad kernels, side-stream collectives on bucket ranges, stream
    waits, per-range AdamW with 1/world, decoder backward replayed as a hipGraph -- runs with real rank-averaged semantics on
    hardware: replicas stay bit-identical over 4 steps, both ranks train, and the update equals the single-process step on the
    averaged gradient."""
    import torch.multiprocessing as mp
    world, port = 2, free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_two_rank_worker, args=(world, port, ret), nprocs=world, join=True)
    r0, r1 = ret[0], ret[1]
    assert torch.equal(r0["p"], r1["p"]), "replicas diverged"
    assert r0["waits"] == 4 * 6 and r1["waits"] == 4 * 6
    assert r0["losses"][-1] < r0["losses"][0] and r1["losses"][-1] < r1["losses"][0]
    # single-process replay of the FIRST step: both ranks' gradients computed here, averaged by hand (sum, then the 1/world
    # the AdamW kernel applies), one AdamW step -- must EQUAL rank 0's replica after its first update, bit for bit (a two-term
    # fp32 sum is order-independent, AdamW is elementwise, the kernels are run-to-run deterministic)
    model, eng = build(False, None, graphs=False)
    grads, losses = [], []
    for rank in range(2):
        raw = synthetic_text_batch(model.core.geo, 3, seed=50 + rank, prompt_len=9, n_audio=21, target_len=17, speech_pos=4,
                                   feat_frames=8, noise=False, ragged=True)
        out, _ = eng(input_ids=raw["input_ids"], attention_mask=raw["attention_mask"], labels=raw["labels"], input_features=None,
                     input_feature_length=None, GT=[" ".join(map(str, p)) for p in raw["post_ids"]])
        eng.backward(out.loss)
        grads.append(model.core.proj.g.clone())
        losses.append(float(out.loss.detach()))
    assert abs(losses[0] - r0["losses"][0]) < 1e-6 and abs(losses[1] - r1["losses"][0]) < 1e-6
    assert not torch.equal(grads[0], grads[1])                # the ranks really saw different batches
    model.core.proj.g.copy_(grads[0] + grads[1])
    eng.world = 2                                             # grad_scale = 1/world inside the AdamW kernel: 0.5 * (g0 + g1)
    eng._last_state = None
    eng.step()
    torch.cuda.synchronize()
    assert torch.equal(model.core.proj.p.cpu(), r0["p1"]), "the 2-rank update is not the AdamW step on the averaged gradient"
    assert torch.equal(r0["p1"], r1["p1"])


@pytest.mark.parametrize("graphs", [False, True])
def test_lora_bucket_exchange_on_rccl_is_bit_identical(nccl_group, graphs):
    """use_peft=true through the same machinery: the adapters' range (complete when the decoder's backward -- with its
    side-stream weight-gradient chains joined -- is) exchanged first, then the projector's ranges; losses, the whole bucket and
    Adam moments bit-identical to the step without exchange (no dropout: the mask counter is not part of this comparison)."""
    def build_lora(force, chunks):
        tc = TrainConfig(freeze_llm=True, freeze_encoder=True, gt_emb=True, gt_emb_noise=False, ctc_posterior=True, do_psd=True,
                         use_fp16=True, use_peft=True)
        tc.peft_config.r, tc.peft_config.lora_dropout = 16, 0.0
        mc = ModelConfig(llm_path="synthetic:mid", encoder_projector="linear-silu", llm_dim=256)
        model, _ = model_factory(tc, mc, device="cuda:0", init_seed=1234, keep_logits=False)
        model.core.use_graphs = graphs
        cfg = load_ds_config(DEFAULT_DS_CONFIG)
        cfg["lr"] = 1e-3
        eng = TasuEngine(model, cfg, force_exchange=force, w1_chunks=chunks)
        eng.sched_iter = 10
        return model, eng
    m0, e0 = build_lora(False, None)
    want = run_steps(m0, e0)
    m1, e1 = build_lora(True, 4)
    lp = m1.core.lora
    seen = []
    issue = e1._issue
    e1._issue = lambda g, lo, hi: (seen.append((lo, hi)), issue(g, lo, hi))[1]
    got = run_steps(m1, e1)
    assert seen[0] == lp.layer_range[1] and seen[1] == lp.layer_range[0] and len(seen) == 4 * 8   # the adapters per span of layers, then the projector's six ranges
    for a, b in zip(want, got):
        assert torch.equal(a, b)
    assert float(m1.core.proj.p[lp.base:].abs().max()) > 0 and float(want[0][-1]) < float(want[0][0])


def test_side_streams_run_next_to_the_main_stream():
    """ps_slm_amd/streams.py: HIP maps streams onto a few hardware queues in creation order, and every fourth pooled torch stream
    shares the default stream's queue (strictly serial execution).  side_stream() probes with spin kernels and DEVICE timestamps
    (round 5: no host clock in the verdict): what it hands out overlaps with the current stream and with the side streams handed
    out before -- whatever the process created earlier -- every caller gets its own stream object, and the verdicts are reported."""
    import ps_slm_amd.streams as streams
    from ps_slm_amd.streams import report, runs_concurrently, side_stream
    streams._state.clear()                                         # (streams handed to earlier tests of this process are gone with their owners)
    junk = [torch.cuda.Stream() for _ in range(5)]                 # shift torch's round-robin pool position
    main = torch.cuda.current_stream()
    before = len(report("cuda"))
    a, b, c = side_stream("cuda", "test a"), side_stream("cuda", "test b"), side_stream("cuda", "test c")
    assert a is not b and b is not c and a is not c
    assert runs_concurrently(main, a) and runs_concurrently(main, b) and runs_concurrently(a, b)
    assert runs_concurrently(main, c) and runs_concurrently(a, c) and runs_concurrently(b, c)   # main + three roles: four queues
    assert not runs_concurrently(a, a)                             # the probe itself: one queue = one after the other
    for _ in range(5):                                             # the verdict does not flicker (device timestamps, not the host clock)
        assert runs_concurrently(main, a) and not runs_concurrently(b, b)
    rep = report("cuda")
    assert len(rep) == before + 3 and [r["verdict"] for r in rep if r.get("role", "").startswith("test")] == ["own queue"] * 3
    assert side_stream("cpu") is None
    # round 6: a role has ONE stream per process (every new stream shifts HIP's queue placement of the later ones)
    assert side_stream("cuda", "test a") is a and len(report("cuda")) == before + 4
    streams.release(a)                                             # (one holder's entry; the other stays)
    assert len(report("cuda")) == before + 3
    # a dead owner's stream stops counting (no probe against it, not in the report); release() does the same explicitly
    class Owner:
        pass
    o1, o2 = Owner(), Owner()
    s1, s2 = side_stream("cuda", "owned 1", owner=o1), side_stream("cuda", "owned 2", owner=o2)
    roles = lambda: [r.get("role") for r in report("cuda") if "role" in r]
    assert roles()[-2:] == ["owned 1", "owned 2"]
    del o1
    import gc
    gc.collect()
    assert "owned 1" not in roles() and "owned 2" in roles()
    streams.release(s2)
    assert "owned 2" not in roles() and roles()[:3] == ["test a", "test b", "test c"]
    del junk, s1


@pytest.mark.parametrize("graphs", [False, True])
def test_reference_loop_body_trains_the_plugin_through_autograd_on_gpu(graphs):
    """GPU twin of tests/test_engine_cpu.py::test_reference_loop_body_trains_the_plugin_through_autograd (SURVEY 8b "Autograd glue",
    VERDICT r5 item 6): ``torch.optim.AdamW`` over ``model.parameters()`` + ``outputs.loss.backward()`` -- the reference's loop body,
    Multitask/utils/deepspeed_utils.py:205-236 -- against TasuEngine (hand-scheduled backward, fused AdamW kernel) on a twin model,
    eager and with the step replayed as hipGraphs.  Gradients are bit-identical (the same kernels wrote the same bucket); the
    masters agree to fp32 rounding of the two AdamW formulations (torch's foreach vs the fused kernel), the losses step for step."""
    m_e, eng = build(False, None, graphs)
    m_a, _ = build(False, None, graphs)
    params = [p for p in m_a.parameters() if p.requires_grad]
    assert len(params) == 6 and all(p.is_leaf and p.is_cuda for p in params)
    c = eng.cfg
    opt = torch.optim.AdamW(params, lr=c["lr"], betas=tuple(c["betas"]), eps=c["eps"], weight_decay=c["weight_decay"])
    raw = synthetic_text_batch(m_e.core.geo, 3, seed=5, prompt_len=9, n_audio=21, target_len=17, speech_pos=4, feat_frames=8,
                               noise=False, ragged=True)
    call = dict(input_ids=raw["input_ids"], attention_mask=raw["attention_mask"], labels=raw["labels"], input_features=None,
                input_feature_length=None, GT=[" ".join(map(str, p)) for p in raw["post_ids"]])
    first = None
    for step in range(5):
        out_e, _ = eng(**call)
        lr = eng.get_lr()[0]
        eng.backward(out_e.loss)
        g_e = m_e.core.proj.g.clone()
        eng.step()
        for g in opt.param_groups:
            g["lr"] = lr
        out, _ = m_a(**call)
        assert out.loss.requires_grad and out.loss.grad_fn is not None
        opt.zero_grad()
        (out.loss / 2 * 2).backward()
        torch.cuda.synchronize()
        le, la = float(out_e.loss.detach()), float(out.loss.detach())
        assert abs(le - la) <= 2e-4 * abs(le), (step, le, la)
        if step == 0:
            assert torch.equal(m_a.core.proj.g, g_e)              # same weights, same kernels: the same bucket
            first = le
        for (n, p), (_, gv) in zip(m_a.named_parameters(), m_a._trainable_views(m_a.core.proj.g)):
            assert torch.equal(p.grad, gv), n
        opt.step()
        torch.cuda.synchronize()
        rel = (m_a.core.proj.p - m_e.core.proj.p).abs().max() / m_e.core.proj.p.abs().max()
        assert float(rel) < 1e-4, (step, float(rel))
    assert le < first                                             # and the steps do train

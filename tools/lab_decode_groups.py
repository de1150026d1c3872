"""Lab: beam-4 decode of 16 utterances as G independent groups (16 / G utterances each) running concurrently on G streams (one host
thread per group, weights shared, workspaces / KV caches / step graphs per group) against the one-group loop that ships.
Utterances are independent in generate() (ps-slm.py:660-675): results per utterance do not depend on the grouping."""
import os, sys, time, json, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ps_slm_amd.decode import beam_search_generate
from ps_slm_amd.model import Geometry, TasuModel
from ps_slm_amd.ops import HipOps
from ps_slm_amd.synthetic import synthetic_text_batch

geo = Geometry.qwen25_1p5b()
core = TasuModel(geo, HipOps(), "cuda", keep_logits=False)
core.init_random(1234)
B, NEW = 16, 200
raw = synthetic_text_batch(geo, B, seed=1234, noise=False)
ids = raw["input_ids"][:, :25]
am = torch.ones_like(ids, dtype=torch.bool)

def run(m, lo, hi):
    st = m.prepare_text(ids[lo:hi], am[lo:hi], None, raw["post_ids"][lo:hi], None, None)
    m.forward_projector_text(st)
    return beam_search_generate(m, st, num_beams=4, max_new_tokens=NEW, eos_token_id=-1, pad_token_id=0)

def view():
    o = HipOps()
    o._frag = core.ops._frag                            # the fragment-order weight copies are registered per operator object
    m = TasuModel(geo, o, "cuda", keep_logits=False)
    m.llm, m.proj = core.llm, core.proj
    return m

res = {}
run(core, 0, B); torch.cuda.synchronize()
t0 = time.perf_counter(); ref = run(core, 0, B); torch.cuda.synchronize()
dt = time.perf_counter() - t0
res["one_group"] = {"ms_per_position": round(dt / NEW * 1e3, 3), "tok_s": round(B * NEW / dt, 1)}
for G in (2, 4):
    n = B // G
    views = [view() for _ in range(G)]
    streams = [torch.cuda.Stream() for _ in range(G)]
    outs = [None] * G
    for g in range(G):                                   # warm-up (eager + graph capture) one group at a time, on its stream
        with torch.cuda.stream(streams[g]):
            run(views[g], g * n, (g + 1) * n)
        torch.cuda.synchronize()
    bar = threading.Barrier(G + 1)
    def worker(g):
        torch.cuda.set_device(0)
        with torch.cuda.stream(streams[g]):
            bar.wait()
            outs[g] = run(views[g], g * n, (g + 1) * n)
            streams[g].synchronize()
    ths = [threading.Thread(target=worker, args=(g,)) for g in range(G)]
    for t in ths: t.start()
    t0 = time.perf_counter(); bar.wait()
    for t in ths: t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    same = all(torch.equal(outs[g], ref[g * n:(g + 1) * n]) for g in range(G))
    res[f"{G}_groups"] = {"ms_per_position": round(dt / NEW * 1e3, 3), "tok_s": round(B * NEW / dt, 1), "tokens_equal_one_group": same}
    del views
print(json.dumps(res))

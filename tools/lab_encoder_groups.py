"""Lab: the frozen encoder pass of 16 utterances (500 frames) as G concurrent groups of 16 / G utterances on G streams (weights
shared, workspaces / graphs per group) against one pass over the batch."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ps_slm_amd.encoder import EncoderWeights, encoder_posterior
from ps_slm_amd.model import Geometry, TasuModel
from ps_slm_amd.ops import HipOps
from ps_slm_amd.synthetic import synthetic_text_batch

geo = Geometry.qwen25_1p5b()
enc = EncoderWeights(geo, torch.device("cuda"))
enc.init_random(4323)
B = 16
raw = synthetic_text_batch(geo, B, seed=1234, noise=False)
feats, lens = raw["input_features"].cuda(), raw["input_feature_length"]

class View:                                  # the slice of TasuModel the encoder pass uses
    def __init__(self):
        self.m = TasuModel.__new__(TasuModel)
def view():
    m = TasuModel.__new__(TasuModel)
    m.geo, m.ops, m.device, m.encoder = geo, HipOps(), torch.device("cuda"), enc
    m._ws, m._buf_gen, m._graphs, m._graph_seen, m.use_graphs, m.graph_cache_size = {}, 0, {}, {}, True, 64
    return m

res = {}
for G in (1, 2, 4):
    n = B // G
    views = [view() for _ in range(G)]
    streams = [torch.cuda.Stream() for _ in range(G)]
    def one():
        cur = torch.cuda.current_stream()
        for g in range(G):
            streams[g].wait_stream(cur)
            with torch.cuda.stream(streams[g]):
                encoder_posterior(views[g], feats[g * n:(g + 1) * n], lens[g * n:(g + 1) * n], want_post=False)
        for g in range(G):
            cur.wait_stream(streams[g])
    for g in range(G):                       # eager, capture: one group at a time
        for _ in range(3):
            with torch.cuda.stream(streams[g]):
                encoder_posterior(views[g], feats[g * n:(g + 1) * n], lens[g * n:(g + 1) * n], want_post=False)
            torch.cuda.synchronize()
    for _ in range(3): one()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): one()
    torch.cuda.synchronize()
    res[f"{G}_groups_ms"] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
    del views
    torch.cuda.empty_cache()
print(json.dumps(res))

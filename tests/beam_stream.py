"""Drives a beam-update implementation (``ops.beam_update`` over ps_slm_amd.decode.DeviceBeam: the HIP kernel or the CPU
double) and the vectorised host restatement (ps_slm_amd.decode.BeamState) through the same synthetic score stream: logits
depend only on (step, last token), EOS is competitive so beams really finish, and a quantised table produces exact score ties
(tie-breaking is part of the contract: score, then beam, then token id)."""
import numpy as np
import torch

from ps_slm_amd.decode import BeamState, DeviceBeam
from ps_slm_amd.model import Geometry, TasuModel
from ps_slm_amd.synthetic import MID_GEOMETRY


def make_table(seed, V, eos, quantise):
    g = torch.Generator().manual_seed(seed)
    table = torch.randn(64, V, generator=g) * 2.0
    if quantise:
        table = (table * 2).round() / 2            # many equal logits -> exact ties in log-probs and in the running sums
    table[:, eos] += 1.5
    return table


def step_scores(table, last, t, ban, eos, K):
    lp = torch.log_softmax(table[(last * 7 + t) % 64], -1)
    if ban:
        lp[:, eos] = float("-inf")
    v, i = torch.sort(lp, dim=-1, descending=True, stable=True)
    return v[:, :K].contiguous(), i[:, :K].to(torch.int32).contiguous()


def run_host(table, B, nb, T, eos, lpw, min_len):
    state = BeamState(B, nb, T, eos, eos, lpw, min_len)
    while not state.done:
        t = state.cur
        seqs = torch.from_numpy(state.run_seq).view(B * nb, -1)
        last = seqs[:, t - 1] if t > 0 else torch.zeros(B * nb, dtype=torch.long)
        v, i = step_scores(table, last, t, state.ban_eos(), eos, 2 * nb)
        state.update(v.numpy().reshape(B, nb, -1), i.numpy().reshape(B, nb, -1).astype(np.int64))
    return state.result()


def run_device(ops, device, table, B, nb, T, eos, lpw, min_len, extra_steps=0):
    """Returns (tokens [B, n], number of update calls made).  ``extra_steps`` more calls are issued after the state reports
    done: they must leave it untouched (the decode loop runs a couple of positions ahead of its look at the done word)."""
    geo = Geometry.from_dict(dict(MID_GEOMETRY, llm_layers=0))
    model = TasuModel(geo, ops, device)
    dev = model.device
    bs = DeviceBeam(model, B, nb, T, eos, lpw, min_len, S=5, valid=[3 + b % 3 for b in range(B)])
    K = 2 * nb
    last = torch.zeros(B, dtype=torch.long)
    v, i = step_scores(table, last, 0, min_len > 0, eos, K)
    ops.beam_update(v.to(dev), i.to(dev), bs, True)
    calls, t = 1, 1
    while not int(bs.ctl.cpu()[1]):
        assert int(bs.ctl.cpu()[0]) == t
        ids = bs.next_ids.cpu().long()
        assert (bs.next_slot.cpu() == 5 + t - 1).all() and (bs.next_lens.cpu() == 5 + t).all()
        assert torch.equal(bs.next_pos.cpu().view(B, nb), (bs.valid.cpu().view(B, 1) + t - 1).expand(B, nb).to(torch.int32))
        assert (bs.next_src.cpu().view(B, nb) // nb == torch.arange(B)[:, None]).all()          # parents stay inside the utterance
        ban = int(bs.banned.cpu()[0]) == eos
        assert ban == (t < min_len)
        v, i = step_scores(table, ids, t, ban, eos, K)
        ops.beam_update(v.to(dev), i.to(dev), bs, False)
        calls += 1
        t += 1
    out = bs.result(eos).numpy()
    snap = [x.clone() for x in (bs.run_scores, bs.fin_scores, bs.fin_len, bs.fin_par, bs.fin_tok, bs.ctl)]
    for _ in range(extra_steps):
        ops.beam_update(v.to(dev), i.to(dev), bs, False)
    for a, b in zip(snap, (bs.run_scores, bs.fin_scores, bs.fin_len, bs.fin_par, bs.fin_tok, bs.ctl)):
        assert torch.equal(a, b)
    return out, calls

"""Host-side integer plan for ``_merge_input_ids_with_audio_features``
(reference: Multitask/model/ps-slm.py:679-873).  The reference builds these indices with a dozen tiny device
ops; here they are a few KB of numpy on the host (the ids arrive from the host anyway) and the device work is
two gather kernels (tasu_embed_merge_fwd / tasu_merge_bwd).  Same results, same ValueErrors.
"""
from dataclasses import dataclass

import numpy as np

IGNORE = -100


@dataclass
class MergePlan:
    B: int
    S: int
    Spad: int
    left_padding: bool
    src_kind: np.ndarray       # [B*S] int32: 0 pad, 1 token, 2 audio
    src_idx: np.ndarray        # [B*S] int32: token id | projector row (b*Lmax + t)
    key_mask: np.ndarray       # [B, Spad] uint8
    position_ids: np.ndarray   # [B*S] int32
    labels: np.ndarray         # [B, S] int64 or None (merged labels, -100 on audio/pad)
    shift_labels: np.ndarray   # [B*S] int32: the label row m must predict (-100 = ignored)
    audio_rows: np.ndarray     # [B*Lmax] int32: merged row of projector row r, -1 for projector padding rows
    count: int                 # number of non-ignored shifted labels


def build_merge_plan(input_ids, attention_mask, labels, num_audio, speech_id, Lmax) -> MergePlan:
    ids = np.asarray(input_ids, dtype=np.int64)
    am = np.asarray(attention_mask).astype(bool)
    na = np.asarray(num_audio, dtype=np.int64)
    B, L = ids.shape
    lp = bool((~am[:, 0]).any())
    rp = bool((~am[:, -1]).any())
    left = True                                                   # ps-slm.py:774-785
    if B > 1:
        if lp and rp:
            raise ValueError(f"both side of attention_mask has zero, invalid. {attention_mask}")
        left = not (rp and not lp)
    is_sp = ids == speech_id
    if int(is_sp.sum()) != B or not (is_sp.sum(1) == 1).all():
        # the reference indexes num_audio_tokens by the flattened <speech> mask (one per row, :805-806)
        raise ValueError("The input provided to the model are wrong: exactly one <speech> token per row is required")
    width = np.ones_like(ids)
    width[is_sp] = na
    new_pos = np.cumsum(width, -1) - 1
    tot = width.sum(-1)
    S = int(tot.max())
    if left:
        new_pos = new_pos + (S - 1 - new_pos[:, -1])[:, None]
    is_text = (~is_sp) & am
    npad = (~am).sum(-1)
    real = tot - npad
    col = np.arange(S)[None, :]
    live = (S - col) <= real[:, None] if left else col < real[:, None]
    kind = np.zeros((B, S), dtype=np.int32)
    idx = np.zeros((B, S), dtype=np.int32)
    bi, li = np.nonzero(is_text)
    di = new_pos[bi, li]
    kind[bi, di] = 1
    idx[bi, di] = ids[bi, li]
    audio_slot = live & (kind == 0)
    if int(audio_slot.sum()) != int(na.sum()):                   # ps-slm.py:861-865
        raise ValueError(
            f"The input provided to the model are wrong. The number of audio tokens is {is_sp.sum(-1)} while the "
            f"number of audio given to the model is {B}. This prevents correct indexing and breaks batch generation.")
    audio_rows = np.full(B * Lmax, -1, dtype=np.int32)
    for b in range(B):                                           # row-major fill order of :867-869
        cols = np.nonzero(audio_slot[b])[0]
        kind[b, cols] = 2
        idx[b, cols] = b * Lmax + np.arange(len(cols))
        audio_rows[b * Lmax: b * Lmax + len(cols)] = b * S + cols
    mask = kind != 0
    pos = np.where(mask, np.cumsum(mask, -1) - 1, 1).astype(np.int32)   # :871
    Spad = (S + 63) // 64 * 64
    key_mask = np.zeros((B, Spad), dtype=np.uint8)
    key_mask[:, :S] = mask
    lab = None
    shift = np.full((B, S), IGNORE, dtype=np.int32)
    count = 0
    if labels is not None:
        labs = np.asarray(labels, dtype=np.int64)
        lab = np.full((B, S), IGNORE, dtype=np.int64)
        lab[bi, di] = labs[bi, li]
        shift[:, :-1] = lab[:, 1:]
        count = int((shift != IGNORE).sum())
    return MergePlan(B=B, S=S, Spad=Spad, left_padding=left, src_kind=kind.reshape(-1), src_idx=idx.reshape(-1),
                     key_mask=key_mask, position_ids=pos.reshape(-1), labels=lab, shift_labels=shift.reshape(-1),
                     audio_rows=audio_rows, count=count)

"""TEST INFRASTRUCTURE ONLY -- tests/golden/mid_text_lora*.npz: the REAL reference model (imported from /root/reference through
oracle/ref_import.py) at the kernel-compatible "mid" geometry with the LoRA formula applied by hand to its HF decoder
(oracle/lora_oracle.py: peft itself is not available), fp32, one text-branch training step: loss, accuracy, sampled logit
columns, the projector's gradients and every adapter's gradient.

Run in the build container only:  python oracle/make_golden_lora.py
Fixtures are data (seeds + the reference's outputs); weights come from ps_slm_amd.synthetic (seeded), nothing is copied.
"""
import dataclasses
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.lora_oracle import apply_hand_lora  # noqa: E402
from oracle.make_golden import quiet, save  # noqa: E402
from oracle.ref_import import build_reference_model  # noqa: E402

CASES = {
    # name: (r, alpha, targets, dropout p, dropout (seed, step))
    "mid_text_lora": (16, 32, ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj"), 0.0, None),
    "mid_text_lora_qv": (64, 16, ("q_proj", "v_proj"), 0.0, None),
    "mid_text_lora_drop": (64, 16, ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj"), 0.25, (777, 1)),
}


def main():
    from ps_slm_amd.lora import LoraConfig
    from ps_slm_amd.model import Geometry
    from ps_slm_amd.synthetic import MID_GEOMETRY, random_lora_state_dict, random_state_dict, synthetic_text_batch

    geo = Geometry.from_dict(MID_GEOMETRY)
    gd = dataclasses.asdict(geo)
    seed_w, seed_b, seed_l = 2026, 31, 909
    sd = random_state_dict(geo, seed_w, with_encoder=True)
    batch = synthetic_text_batch(geo, 3, seed=seed_b, prompt_len=9, n_audio=21, target_len=17, speech_pos=4,
                                 feat_frames=12, noise=True, drop_prob=0.15, ragged=True)
    kept = [list(np.asarray(p)[np.asarray(k, dtype=bool)]) for p, k in zip(batch["post_ids"], batch["keeps"])]
    GT = [" ".join(map(str, k)) for k in kept]
    g = torch.Generator().manual_seed(5)
    cols = torch.randperm(geo.llm_vocab, generator=g)[:64].sort().values
    for name, (r, alpha, targets, p, rng) in CASES.items():
        cfg = LoraConfig(r=r, lora_alpha=alpha, lora_dropout=p, target_modules=targets)
        model = build_reference_model(gd, 0, dict(gt_emb=True, gt_emb_noise=False))
        missing, unexpected = model.load_state_dict(sd, strict=False)
        assert not unexpected and set(missing) <= {"llm.lm_head.weight"}
        lsd = random_lora_state_dict(geo, cfg, seed_l)
        lparams = apply_hand_lora(model.llm, lsd, targets, cfg.scaling, p, rng)
        out, acc = quiet(model, input_ids=batch["input_ids"], input_features=batch["input_features"],
                         attention_mask=batch["attention_mask"], input_feature_length=batch["input_feature_length"], GT=GT,
                         labels=batch["labels"])
        out.loss.backward()
        lg = out.logits.detach().float()
        arrs = dict(seed_w=seed_w, seed_b=seed_b, seed_l=seed_l, r=r, alpha=alpha, p=p, rng=np.asarray(rng if rng else (0, 0)),
                    targets=np.asarray(",".join(targets)), loss=out.loss.detach().float(), acc=torch.as_tensor(acc).float(),
                    cols=cols, logits_cols=lg[:, :, cols], lse=torch.logsumexp(lg, -1))
        for n, prm in model.encoder_projector.named_parameters():
            if n != "ffn.0.weight":
                arrs["grad." + n] = prm.grad.clone()
        for k, prm in lparams.items():
            g_ = prm.grad.clone()
            if r >= 64:
                g_ = g_[::2, ::2]                                  # r = 64 fixtures: every 2nd row / column (fixture size)
            arrs["lgrad." + k] = g_.half()                         # fp16 storage: compared by cosine / relative norm
        save(name, **arrs)
        print(name, "loss", float(out.loss), "acc", float(acc))

    # beam-4 generate() of the adapted model on the inputs of tests/golden/mid_generate_beam4.npz (text path: the clean pseudo-posterior
    # of the regex-cleaned targets), fp32, no dropout (eval): the decode loop with adapters in place
    from oracle.make_golden import npify  # noqa: F401
    zg = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "mid_generate_beam4.npz"))
    r, alpha, targets, _, _ = CASES["mid_text_lora"]
    cfg = LoraConfig(r=r, lora_alpha=alpha, lora_dropout=0.0, target_modules=targets)
    model = build_reference_model(gd, 0, dict(gt_emb=True, gt_emb_noise=False))
    model.load_state_dict(sd, strict=False)
    apply_hand_lora(model.llm, random_lora_state_dict(geo, cfg, seed_l), targets, cfg.scaling)
    words = ["ab cd ef gh", "x yz"]
    feats = torch.zeros(2, 12, geo.feat_dim)
    with torch.no_grad():
        toks = quiet(model.generate, input_ids=torch.from_numpy(zg["input_ids"]), input_features=feats,
                     attention_mask=torch.from_numpy(zg["attention_mask"]), input_feature_length=torch.tensor([12, 12]),
                     max_new_tokens=16, targets=words)
    word_ids = [model.encoder_tokenizer.encode(t) for t in words]
    assert np.array_equal(np.concatenate([np.asarray(w) for w in word_ids]), zg["post_ids_flat"])
    save("mid_generate_lora", seed_w=seed_w, seed_l=seed_l, r=r, alpha=alpha, targets=np.asarray(",".join(targets)), tokens_text=toks)
    print("mid_generate_lora", toks.tolist(), "base model:", zg["tokens_text"].tolist())


if __name__ == "__main__":
    main()

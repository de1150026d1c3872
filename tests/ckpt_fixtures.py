"""Writes the on-disk checkpoint layouts the reference recipe loads (no real checkpoint exists on the test boxes):
  <dir>/llm/         HF Qwen2 directory: config.json + model.safetensors            (ps-slm.py:92-97, AutoModelForCausalLM)
  <dir>/sensevoice/  funasr SenseVoiceSmall directory: config.yaml + model.pt       (ps-slm.py:91-106, funasr AutoModel)
  <dir>/projector.pt the trainable part under the reference's key names             (checkpoint_handler.py:169-182)
from a seeded reference-named state dict (ps_slm_amd.synthetic.random_state_dict)."""
import json

import torch
import yaml
from safetensors.torch import save_file


def write_checkpoint_dirs(tmp_path, geo, sd):
    hf, enc = tmp_path / "llm", tmp_path / "sensevoice"
    hf.mkdir()
    enc.mkdir()
    json.dump(dict(vocab_size=geo.llm_vocab, hidden_size=geo.llm_dim, intermediate_size=geo.llm_inter,
                   num_hidden_layers=geo.llm_layers, num_attention_heads=geo.llm_heads, num_key_value_heads=geo.llm_kv_heads,
                   head_dim=128, rope_theta=geo.rope_theta, rms_norm_eps=geo.rms_eps, tie_word_embeddings=bool(geo.tied)),
              open(hf / "config.json", "w"))
    save_file({k[4:]: v.contiguous() for k, v in sd.items() if k.startswith("llm.")}, str(hf / "model.safetensors"))
    yaml.safe_dump(dict(input_size=geo.feat_dim,
                        encoder_conf=dict(output_size=geo.enc_dim, attention_heads=geo.enc_heads, linear_units=geo.enc_ffn,
                                          num_blocks=geo.enc_blocks, tp_blocks=geo.enc_tp_blocks, kernel_size=geo.enc_kernel,
                                          sanm_shfit=0)), open(enc / "config.yaml", "w"))
    torch.save({k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}, enc / "model.pt")
    ckpt = tmp_path / "projector.pt"
    torch.save({k: v for k, v in sd.items() if k.startswith("encoder_projector.")}, ckpt)
    return hf, enc, ckpt

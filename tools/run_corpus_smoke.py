"""End-to-end smoke of the training entrypoint on a tiny generated wav-in-ark corpus at full Qwen2.5-1.5B / SenseVoiceSmall
geometry (random weights): jsonl -> ps_slm_amd/dataset.py -> HIP fbank/LFR -> SANM encoder -> CTC -> PSD -> projector -> LLM
-> backward -> AdamW.  Prompts are integer strings because the synthetic tokenizer maps "12 7" -> [12, 7]."""
import json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dataset_fixtures as fx
from ps_slm_amd.finetune_deepspeed import main

with tempfile.TemporaryDirectory() as root:
    dirs = fx.write_corpus(root, split_sizes=(("train", 13),))
    with open(os.path.join(root, "multiprompt.jsonl"), "w") as f:
        for task, prompt in (("ASR", "11 12 13"), ("ST", "21 22"), ("hotword", "31 32 33 34")):
            f.write(json.dumps({"task": task, "prompt": prompt}) + "\n")
    res = main(["++model_config.file=ps_slm_amd/ps_slm.py:model_factory", "++model_config.llm_path=synthetic:qwen2.5-1.5b",
                "++model_config.llm_dim=1536", "++model_config.encoder_dim=25055", "++model_config.encoder_projector=linear-silu",
                "++train_config.freeze_llm=true", "++train_config.freeze_encoder=true", "++train_config.gt_emb=false",
                "++train_config.ctc_posterior=true", "++train_config.do_psd=true", "++train_config.num_epochs=1", "++train_config.use_fp16=true",
                "++dataset_config.file=ps_slm_amd/dataset.py:get_speech_dataset", f"++dataset_config.train_scp_file_path={dirs['train']}",
                f"++dataset_config.multitask_prompt_path={root}/multiprompt.jsonl", "++dataset_config.prompt_style={} 151665",
                "++dataset_config.train_max_frame_length=40", "++dataset_config.ds_rate=8", "++metric=acc", "++log_config.log_interval=1"])
    print("RESULT", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in res.items()})

    # the decode recipe on the same corpus: inference entrypoint (left padding, keys / targets) -> scorer
    import io
    from ps_slm_amd import compute_cer
    from ps_slm_amd.inference_batch import main as decode_main
    test = fx.write_corpus(root, split_sizes=(("test", 5),))["test"]
    with open(os.path.join(root, "multiprompt.jsonl"), "w") as f:
        for task, prompt in (("ASR", "11 12 13"), ("ST", "21 22"), ("hotword", "31 32 33 34")):
            f.write(json.dumps({"task": task, "prompt": prompt}) + "\n")
    pred, gt = decode_main(["++model_config.file=ps_slm_amd/ps_slm.py:model_factory", "++model_config.llm_path=synthetic:qwen2.5-1.5b",
                            "++model_config.llm_dim=1536", "++model_config.encoder_dim=25055", "++model_config.encoder_projector=linear-silu",
                            "++train_config.freeze_llm=true", "++train_config.gt_emb=false", "++train_config.ctc_posterior=true",
                            "++train_config.do_psd=true", "++dataset_config.file=ps_slm_amd/dataset.py:get_speech_dataset",
                            f"++dataset_config.test_scp_file_path={test}", f"++dataset_config.multitask_prompt_path={root}/multiprompt.jsonl",
                            "++dataset_config.prompt_style={} 151665", "++dataset_config.inference_mode=true",
                            "++dataset_config.eval_max_frame_length=60", "++dataset_config.ds_rate=8", "++max_new_tokens=12",
                            f"++decode_log={root}/decode_log"])
    print(open(pred).read().strip().splitlines()[:3])
    report = io.StringIO()
    total, per = compute_cer.score(gt, pred, tochar=True, out=report)
    print("SCORER", report.getvalue().strip().splitlines()[-1], "utterances", len(per))

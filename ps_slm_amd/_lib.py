"""ctypes binding of libtasu_hip.so.  Fails loudly: there is NO fallback path."""
import ctypes as C
import os

# torch bundles its own ROCm runtime (libamdhip64.so inside torch/lib).  It MUST be loaded before libtasu_hip.so so
# that both share ONE HIP runtime: loading ours first binds it to /opt/rocm's copy and every launch on a torch stream
# then fails (observed on the GPU box as "launch failure" from the first kernel).
import torch  # noqa: F401  (load order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TASU_LIB_PATH") or os.path.join(_HERE, "libtasu_hip.so")     # (override: instrumented debug builds)

vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> argtypes (all return int); mirrors include/tasu_hip.h one to one
PROTOTYPES = {
    "tasu_abi_version": [],
    "tasu_gemm_nt_bf16": [vp, i32, vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, vp],
    "tasu_gemm_nt_bf16_ws": [vp, i32, vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, vp, i64, vp],
    "tasu_gemm_bias_relu_bf16": [vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, vp, i64, vp],
    "tasu_gemm_gate_up_swiglu": [vp, i32, vp, i32, vp, vp, i32, i32, i32, vp],
    "tasu_gemm_nt_bf16_kernel": [vp, i32, vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp],
    "tasu_gemm_skinny_bf16": [vp, i32, vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, vp, i64, vp],
    "tasu_transpose_bf16": [vp, i32, vp, i32, i32, i32, i32, i32, vp],
    "tasu_cast_f32_bf16": [vp, vp, i64, vp],
    "tasu_rmsnorm_fwd": [vp, vp, vp, vp, i32, i32, f32, vp],
    "tasu_rmsnorm_bwd": [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "tasu_rmsnorm_fwd_rows": [vp, vp, vp, vp, vp, i32, i32, f32, vp],
    "tasu_rmsnorm_bwd_rows": [vp, vp, vp, vp, vp, vp, vp, i32, i32, vp],
    "tasu_rmsnorm_bwd_rows_resid": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp],
    "tasu_rope_table": [vp, vp, vp, i32, i32, f32, vp],
    "tasu_rope_fwd": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "tasu_rope_bwd": [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "tasu_attn_fwd": [vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp],
    "tasu_attn_bwd_prep": [vp, vp, vp, vp, i32, i32, i32, vp],
    "tasu_attn_bwd_dq": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp],
    "tasu_attn_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp],
    "tasu_attn_gqa_supported": [i32, i32, i32],
    "tasu_attn_bwd_rope": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, i32, vp],
    "tasu_attn_sp_supported": [i32, i32, i32],
    "tasu_attn_fwd_kernel": [vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, i32, vp],
    "tasu_attn_bwd_fused": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, i32, vp],
    "tasu_attn_bwd_dkv": [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp],
    "tasu_swiglu_fwd": [vp, vp, i32, i32, vp],
    "tasu_swiglu_bwd": [vp, vp, vp, i32, i32, vp],
    "tasu_silu_fwd": [vp, vp, i64, vp],
    "tasu_silu_bwd": [vp, vp, vp, i64, vp],
    "tasu_relu_bwd": [vp, vp, vp, i64, vp],
    "tasu_relu_fwd": [vp, vp, i64, vp],
    "tasu_gemm_nt_rank": [vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, i32, vp],
    "tasu_gemm_nt_rank_group": [i32, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp],
    "tasu_lora_apply_group": [vp, i32, i32, vp, i32, vp, i32, vp, i32, i32, i32, f32, f32, vp, vp],
    "tasu_lora_dropout_norm_group": [vp, vp, vp, i32, vp, vp, i32, i32, f32, vp, vp],
    "tasu_gemm_tn_rank": [vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, vp],
    "tasu_scale_bf16": [vp, vp, f32, i64, vp],
    "tasu_lora_apply": [vp, i32, vp, i32, vp, i32, i32, i32, i32, f32, f32, vp, i32, vp, vp, i32, vp],
    "tasu_lora_dropout": [vp, i32, vp, i32, i32, i32, f32, vp, i32, vp],
    "tasu_lora_dropout_norm": [vp, vp, vp, vp, i32, i32, f32, vp, i32, vp],
    "tasu_rng_advance": [vp, vp],
    "tasu_rmsnorm_fwd_ld": [vp, vp, vp, i32, vp, i32, i32, f32, vp],
    "tasu_gemm_gate_up_swiglu_ld": [vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, vp, i64, vp],
    "tasu_copy_rows_bf16": [vp, i32, vp, i32, i32, i32, vp],
    "tasu_lora_refresh": [vp, vp, i32, i32, vp],
    "tasu_ce_fwd_bwd": [vp, i32, vp, i32, i32, vp, vp, vp, vp, vp, vp],
    "tasu_ce_reduce": [vp, vp, vp, i32, vp, vp],
    "tasu_layernorm_fwd": [vp, i32, vp, vp, vp, i32, i32, vp, vp, i32, i32, f32, vp],
    "tasu_layernorm_bwd_params": [vp, i32, vp, i32, vp, vp, vp, vp, vp, i32, i32, vp],
    "tasu_colsum_bf16": [vp, i32, vp, i32, i32, vp],
    "tasu_posterior_build": [vp, vp, vp, i32, i32, i32, vp],
    "tasu_embed_merge_fwd": [vp, vp, vp, vp, vp, i32, i32, vp],
    "tasu_merge_bwd": [vp, vp, vp, i32, i32, vp],
    "tasu_adamw": [vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, f32, vp],
    "tasu_sinusoid_pe": [vp, vp, i32, i32, i32, f32, vp],
    "tasu_fsmn_fwd": [vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "tasu_fsmn_ln_fwd": [vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, vp],
    "tasu_softmax_rows": [vp, i32, i32, vp, i32, i32, i32, vp],
    "tasu_psd_frame_stats": [vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "tasu_psd_plan": [vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, vp],
    "tasu_psd_gather": [vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "tasu_psd_logit_stats": [vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "tasu_psd_gather_softmax": [vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "tasu_kv_fill": [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "tasu_kv_append": [vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "tasu_gemm_skinny_norm": [vp, i32, vp, i32, vp, vp, i32, i32, i32, vp, vp, f32, i32, vp, i64, vp],
    "tasu_gemm_skinny_qkv_rope": [vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, i32, vp, i64, vp],
    "tasu_gemm_skinny_swiglu": [vp, i32, vp, i32, vp, i32, i32, i32, i32, vp, i64, vp],
    "tasu_rope_append": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "tasu_kv_index_init": [vp, i32, i32, i32, i32, vp],
    "tasu_kv_index_reorder": [vp, vp, vp, vp, i32, i32, vp],
    "tasu_attn_decode": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp],
    "tasu_logprob_topk": [vp, i32, i32, i32, i32, vp, i32, vp, vp, vp, i64, vp],
    "tasu_stream_supported": [i32, i32],
    "tasu_gemm_stream_bf16": [vp, i32, vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "tasu_gemm_stream_swiglu": [vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "tasu_gemm_stream_resid_prenorm": [vp, i32, vp, i32, vp, vp, i32, i32, i32, vp, vp, i32, vp, i32, i32, vp],
    "tasu_gemm_stream_swiglu_rstd": [vp, i32, vp, i32, vp, i32, i32, i32, i32, vp, i32, f32, i32, i32, i32, vp],
    "tasu_gemm_stream_qkv_rope": [vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "tasu_gemm_stream_slabs": [vp, i32, vp, i32, vp, i64, i32, i32, i32, i32, i32, i32, vp],
    "tasu_stream_finish_norm": [vp, i32, vp, vp, i32, i32, vp, vp, f32, i32, vp],
    "tasu_stream_finish_prenorm": [vp, i32, vp, vp, i32, i32, vp, vp, i32, vp, vp],
    "tasu_gemm_stream_qkv_rope_rstd": [vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, i32, vp, i32, f32, i32, i32, vp],
    "tasu_gemm_stream_norm": [vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, vp, i64, vp, vp, f32, i32, i32, i32, vp, vp],
    "tasu_rmsnorm_fwd_frag": [vp, vp, vp, i32, i32, f32, vp],
    "tasu_to_fragment_order": [vp, i32, vp, i32, i32, i32, i32, i32, vp],
    "tasu_gemm_nt_bf16_splitk": [vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, vp],
    "tasu_gemm_nt_bf16_slabs": [vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, vp],
    "tasu_sum_slabs_bf16": [vp, i32, i64, vp, i64, vp],
    "tasu_flac_info": [vp, i64, vp, vp],
    "tasu_flac_decode": [vp, i64, vp, i64, vp],
    "tasu_beam_update": [vp] * 21 + [i32] * 7 + [vp],
    "tasu_fbank": [vp, i64, f32, i32, i32, vp, vp, i32, f32, vp, vp],
    "tasu_lfr_cmvn": [vp, i32, i32, i32, i32, vp, vp, vp, vp],
    "tasu_embed_rows": [vp, vp, vp, i32, i32, vp],
    "tasu_decode_step_prologue": [vp, vp, vp, vp, vp, f32, vp, vp, vp, f32, vp, vp, vp, i32, i32, i32, i32, vp],
}
RESTYPE_I64 = {"tasu_gemm_launch_count"}
PROTOTYPES.update({
    "tasu_scale_softmax_rows_bf16": [vp, vp, vp, i32, i32, i32, f32, vp],
    "tasu_softmax_bwd_rows_bf16": [vp, vp, vp, vp, i32, i32, i32, f32, vp],
    "tasu_gemm_gate_up_swiglu_ws": [vp, i32, vp, i32, vp, vp, i32, i32, i32, vp, i64, vp],
    "tasu_gemm_qkv_rope": [vp, i32, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, vp, i64, vp],
    "tasu_gemm_dswiglu": [vp, i32, vp, i32, vp, vp, vp, i32, i32, i32, vp, i64, vp],
    "tasu_gemm_plan": [i32, i32, i32, i32, i32],
    "tasu_streamk_schedule": [i32, i32, i32, vp, vp, i32],
    "tasu_gemm_nt_bf16_streamk": [vp, i32, vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, vp, i64, vp],
    "tasu_comm_available": [],
    "tasu_comm_unique_id": [vp],
    "tasu_comm_init": [vp, i32, i32, vp],
    "tasu_comm_destroy": [vp],
    "tasu_comm_library": [vp, i32],
    "tasu_comm_count": [vp, vp],
    "tasu_allreduce_f32": [vp, vp, i64, vp],
    "tasu_allreduce_min_i32": [vp, vp, i64, vp],
    "tasu_gemm_launch_count": [],
    # fp32 arithmetic mode of the decode path (csrc/fp32.hip)
    "tasu_f32_gemm_nt": [vp, i32, vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, vp, i64, vp],
    "tasu_f32_gemm_stream": [vp, i32, vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp, i64, vp],
    "tasu_f32_to_fragment_order": [vp, i32, vp, i32, i32, vp],
    "tasu_f32_gemm_streams": [i32, i32, i32, i64],
    "tasu_f32_rmsnorm": [vp, vp, vp, i32, i32, f32, vp],
    "tasu_f32_gemm_resid_rmsnorm": [vp, i32, vp, i32, vp, i32, vp, vp, vp, vp, i32, i32, i32, f32, vp, i64, vp],
    "tasu_f32_gemm_swiglu": [vp, i32, vp, i32, vp, vp, i32, i32, i32, vp, i64, vp],
    "tasu_f32_gemm_qkv_rope": [vp, i32, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, i32, vp, i64, vp],
    "tasu_f32_rope": [vp, vp, vp, i32, i32, i32, vp, vp, vp, i32, i32, vp],
    "tasu_f32_kv_fill": [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "tasu_f32_attn_prefill": [vp, vp, vp, vp, i32, i32, i32, i32, f32, vp],
    "tasu_f32_fsmn": [vp, i32, vp, vp, vp, i32, i32, i32, i32, vp],
    "tasu_f32_attn_decode": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp],
    "tasu_f32_swiglu": [vp, vp, i32, i32, vp],
    "tasu_f32_embed_merge": [vp, vp, i32, vp, vp, vp, i32, i32, vp],
    "tasu_f32_logprob_topk": [vp, i32, i32, i32, i32, vp, i32, vp, vp, vp, i64, vp],
    "tasu_f32_ce": [vp, i32, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp],
    # fp32 training step: backward kernels (csrc/fp32_train.hip)
    "tasu_f32_rmsnorm_bwd": [vp, vp, vp, vp, i32, i32, f32, i32, vp],
    "tasu_f32_swiglu_bwd": [vp, vp, vp, i32, i32, vp],
    "tasu_f32_silu": [vp, vp, vp, i64, vp],
    "tasu_f32_colsum": [vp, i32, vp, i32, i32, vp],
    "tasu_f32_layernorm_bwd_params": [vp, i32, vp, i32, vp, vp, vp, vp, i32, i32, vp],
    "tasu_f32_transpose": [vp, i32, vp, i32, i32, i32, i32, vp],
    "tasu_f32_gather_rows": [vp, vp, vp, i32, i32, vp],
    "tasu_f32_attn_bwd": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp],
})

ABI_VERSION = 16
_lib = None

GEMM_SOURCES = ("common.h", "gemm_epilogue.h", "gemm.hip", "gemm_pipe.hip", "gemm_pp.hip")


def gemm_source_hash():
    """sha256 (16 hex digits) over the training GEMM kernels' sources: profiles/*_gemm_pmc.json records it, and bench.py reports the
    counters' ``roofline.traffic`` only while it still describes the kernels that ran (the figure used to go stale
    silently when the kernels changed)."""
    import hashlib
    h = hashlib.sha256()
    for name in GEMM_SOURCES:
        with open(os.path.join(_HERE, "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


class TasuLibraryError(RuntimeError):
    pass


def load():
    """Loads libtasu_hip.so (built by ``__graft_entry__.build()`` / ``make -C ps_slm_amd/csrc``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise TasuLibraryError(f"{LIB_PATH} is missing: build it with `make -C ps_slm_amd/csrc` "
                               f"(or python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise TasuLibraryError(f"{LIB_PATH} does not export {name} (stale build?)") from e
        fn.argtypes = argtypes
        fn.restype = C.c_int64 if name in RESTYPE_I64 else C.c_int
    if lib.tasu_abi_version() != ABI_VERSION:
        raise TasuLibraryError(f"ABI version mismatch: library {lib.tasu_abi_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib

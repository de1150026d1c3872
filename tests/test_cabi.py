"""The C-ABI library loads and exports exactly what include/tasu_hip.h declares (no compute calls: CPU box)."""
import os
import re

import pytest

from conftest import ROOT


def header_functions():
    txt = open(os.path.join(ROOT, "include", "tasu_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return {m.group(1): m.group(2) for m in re.finditer(r"\bint(?:64_t)?\s+(tasu_\w+)\s*\(([^;]*?)\)\s*;", txt, flags=re.S)}


def test_every_declared_symbol_is_exported_and_bound():
    from ps_slm_amd import _lib
    decl = header_functions()
    assert len(decl) >= 30
    lib = _lib.load()
    assert lib.tasu_abi_version() == _lib.ABI_VERSION
    assert set(decl) == set(_lib.PROTOTYPES), set(decl) ^ set(_lib.PROTOTYPES)
    for name, args in decl.items():
        assert hasattr(lib, name), name
        n_args = 0 if args.strip() in ("", "void") else args.count(",") + 1
        assert n_args == len(_lib.PROTOTYPES[name]), (name, n_args, len(_lib.PROTOTYPES[name]))


def test_signatures_are_plain_c():
    for name, args in header_functions().items():
        assert "torch" not in args and "at::" not in args and "std::" not in args, name


def test_bad_arguments_are_rejected_without_a_gpu():
    """Argument validation happens before any launch, so it can be exercised here: rc == 1 (TASU_ERR_ARG)."""
    from ps_slm_amd import _lib
    lib = _lib.load()
    assert lib.tasu_gemm_nt_bf16(None, 64, None, 64, None, 64, None, None, 64, 64, 64, 0, None) == 1
    assert lib.tasu_rmsnorm_fwd(None, None, None, None, 4, 6, 1e-6, None) == 1
    assert lib.tasu_adamw(None, None, None, None, None, 0, 5e-5, 0.9, 0.999, 1e-6, 0.0, 1, 1.0, None) == 1
    # round 5's decode entry points: null operands, and the in-GEMM norms' K <= 2048 rule (checked before the pointers are looked at
    # only for the null case here: no device memory on this box)
    assert lib.tasu_gemm_stream_resid_prenorm(None, 1536, None, 1536, None, None, 64, 1536, 1536, None, None, 1, None, 1, 1, None) == 1
    assert lib.tasu_gemm_stream_swiglu_rstd(None, 1536, None, 1536, None, 8960, 64, 8960, 1536, None, 96, 1e-6, 1, 1, 1, None) == 1
    assert lib.tasu_stream_finish_prenorm(None, 7, None, None, 64, 1536, None, None, 1, None, None) == 1
    assert lib.tasu_gemm_stream_qkv_rope_rstd(None, 1536, None, 1536, None, None, 64, 12, 2, 1536, None, None, None, None, None, 328, None, 96,
                                              1e-6, 1, 1, None) == 1
    assert lib.tasu_stream_finish_norm(None, 13, None, None, 64, 3584, None, None, 1e-6, 1, None) == 1


def test_stream_k_ranges_equal_and_ragged():
    """tasu_stream_supported (host code): the K ranges the weight-streaming decode GEMMs take -- equal ranges of 256 / 512 / 1280 /
    1536 / 1792, ONE range of 3584 (Qwen2.5-7B's hidden size: 14 k-steps per wave on row halves), or n - 1 equal ranges and a
    shorter last one (the 7B's down projection K = 18944 = 12 x 1536 + 512)."""
    from ps_slm_amd import _lib
    lib = _lib.load()
    yes = [(256, 1), (1536, 1), (1792, 1), (3584, 1), (3584, 2), (8960, 7), (8960, 5), (18944, 13), (18944, 37), (512, 2)]
    no = [(3584, 4), (18944, 12), (18944, 1), (7168, 1), (1000, 1), (8960, 3), (1536, 0), (2048, 1)]
    for K, ks in yes:
        assert lib.tasu_stream_supported(K, ks) == 1, (K, ks)
    for K, ks in no:
        assert lib.tasu_stream_supported(K, ks) == 0, (K, ks)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from ps_slm_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.TasuLibraryError):
        _lib.load()


def test_hipops_refuses_cpu_box():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ps_slm_amd.ops import HipOps, TasuOpError
    with pytest.raises(TasuOpError):
        HipOps()


@pytest.mark.parametrize("tiles,pairs,grid", [(96, 140, 256), (96, 70, 256), (48, 140, 256), (24, 140, 256), (72, 64, 256), (45, 48, 256),
                                              (288, 8, 256), (400, 12, 256), (16, 32, 256), (256, 12, 256), (7, 1187, 256), (100, 9, 64)])
def test_streamk_schedule_covers_every_tile_once_and_cannot_deadlock(tiles, pairs, grid):
    """The stream-K work-item lists of the 256 x 256 GEMM, produced by the kernel's own schedule code on the host
    (tasu_streamk_schedule): every K-tile of every output tile belongs to exactly one item; the items of a cut tile are
    contiguous in K and belong to consecutive workgroups; a partial-tile producer piece is always its workgroup's FIRST item
    (it never waits, so the owner that waits for it -- a lower-numbered workgroup, on its last range piece -- cannot be held up
    by anything but time); every piece is an even number (>= 2) of K-tiles; whole tiles are dealt round-robin."""
    import ctypes as C
    import numpy as np
    from ps_slm_amd import _lib
    lib = _lib.load()
    max_items = 16
    items = np.full((grid, max_items, 4), -1, dtype=np.int32)
    counts = np.zeros(grid, dtype=np.int32)
    sk = lib.tasu_streamk_schedule(tiles, pairs, grid, items.ctypes.data_as(C.c_void_p), counts.ctypes.data_as(C.c_void_p), max_items)
    assert 0 <= sk <= tiles
    cover = np.zeros((tiles, 2 * pairs), dtype=np.int32)
    owner_of, parts_of = {}, {}
    for w in range(grid):
        seen_whole = False
        for i in range(counts[w]):
            tile, k0, nk, kind = (int(v) for v in items[w, i])
            assert 0 <= tile < tiles and nk >= 2 and nk % 2 == 0 and k0 % 2 == 0 and k0 + nk <= 2 * pairs
            cover[tile, k0:k0 + nk] += 1
            if kind == 1:
                assert i == 0, "a producer piece must be the workgroup's first item"
                assert k0 > 0
                parts_of.setdefault(tile, []).append((k0, w))
            elif kind == 2:
                assert k0 == 0 and nk < 2 * pairs and not seen_whole, "an owner piece begins its tile and precedes the whole tiles"
                assert tile not in owner_of
                owner_of[tile] = (w, nk)
            else:
                assert k0 == 0 and nk == 2 * pairs
                seen_whole = seen_whole or tile < tiles - sk
    assert (cover == 1).all()
    assert set(parts_of) == set(owner_of)
    for tile, (w, end) in owner_of.items():
        assert tile >= tiles - sk
        for j, (k0, pw) in enumerate(sorted(parts_of[tile])):
            assert pw == w + 1 + j and k0 == end, "the pieces of a tile follow each other in K on consecutive workgroups"
            end = k0 + int(items[pw, 0, 2])
        assert end == 2 * pairs
    if sk == 0:
        assert not owner_of and counts.max() == -(-tiles // grid)
    else:
        work = np.array([sum(int(items[w, i, 2]) for i in range(counts[w])) for w in range(grid)])
        assert work.max() - work.min() <= 2 * pairs + 16 if tiles > grid else work.max() - work.min() <= 16   # balanced up to the snapping


def test_gemm_policy_on_the_training_step_shapes():
    """tasu_gemm_plan: the dispatcher's choice (no launch, no GPU) for the GEMM shapes of the benchmark step and its neighbours --
    the measured winners of profiles/r03_gemm_lab.txt / r03_gemm_streamk.txt.  A policy edit that moves one of them shows up here
    (round 3: the stream-K option once shadowed whole 256 x 256 tiles and cost the 7B and audio-SFT steps 6-8 %)."""
    from ps_slm_amd import _lib
    lib = _lib.load()
    PP, PP_P128, PP_P192, SK, P128, P192, P96, SPLITK, TILES = range(1, 10)
    BF, F32, RES = 0, 1, 2
    M = 4096
    want = [
        ((M, 2048, 1536, BF), P128),            # q|k|v (the step itself runs tasu_gemm_qkv_rope: the same 256 x 128 tiles)
        ((M, 1536, 1536, RES), P192),           # o: one round of 128 x 192
        ((M, 1536, 1536, BF), P192),            # d_o
        ((M, 1536, 2048, BF), P192),            # d_qkv
        ((M, 1536, 8960, RES), P192),           # down: the cut loses at K = 8960 (108 against 100 us)
        ((M, 8960, 1536, BF), PP_P192),         # d_down: two whole rounds + column tail
        ((M, 17920, 1536, BF), PP_P192),        # gate|up as a plain GEMM: four whole rounds + column tail
        ((M, 1536, 17920, BF), SK),             # d_gate_up: 96 tiles on 256 CUs, stream-K
        ((2048, 151936, 1536, BF), PP),         # lm_head on the labelled rows
        ((1664, 2048, 25088, BF), SK),          # projector
        ((M, 3584, 18944, RES), PP),            # 7B down: 224 tiles, whole (the cut loses above 3/4 of a round)
        ((M, 3584, 37888, BF), PP),             # 7B d_gate_up
        ((2048, 1536, 17920, BF), SK),          # batch 8: 48 tiles
        ((2048, 1536, 8960, RES), SK),
        ((3072, 1536, 17920, BF), P192),        # batch 12: 72 tiles, K offsets not aligned within an XCD
        ((1024, 1536, 17920, BF), SK),          # 24 tiles: up to a quarter round the offsets do not matter
        ((1024, 1536, 8960, BF), P192),         # ... but 6.5 K-tile pairs per CU are too few
        ((512, 1536, 17920, BF), SPLITK),       # nothing else fills the chip
        ((128, 1536, 8960, BF), P192),
        ((64, 1536, 1536, BF), TILES),
    ]
    for (m, n, k, mode), plan in want:
        assert lib.tasu_gemm_plan(m, n, k, mode, 1) == plan, (m, n, k, mode, lib.tasu_gemm_plan(m, n, k, mode, 1), plan)
    # without the workspace the stream-K schedule is not available
    assert lib.tasu_gemm_plan(M, 1536, 17920, BF, 0) == P192
    assert lib.tasu_gemm_plan(M, 1536, 17920, 5, 1) == -1 and lib.tasu_gemm_plan(M, 1536, 100, BF, 1) == -1


def test_rccl_binding_picks_the_mapped_copy_and_refuses_a_second_one():
    """csrc/comm.hip: the library's RCCL functions come from the copy already mapped into the process (torch ships its own under
    torch/lib and loads it by path, so a soname lookup would miss it and bind /opt/rocm's beside it); TASU_RCCL_PATH naming
    another file is refused with a reason instead of loading a second RCCL (host-only calls: no GPU needed)."""
    import ctypes
    import subprocess
    import sys
    code = ("import ctypes, os\nfrom ps_slm_amd import _lib\nlib = _lib.load()\nb = ctypes.create_string_buffer(640)\n"
            "rc = lib.tasu_comm_library(b, 640)\nprint(lib.tasu_comm_available(), rc, b.value.decode())\n"
            "print(len({l.split()[-1] for l in open('/proc/self/maps') if 'librccl' in l}))\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    env.pop("TASU_RCCL_PATH", None)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=root, timeout=300).stdout.split("\n")
    avail, rc, path = out[0].split(" ", 2)
    mapped = [l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l]
    if mapped:                                             # torch maps its RCCL at import on a ROCm build
        assert (avail, rc) == ("1", "0") and os.path.samefile(path, mapped[0]) and out[1] == "1", out
        other = "/opt/rocm/lib/librccl.so.1"
        if os.path.isfile(other) and not os.path.samefile(other, mapped[0]):
            env["TASU_RCCL_PATH"] = other
            out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=root, timeout=300).stdout.split("\n")
            avail, rc, why = out[0].split(" ", 2)
            assert (avail, rc) == ("0", "2") and "second RCCL" in why and out[1] == "1", out
    from ps_slm_amd import _lib
    lib = _lib.load()
    assert lib.tasu_comm_count(None, None) == 1 and lib.tasu_comm_library(None, 0) == 1

"""TEST DOUBLE (never shipped, never imported by ps_slm_amd): a torch-CPU implementation of the operator
interface of ps_slm_amd/ops.py with the same signatures and the same rounding points as the HIP kernels.
It lets the CPU test-suite drive the REAL host code (merge plan, buffer bookkeeping, forward/backward
schedule, engine, DP) end to end and compare it with the oracle, so that the only thing left to check on
the GPU is the kernels themselves (tests/test_gpu_*.py compare HipOps with this double op by op)."""
import contextlib
import math

import numpy as np

import torch
import torch.nn.functional as F

HD = 128
GEMM_BF16, GEMM_F32, GEMM_RESID = 0, 1, 2


def _bf(x):
    return x.to(torch.bfloat16)


from oracle.lora_oracle import lora_keep_scale  # noqa: E402  (numpy restatement of csrc/lora.hip's dropout mask)


class FakeOps:
    name = "fake-cpu"

    # ---------------------------------------------------------------- GEMM & layout
    @contextlib.contextmanager
    def alt_workspace(self, who):
        """HipOps.alt_workspace: the double has no GEMM workspace."""
        yield

    def gemm(self, a, b, c, M, N, K, bias=None, resid=None, mode=GEMM_BF16, lda=None, ldb=None, ldc=None):
        assert K % 64 == 0, "K must be a multiple of 64 (layout contract of tasu_gemm_nt_bf16)"
        A = a.reshape(-1, a.shape[-1])[:M, :K].float()
        Bm = b.reshape(-1, b.shape[-1])[:N, :K].float()
        acc = A @ Bm.t()
        if bias is not None:
            acc = acc + bias[:N].float()
        C = c.reshape(-1, c.shape[-1])
        if mode == GEMM_BF16:
            C[:M, :N] = _bf(acc)
        elif mode == GEMM_F32:
            C[:M, :N] = acc
        else:
            R = resid.reshape(-1, resid.shape[-1])
            C[:M, :N] = R[:M, :N] + _bf(acc).float()

    def gemm_skinny(self, a, b, c, M, N, K, ws, bias=None, resid=None, mode=GEMM_BF16):
        assert M <= 64
        self.gemm(a, b, c, M, N, K, bias=bias, resid=resid, mode=mode)

    def transpose(self, src, dst, R, C, Rpad, Cpad):
        dst[:Cpad, :Rpad] = 0
        dst[:C, :R] = src[:R, :C].t()

    def cast_bf16(self, src, dst):
        dst.copy_(_bf(src))

    # ---------------------------------------------------------------- norms
    def rmsnorm_fwd(self, x, w, y, rstd, eps):
        r = torch.rsqrt(x.pow(2).mean(-1) + eps)
        if rstd is not None:
            rstd.copy_(r)
        y.copy_(_bf(w * (x * r[:, None])))

    def rmsnorm_bwd(self, dy, x, w, rstd, dx, dx_bf16, accumulate):
        D = x.shape[-1]
        d = dy.float()
        xh = x * rstd[:, None]
        dot = (w * d * xh).sum(-1, keepdim=True) / D
        upd = rstd[:, None] * (w * d - xh * dot)
        if accumulate:
            dx.add_(upd)
        else:
            dx.copy_(upd)
        if dx_bf16 is not None:
            dx_bf16.copy_(_bf(dx))

    def rmsnorm_fwd_rows(self, x, src_rows, w, y, rstd, eps):
        src = src_rows[: y.shape[0]].long()
        ok = src >= 0
        xs = x[src.clamp_min(0)]
        r = torch.rsqrt(xs.pow(2).mean(-1) + eps) * ok
        rstd[: y.shape[0]].copy_(r)
        y.copy_(_bf(w * (xs * r[:, None])))

    def rmsnorm_bwd_rows(self, dy, x, w, rstd, slot, dx, dx_bf16):
        M, D = x.shape
        s = slot[:M].long()
        ok = (s >= 0)[:, None]
        d = dy.float()[s.clamp_min(0)]
        r = rstd[s.clamp_min(0)][:, None]
        xh = x * r
        dot = (w * d * xh).sum(-1, keepdim=True) / D
        dx.copy_(torch.where(ok, r * (w * d - xh * dot), torch.zeros_like(x)))
        if dx_bf16 is not None:
            dx_bf16.copy_(_bf(dx))

    def rmsnorm_bwd_rows_resid(self, dy, x, w, rstd, slot, resid, dx, dx_bf16):
        M, D = x.shape
        s = slot[:M].long()
        ok = (s >= 0)[:, None]
        d = dy.float()[s.clamp_min(0)]
        r = rstd[s.clamp_min(0)][:, None]
        xh = x * r
        dot = (w * d * xh).sum(-1, keepdim=True) / D
        dx.copy_(torch.where(ok, resid[s.clamp_min(0)] + r * (w * d - xh * dot), torch.zeros_like(x)))
        if dx_bf16 is not None:
            dx_bf16.copy_(_bf(dx))

    def scale_softmax_rows(self, s, p, R, V, denom, stats=None):
        z = _bf(s[:R, :V].float() / denom).float()
        m = z.max(-1, keepdim=True).values
        e = torch.exp(z - m)
        inv = 1.0 / e.sum(-1, keepdim=True)
        p.zero_()
        p[:R, :V] = _bf(e * inv)
        if stats is not None:
            stats[:R, 0], stats[:R, 1] = m[:, 0], inv[:, 0]

    def softmax_bwd_rows(self, s, stats, dp, ds, R, V, denom):
        z = _bf(s[:R, :V].float() / denom).float()
        P = torch.exp(z - stats[:R, 0:1]) * stats[:R, 1:2]        # the fp32 softmax output autograd saves
        dP = dp[:R, :V].float()
        ds.zero_()
        ds[:R, :V] = _bf(_bf(P * (dP - (P * dP).sum(-1, keepdim=True))).float() / denom)

    def layernorm_fwd(self, x, gamma, beta, y, mean, rstd, R, D, eps):
        xr = x[:R, :D]
        mu = xr.mean(-1)
        var = ((xr - mu[:, None]) ** 2).mean(-1)
        r = torch.rsqrt(var + eps)
        if mean is not None:
            mean[:R] = mu
        if rstd is not None:
            rstd[:R] = r
        y[:R].zero_()
        y[:R, :D] = ((xr - mu[:, None]) * r[:, None] * gamma[:D] + beta[:D]).to(y.dtype)

    def layernorm_bwd_params(self, dy, x, mean, rstd, dgamma, dbeta, ws, R, D):
        d = dy[:R, :D].float()
        xh = (x[:R, :D] - mean[:R, None]) * rstd[:R, None]
        dgamma[:D] = (d * xh).sum(0)
        dbeta[:D] = d.sum(0)

    def colsum(self, x, out, R, Cn):
        out[:Cn] = x[:R, :Cn].float().sum(0)

    # ---------------------------------------------------------------- rope + attention
    def rope_table(self, pos, cos, sin, head_dim, theta):
        inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
        ang = pos.float()[:, None] * inv[None]
        cos.copy_(ang.cos())
        sin.copy_(ang.sin())

    @staticmethod
    def _rot(x, cos, sin, inverse=False):
        x1, x2 = x[..., :64], x[..., 64:]
        if inverse:
            return torch.cat([x1 * cos + x2 * sin, x2 * cos - x1 * sin], -1)
        return torch.cat([x1 * cos - x2 * sin, x2 * cos + x1 * sin], -1)

    def rope_fwd(self, qkv, cos, sin, qt, kt, vt, B, S, H, G):
        Spad = (S + 63) // 64 * 64
        v = qkv.view(B, S, H + 2 * G, HD)
        c, s = cos.view(B, S, 1, 64), sin.view(B, S, 1, 64)
        v[:, :, : H + G] = _bf(self._rot(v[:, :, : H + G].float(), c, s))
        for dst, lo, n in ((qt, 0, H), (kt, H, G), (vt, H + G, G)):
            if dst is None:
                continue
            t = dst.view(B, n, HD, Spad)
            t.zero_()
            t[..., :S] = v[:, :, lo:lo + n].permute(0, 2, 3, 1)

    def rope_bwd(self, dqkv, dk_part, dv_part, cos, sin, B, S, H, G):
        v = dqkv.view(B, S, H + 2 * G, HD)
        c, s = cos.view(B, S, 1, 64), sin.view(B, S, 1, 64)
        rep = H // G
        hpb = 3 if rep % 3 == 0 else (2 if rep % 2 == 0 else 1)
        npg, M = rep // hpb, B * S
        dk = dk_part.reshape(-1)[: M * G * npg * HD].view(B, S, G, npg, HD).sum(3)
        dv = dv_part.reshape(-1)[: M * G * npg * HD].view(B, S, G, npg, HD).sum(3)
        v[:, :, :H] = _bf(self._rot(v[:, :, :H].float(), c, s, inverse=True))
        v[:, :, H:H + G] = _bf(self._rot(dk, c, s, inverse=True))
        v[:, :, H + G:] = _bf(dv)

    @staticmethod
    def _allow(key_mask, B, S, causal):
        km = key_mask.view(B, -1)[:, :S].bool()
        allow = km[:, None, None, :].expand(B, 1, S, S)
        if causal:
            allow = allow & torch.tril(torch.ones(S, S, dtype=torch.bool))[None, None]
        return allow

    def _qkv_heads(self, qkv, B, S, H, G):
        v = qkv.view(B, S, H + 2 * G, HD).float()
        rep = H // G
        q = v[:, :, :H].permute(0, 2, 1, 3)
        k = v[:, :, H:H + G].permute(0, 2, 1, 3).repeat_interleave(rep, 1)
        vv = v[:, :, H + G:].permute(0, 2, 1, 3).repeat_interleave(rep, 1)
        return q, k, vv

    def attn_fwd(self, qkv, vt, key_mask, out, lse, B, S, H, G, scale, causal):
        Spad = (S + 63) // 64 * 64
        q, k, v = self._qkv_heads(qkv, B, S, H, G)
        sc = (q @ k.transpose(-1, -2)) * scale
        allow = self._allow(key_mask, B, S, causal)
        sc = sc.masked_fill(~allow, float("-inf"))
        m = sc.amax(-1, keepdim=True)
        m = torch.where(torch.isinf(m), torch.zeros_like(m), m)
        p = torch.exp(sc - m)
        l = p.sum(-1, keepdim=True)
        o = (_bf(p).float() @ v) / torch.where(l > 0, l, torch.ones_like(l))
        o = torch.where(l > 0, o, torch.zeros_like(o))
        out.view(B, S, H, HD).copy_(_bf(o.permute(0, 2, 1, 3)))
        ls = torch.where(l > 0, m + torch.log(l), torch.zeros_like(l))[..., 0]
        lse.view(B, H, Spad)[..., :S] = ls

    def attn_bwd_prep(self, dout, out, delta, dout_t, B, S, H):
        Spad = (S + 63) // 64 * 64
        d = dout.view(B, S, H, HD)
        delta.view(B, H, Spad)[..., :S] = (d.float() * out.view(B, S, H, HD).float()).sum(-1).permute(0, 2, 1)
        if dout_t is not None:
            t = dout_t.view(B, H, HD, Spad)
            t.zero_()
            t[..., :S] = d.permute(0, 2, 3, 1)

    def _bwd_common(self, qkv, key_mask, dout, lse, delta, B, S, H, G, scale, causal):
        Spad = (S + 63) // 64 * 64
        q, k, v = self._qkv_heads(qkv, B, S, H, G)
        do = dout.view(B, S, H, HD).float().permute(0, 2, 1, 3)
        allow = self._allow(key_mask, B, S, causal)
        ls = lse.view(B, H, Spad)[..., :S]
        dl = delta.view(B, H, Spad)[..., :S]
        p = torch.exp((q @ k.transpose(-1, -2)) * scale - ls[..., None])
        p = torch.where(allow, p, torch.zeros_like(p))
        dp = do @ v.transpose(-1, -2)
        ds = p * (dp - dl[..., None])
        return q, k, v, do, p, ds

    def attn_bwd_dq(self, qkv, kt, key_mask, dout, lse, delta, dqkv, B, S, H, G, scale, causal):
        q, k, v, do, p, ds = self._bwd_common(qkv, key_mask, dout, lse, delta, B, S, H, G, scale, causal)
        dq = (_bf(ds).float() @ k) * scale
        dqkv.view(B, S, H + 2 * G, HD)[:, :, :H] = _bf(dq.permute(0, 2, 1, 3))

    def attn_bwd_dkv(self, qkv, qt, key_mask, dout, dout_t, lse, delta, dk_part, dv_part, B, S, H, G, scale, causal):
        q, k, v, do, p, ds = self._bwd_common(qkv, key_mask, dout, lse, delta, B, S, H, G, scale, causal)
        dv = _bf(p).float().transpose(-1, -2) @ do
        dk = (_bf(ds).float().transpose(-1, -2) @ q) * scale
        rep = H // G
        hpb = 3 if rep % 3 == 0 else (2 if rep % 2 == 0 else 1)
        M = B * S
        # partial layout of tasu_attn_bwd_dkv: [M, (H/hpb) * 128], heads summed in groups of hpb
        dk_part.reshape(-1)[: M * (H // hpb) * HD].view(B, S, H // hpb, HD).copy_(
            dk.permute(0, 2, 1, 3).reshape(B, S, H // hpb, hpb, HD).sum(3))
        dv_part.reshape(-1)[: M * (H // hpb) * HD].view(B, S, H // hpb, HD).copy_(
            dv.permute(0, 2, 1, 3).reshape(B, S, H // hpb, hpb, HD).sum(3))

    # ---------------------------------------------------------------- activations
    def attn_bwd(self, qkv, qt, kt, key_mask, dout, dout_t, lse, delta, dqkv, dk_part, dv_part, B, S, H, G, scale, causal):
        self.attn_bwd_dq(qkv, kt, key_mask, dout, lse, delta, dqkv, B, S, H, G, scale, causal)
        self.attn_bwd_dkv(qkv, qt, key_mask, dout, dout_t, lse, delta, dk_part, dv_part, B, S, H, G, scale, causal)

    def attn_bwd_rope(self, qkv, key_mask, dout, lse, delta, cos, sin, dqkv, dk_part, dv_part, B, S, H, G, scale, causal, kernel=0):
        self.attn_bwd(qkv, None, None, key_mask, dout, None, lse, delta, dqkv, dk_part, dv_part, B, S, H, G, scale, causal)
        self.rope_bwd(dqkv, dk_part, dv_part, cos, sin, B, S, H, G)

    def attn_fwd_on(self, kernel, qkv, key_mask, out, lse, B, S, H, G, scale, causal):
        self.attn_fwd(qkv, None, key_mask, out, lse, B, S, H, G, scale, causal)

    def attn_bwd_fused(self, qkv, key_mask, dout, out, lse, delta, cos, sin, dqkv, dk_part, dv_part, B, S, H, G, scale, causal,
                       kernel=None):
        self.attn_bwd_prep(dout, out, delta, None, B, S, H)
        self.attn_bwd_rope(qkv, key_mask, dout, lse, delta, cos, sin, dqkv, dk_part, dv_part, B, S, H, G, scale, causal)

    def swiglu_fwd(self, gu, act, M, I):
        g, u = gu[:, :I].float(), gu[:, I:].float()
        act.copy_(_bf(_bf(F.silu(g)).float() * u))

    def gemm_qkv_rope(self, a, wqkv, bias, qkv, cos, sin, M, H, G, K):
        self.gemm(a, wqkv, qkv, M, (H + 2 * G) * 128, K, bias=bias)
        self.rope_fwd(qkv, cos, sin, None, None, None, 1, M, H, G)

    def gemm_dswiglu(self, dy, wd_t, gu, dgu, dact_ws, M, I, K):
        dact = torch.zeros(M, I, dtype=torch.bfloat16) if dact_ws is None else dact_ws
        self.gemm(dy, wd_t, dact, M, I, K)
        self.swiglu_bwd(dact, gu, dgu, M, I)

    def swiglu_bwd(self, dact, gu, dgu, M, I):
        g, u, d = gu[:, :I].float(), gu[:, I:].float(), dact.float()
        sg = torch.sigmoid(g)
        dgu[:, :I] = _bf(d * u * sg * (1 + g * (1 - sg)))
        dgu[:, I:] = _bf(d * g * sg)

    def silu_fwd(self, x, y):
        y.copy_(_bf(F.silu(x.float())))

    def silu_bwd(self, dy, x, dx):
        f = x.float()
        sg = torch.sigmoid(f)
        dx.copy_(_bf(dy.float() * sg * (1 + f * (1 - sg))))

    def relu_fwd(self, x, y):
        y.copy_(torch.relu(x))

    def gemm_bias_relu(self, a, b, c, M, N, K, bias):
        self.gemm(a, b, c, M, N, K, bias=bias)
        c.copy_(torch.relu(c))

    # ---------------------------------------------------------------- LoRA (csrc/gemm_rank.hip, csrc/lora.hip)
    def gemm_rank(self, a, b, c, M, N, K, f32=False, transposed=False):
        assert N <= 64 and K % 64 == 0
        acc = a[:M, :K].float() @ b[:N, :K].float().t()
        out = acc if f32 else _bf(acc)
        if transposed:
            c[:N, :M] = out.t()
        else:
            c[:M, :N] = out

    def gemm_rank_tn(self, at, b, c, M, N, K, transposed=False):
        assert N <= 64 and K % 64 == 0 and M % 64 == 0
        acc = at[:K, :M].float().t() @ b[:N, :K].float().t()
        if transposed:
            c[:N, :M] = acc.t()
        else:
            c[:M, :N] = acc

    def lora_apply(self, y, u, w, M, N, R, s=1.0, p=0.0, rng=None, sid=0, x_in=None, x_out=None):
        d = _bf(_bf(u[:M, :R].float() @ w[:N, :R].float().t()).float() * float(s)).float()
        if p > 0.0:
            d = _bf(d * lora_keep_scale(rng, sid, (M, N), p)).float()
        y[:M, :N] = _bf(y[:M, :N].float() + d)
        if x_in is not None:
            x_out[:M, :N] = x_in[:M, :N] + y[:M, :N].float()

    def copy_rows(self, src, dst, M, C):
        dst[:M, :C] = src[:M, :C]

    def scale_bf16(self, src, dst, s):
        dst.copy_(_bf(src.float() * float(s)))

    def lora_dropout(self, src, dst, p, rng, sid):
        dst.copy_(_bf(src.float() * lora_keep_scale(rng, sid, src.shape, p)))

    def lora_dropout_norm(self, x, w, rstd, dst, M, D, p, rng, sid):
        dst.copy_(_bf((w * (x[:M] * rstd[:M, None])) * lora_keep_scale(rng, sid, (M, D), p)))

    def rng_advance(self, rng):
        rng[1] += 1

    def relu_bwd(self, dy, x, dx):
        dx.copy_(torch.where(x.float() > 0, dy, torch.zeros_like(dy)))

    # ---------------------------------------------------------------- loss
    def ce_fwd_bwd(self, logits, shift_labels, M, V, row_loss, row_hit, row_argmax, dlogits, inv_count):
        lg = logits[:M, :V].float()
        lab = shift_labels[:M].long()
        valid = lab >= 0
        lse = torch.logsumexp(lg, -1)
        arg = lg.argmax(-1)
        safe = lab.clamp_min(0)
        row_loss[:M] = torch.where(valid, lse - lg.gather(1, safe[:, None])[:, 0], torch.zeros_like(lse))
        row_hit[:M] = (valid & (arg == lab)).to(row_hit.dtype)
        if row_argmax is not None:
            row_argmax[:M] = arg.to(row_argmax.dtype)
        if dlogits is not None:
            g = torch.exp(lg - lse[:, None])
            g[torch.arange(M), safe] -= 1.0
            g = g * inv_count.float() * valid[:, None]
            dlogits[:M].zero_()
            dlogits[:M, :V] = _bf(g)

    def ce_reduce(self, row_loss, row_hit, shift_labels, M, out):
        c = (shift_labels[:M] >= 0).sum().float()
        out[0] = row_loss[:M].sum() / c
        out[1] = row_hit[:M].sum().float() / c
        out[2] = c
        out[3] = 1.0 / c

    # ---------------------------------------------------------------- front end / merge
    def posterior_build(self, ids, alpha, out, R, V):
        out[:R].zero_()
        idl = ids[:R].long()
        ok = idl >= 0
        a = alpha[:R] if alpha is not None else torch.zeros(R)
        base = torch.where(ok, a / V, torch.zeros_like(a))
        out[:R, :V] = base[:, None]
        rows = torch.nonzero(ok)[:, 0]
        out[rows, idl[rows]] = (1 - a[rows]) + base[rows]

    def embed_merge(self, table, proj, kind, idx, x, M, D):
        k, i = kind[:M].long(), idx[:M].long()
        x[:M].zero_()
        t = torch.nonzero(k == 1)[:, 0]
        x[t] = table[i[t]]
        a = torch.nonzero(k == 2)[:, 0]
        x[a] = proj.reshape(-1, D)[i[a]].float()

    def merge_bwd(self, dx, audio_rows, dproj, n, D):
        r = audio_rows[:n].long()
        dproj[:n].zero_()
        ok = torch.nonzero(r >= 0)[:, 0]
        dproj[ok] = _bf(dx[r[ok]])

    def beam_update(self, vals, idx, bs, first):
        """Scalar restatement of csrc/decode.hip::beam_update_kernel (float32 arithmetic in the same order)."""
        f32 = np.float32
        NEG = f32(-1.0e9)
        if int(bs.ctl[1]):
            return
        B, nb, K, cur = bs.B, bs.nb, 2 * bs.nb, int(bs.ctl[0])
        V, I = vals.numpy(), idx.numpy()
        unsat_any, stop_all = False, True
        lp_now = f32(bs.len_pow[cur + 1])
        for b in range(B):
            cand = []                                            # (score, beam, token)
            for j in range(nb):
                rs = f32(bs.run_scores[b, j])
                for k in range(K):
                    if first and j > 0:
                        v, t = NEG, 0
                    else:
                        row = b if first else b * nb + j
                        v, t = f32(V[row, k]), int(I[row, k])
                    cand.append((f32(v + rs), j, t))
            cand.sort(key=lambda c: (-c[0], c[1], c[2]))
            top = cand[:K]
            stop = [t == bs.eos or cur + 1 >= bs.max_new for _, _, t in top]
            stop_all = stop_all and all(stop)
            run_lp = [f32(s + (NEG if st else f32(0))) for (s, _, _), st in zip(top, stop)]
            nxt = sorted(range(K), key=lambda i: (-run_lp[i], i))[:nb]
            for n, i in enumerate(nxt):
                bs.bp_tok[cur, b, n], bs.bp_par[cur, b, n] = top[i][2], top[i][1]
                m = b * nb + n
                bs.next_ids[m], bs.next_src[m] = top[i][2], b * nb + top[i][1]
                bs.next_pos[m], bs.next_slot[m], bs.next_lens[m] = int(bs.valid[b]) + cur, bs.S + cur, bs.S + cur + 1
            unsat = bool(bs.unsat[b])
            merged = [(f32(bs.fin_scores[b, j]), int(bs.fin_len[b, j]), int(bs.fin_par[b, j]), int(bs.fin_tok[b, j]),
                       int(bs.is_fin[b, j])) for j in range(nb)]
            for i, (s, bm, t) in enumerate(top):
                just = stop[i] and i < nb
                sc = f32(s / lp_now)
                sc = f32(sc + (f32(0) if unsat else NEG))
                sc = f32(sc + (f32(0) if just else NEG))
                merged.append((sc, cur + 1, bm, t, int(just)))
            keep = sorted(range(nb + K), key=lambda i: (-merged[i][0], i))[:nb]
            for n, i in enumerate(keep):
                bs.fin_scores[b, n] = float(merged[i][0])
                bs.fin_len[b, n], bs.fin_par[b, n], bs.fin_tok[b, n], bs.is_fin[b, n] = merged[i][1:]
                bs.run_scores[b, n] = float(run_lp[nxt[n]])
            best_run = f32(run_lp[nxt[0]] / lp_now)
            min_fin = min(merged[i][0] for i in keep)
            improve = any(best_run > (min_fin if merged[i][4] else NEG) for i in keep)
            bs.unsat[b] = int(unsat and improve)
            unsat_any = unsat_any or bool(bs.unsat[b])
        done = not (unsat_any and not stop_all)
        bs.ctl[0], bs.ctl[1] = cur + 1, int(done)
        bs.banned[0] = bs.eos if cur + 1 < bs.min_length else -1
        if bs.done_host is not None:
            bs.done_host[0] = cur + 1 if done else 0

    # ---------------------------------------------------------------- optimizer
    def adamw(self, p, g, m, v, p_bf16, lr, beta1, beta2, eps, wd, step, grad_scale):
        lr = float(torch.tensor(lr, dtype=torch.float32))        # the kernel takes lr as a C float
        gr = g * grad_scale
        m.mul_(beta1).add_(gr, alpha=1 - beta1)
        v.mul_(beta2).addcmul_(gr, gr, value=1 - beta2)
        bc1 = 1 - beta1 ** step
        bc2 = 1 - beta2 ** step
        denom = v.sqrt() / math.sqrt(bc2) + eps
        p.mul_(1 - lr * wd).addcdiv_(m, denom, value=-lr / bc1)
        if p_bf16 is not None:
            p_bf16.copy_(_bf(p))

    # ---------------------------------------------------------------- encoder / PSD
    def sinusoid_pe(self, x, y, B, T, D, scale):
        pos = torch.arange(1, T + 1, dtype=torch.float32)
        inc = math.log(10000.0) / (D / 2 - 1)
        inv = torch.exp(torch.arange(D // 2, dtype=torch.float32) * (-inc))
        st = pos[:, None] * inv[None]
        pe = torch.cat([st.sin(), st.cos()], 1)
        y.view(B, T, D).copy_(x.view(B, T, D) * scale + pe[None])

    def fsmn_ln_fwd(self, v, ldv, w, lens, x, gamma, beta, xn, B, T, D, ksize, eps):
        self.fsmn_fwd(v, ldv, w, lens, x, B, T, D, ksize, True)
        self.layernorm_fwd(x, gamma, beta, xn, None, None, B * T, D, eps)

    def fsmn_fwd(self, v, ldv, w, lens, out, B, T, D, ksize, accumulate):
        vv = v.reshape(B, T, -1)[..., :D].float()
        mask = (torch.arange(T)[None] < lens[:, None]).float()[..., None]
        vm = vv * mask
        left = (ksize - 1) // 2
        xp = F.pad(vm.transpose(1, 2), (left, ksize - 1 - left))
        fs = F.conv1d(xp, w.view(D, 1, ksize), None, groups=D).transpose(1, 2)
        r = ((fs + vm) * mask).reshape(B * T, D)
        if accumulate:
            out.add_(r)
        else:
            out.copy_(r)

    def softmax_rows(self, x, y, R, V):
        y[:R].zero_()
        y[:R, :V] = torch.softmax(x[:R, :V].float(), -1)

    @staticmethod
    def _post3(post, B, T, bstride):
        ld = post.stride(0)
        return torch.as_strided(post, (B, T, post.shape[1]), (bstride * ld, ld, 1))

    def psd_frame_stats(self, post, lens, fid, fblank, B, T, bstride, V, blank_id):
        p = self._post3(post, B, T, bstride)[..., :V]
        ids = p.argmax(-1).to(torch.int32)
        live = torch.arange(T)[None] < lens[:, None]
        fid.view(B, T).copy_(torch.where(live, ids, torch.full_like(ids, -1)))
        fblank.view(B, T).copy_(torch.where(live, p[..., blank_id], torch.zeros(B, T)))

    def psd_logit_stats(self, logits, lens, fid, fblank, fstat, B, T, bstride, V, blank_id):
        x = self._post3(logits, B, T, bstride)[..., :V].float()
        m = x.max(-1).values
        inv = 1.0 / torch.exp(x - m[..., None]).sum(-1)
        ids = x.argmax(-1).to(torch.int32)
        live = torch.arange(T)[None] < lens[:, None]
        fid.view(B, T).copy_(torch.where(live, ids, torch.full_like(ids, -1)))
        fblank.view(B, T).copy_(torch.where(live, torch.exp(x[..., blank_id] - m) * inv, torch.zeros(B, T)))
        fstat.view(B, T, 2)[..., 0] = torch.where(live, m, torch.zeros(B, T))
        fstat.view(B, T, 2)[..., 1] = torch.where(live, inv, torch.zeros(B, T))

    def psd_gather_softmax(self, logits, fstat, seg_start, seg_len, new_lens, out, B, T, bstride, Tout, V):
        x = self._post3(logits, B, T, bstride)[..., :V].float()
        st = fstat.view(B, T, 2)
        o = out[: B * Tout].view(B, Tout, -1)
        o.zero_()
        for b in range(B):
            for j in range(min(int(new_lens[b]), Tout)):
                s0, ln = int(seg_start.view(B, T)[b, j]), int(seg_len.view(B, T)[b, j])
                p = torch.exp(x[b, s0:s0 + ln] - st[b, s0:s0 + ln, 0:1]) * st[b, s0:s0 + ln, 1:2]
                o[b, j, :V] = p[0] if ln == 1 else p.sum(0) / ln

    def psd_plan(self, fid, fblank, lens, seg_start, seg_len, new_lens, B, T, blank_id, thr):
        ss, sl = seg_start.view(B, T), seg_len.view(B, T)
        for b in range(B):
            L = int(lens[b])
            ids, bp = fid.view(B, T)[b], fblank.view(B, T)[b]
            n, start = 0, 0
            for end in range(1, L + 1):
                if end == L or ids[end] != ids[start] or ids[start] == blank_id or blank_id < 0:
                    ln = end - start
                    s = bp[start:end].sum()
                    mean = s if ln == 1 else s / ln
                    if mean < thr:
                        ss[b, n], sl[b, n] = start, ln
                        n += 1
                    start = end
            new_lens[b] = n

    def psd_gather(self, post, seg_start, seg_len, new_lens, out, B, T, bstride, Tout, V):
        p = self._post3(post, B, T, bstride)
        o = out[: B * Tout].view(B, Tout, -1)
        o.zero_()
        for b in range(B):
            for j in range(min(int(new_lens[b]), Tout)):
                s0, ln = int(seg_start.view(B, T)[b, j]), int(seg_len.view(B, T)[b, j])
                seg = p[b, s0:s0 + ln, :V]
                o[b, j, :V] = seg[0] if ln == 1 else seg.sum(0) / ln

    # ---------------------------------------------------------------- decode loop
    def kv_fill(self, qkv, kc, vc, B, S, H, G, nb, ctx):
        W = G * HD
        v = qkv.view(B, S, (H + 2 * G) * HD)
        k4, v4 = kc.view(B, nb, ctx, W), vc.view(B, nb, ctx, W)
        k4[:, 0, :S] = v[:, :, H * HD:H * HD + W]          # the prompt lives in the first beam's row only
        v4[:, 0, :S] = v[:, :, H * HD + W:]

    def kv_append(self, qkv, kc, vc, pos, M, H, G, ctx):
        W = G * HD
        k3, v3 = kc.view(M, ctx, W), vc.view(M, ctx, W)
        r = torch.arange(M)
        k3[r, pos.long()] = qkv[:M, H * HD:H * HD + W]
        v3[r, pos.long()] = qkv[:M, H * HD + W:]

    def gemm_splitk(self, a, b, c, M, N, K, ksplit, ws):
        per = K // ksplit
        acc = torch.zeros(M, N)
        for s in range(ksplit):
            acc += a[:M, s * per:(s + 1) * per].float() @ b[:N, s * per:(s + 1) * per].float().t()
        c[:M, :N] = _bf(acc)

    def gemm_gate_up_swiglu(self, a, wgu, gu, act, M, I, K):
        self.gemm(a, wgu, gu, M, 2 * I, K)
        self.swiglu_fwd(gu, act, M, I)

    def begin_decode(self, D, HHD, I):
        return False

    def end_decode(self):
        pass

    def register_decode_weight(self, w, kind, N, H=0, G=0, slabs_ok=False):
        pass

    def dec_rmsnorm(self, x, w, y, eps):
        self.rmsnorm_fwd(x, w, y[: x.shape[0]], None, eps)

    def gemm_skinny_norm(self, a, b, c, resid, M, N, K, norm_w, y, eps, ws):
        self.gemm(a, b, c, M, N, K, resid=resid, mode=2)
        self.rmsnorm_fwd(c[:M], norm_w, y[:M], None, eps)

    def gemm_skinny_qkv_rope(self, a, wqkv, bias, qkv, M, H, G, K, cos, sin, kc, vc, pos, ctx, ws):
        self.gemm(a, wqkv, qkv, M, (H + 2 * G) * HD, K, bias=bias)
        self.rope_append(qkv, cos, sin, kc, vc, pos, M, H, G, ctx)

    def gemm_skinny_swiglu(self, a, wgu, act, M, I, K, ws):
        gu = torch.empty(M, 2 * I, dtype=torch.bfloat16)
        self.gemm(a, wgu, gu, M, 2 * I, K)
        self.swiglu_fwd(gu, act, M, I)

    def rope_append(self, qkv, cos, sin, kc, vc, pos, M, H, G, ctx):
        self.rope_fwd(qkv, cos, sin, None, None, None, M, 1, H, G)
        self.kv_append(qkv, kc, vc, pos, M, H, G, ctx)

    def kv_index_init(self, index, B, nb, S, ctx):
        ix = index.view(B * nb, ctx)
        m = torch.arange(B * nb, dtype=ix.dtype)
        ix[:] = m[:, None]
        ix[:, :S] = ((m // nb) * nb)[:, None]

    def kv_index_reorder(self, src, dst, src_row, lens, M, ctx):
        s2, d2 = src.view(M, ctx), dst.view(M, ctx)
        for r in range(M):
            n = int(lens[r])
            d2[r, :n] = s2[r if src_row is None else int(src_row[r]), :n]

    def attn_decode(self, qkv, kc, vc, index, kstart, lens, out, M, H, G, ctx, scale):
        W, rep = G * HD, H // G
        q = qkv[:M, :H * HD].float().view(M, H, HD)
        k3, v3 = kc.view(M, ctx, G, HD).float(), vc.view(M, ctx, G, HD).float()
        o = torch.zeros(M, H, HD)
        for r in range(M):
            a, b = int(kstart[r]), int(lens[r])
            pos = torch.arange(a, b)
            rows = torch.full((b - a,), r, dtype=torch.long) if index is None else index.view(M, ctx)[r, a:b].long()
            kr, vr = k3[rows, pos], v3[rows, pos]                    # [n, G, HD] gathered through the row index
            for h in range(H):
                g = h // rep
                s = (kr[:, g] @ (q[r, h] * scale))
                p = torch.exp(s - s.max())
                o[r, h] = (_bf(p).float() @ vr[:, g]) / p.sum()
        out[:M].view(M, H, HD).copy_(_bf(o))            # out may be padded to whole 64-row chunks

    def logprob_topk(self, logits, M, V, k, banned, n_banned, out_val, out_idx):
        lg = logits[:M, :V].float()
        lp = lg - torch.logsumexp(lg, -1, keepdim=True)
        if n_banned:
            ban = banned[:n_banned].long()
            lp[:, ban[ban >= 0]] = float("-inf")               # an entry of -1 bans nothing (tasu_beam_update lifts the EOS ban so)
        # descending by value, ties by smaller column (torch.sort is stable on the negated, index-ordered input)
        v, i = torch.sort(lp, dim=-1, descending=True, stable=True)
        out_val[:M] = v[:, :k]
        out_idx[:M] = i[:, :k].to(out_idx.dtype)

    def fbank(self, wave, n_samples, scale, win, shift, window, mel, n_mels, preemph, out):
        T = 0 if n_samples < win else 1 + (n_samples - win) // shift
        if T == 0:
            return
        idx = torch.arange(win)[None, :] + shift * torch.arange(T)[:, None]
        fr = wave[:n_samples].float()[idx] * scale
        fr = fr - fr.mean(1, keepdim=True)
        fr = (fr - preemph * torch.cat([fr[:, :1], fr[:, :-1]], 1)) * window[None, :]
        spec = torch.fft.rfft(fr.double(), n=512, dim=1)
        power = (spec.real ** 2 + spec.imag ** 2).float()
        out[:T] = torch.log(torch.clamp(power @ mel.t(), min=1.1920928955078125e-07))

    def lfr_cmvn(self, fb, T, D, lfr_m, lfr_n, means, scales, out):
        T_lfr = (T + lfr_n - 1) // lfr_n
        f = torch.arange(T_lfr)[:, None] * lfr_n + torch.arange(lfr_m)[None, :] - (lfr_m - 1) // 2
        v = fb[:T][f.clamp(0, T - 1)].reshape(T_lfr, lfr_m * D)
        if means is not None:
            v = (v + means[None, :]) * scales[None, :]
        out[:T_lfr] = v

    def decode_step_prologue(self, table, ids, x, norm_w, xn, eps, pos, cos, sin, head_dim, theta, index, index_tmp, src_row, lens,
                             n_beams, M, D, ctx):
        self.kv_index_reorder(index, index_tmp, src_row, lens, M, ctx)
        self.kv_index_reorder(index_tmp, index, None, lens, M, ctx)
        self.embed_rows(table, ids, x, M, D)
        self.rope_table(pos, cos, sin, head_dim, theta)
        self.dec_rmsnorm(x[:M], norm_w, xn, eps)

    def embed_rows(self, table, ids, x, M, D):
        x[:M] = table[ids[:M].long()]

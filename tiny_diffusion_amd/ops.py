"""torch.autograd wrappers of libtdx's row / elementwise building blocks (Linear, LayerNorm,
SiLU / GELU, dropout, add, embedding).  Used by the models that are a plain chain of such layers
(diffusion_transformer.py); every forward and backward is a HIP kernel behind include/tdx.h -
torch only owns the tensors and the autograd graph.  CUDA fp32 tensors only: no CPU fallback."""
from __future__ import annotations

import torch

from . import _lib
from ._lib import lib, check


def _st(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _prep(t):
    if not t.is_cuda:
        raise _lib.TdxError("tiny_diffusion_amd runs on MI355X only: got a CPU tensor and there is no "
                            "CPU fallback (the CPU restatement lives in oracle/ and is test-only)")
    return t.contiguous().float()


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        x, w, b = _prep(x), _prep(w), _prep(b)
        M, K = x.shape
        N = w.shape[0]
        out = torch.empty(M, N, dtype=torch.float32, device=x.device)
        check(lib.tdx_linear_fwd(x.data_ptr(), K, w.data_ptr(), b.data_ptr(), out.data_ptr(), N, M, N, K, 0, _st(x)),
              "tdx_linear_fwd")
        ctx.save_for_backward(x, w)
        return out

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = _prep(gy)
        M, K = x.shape
        N = w.shape[0]
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w)
        db = torch.empty(N, dtype=torch.float32, device=x.device)
        check(lib.tdx_linear_bwd(gy.data_ptr(), N, x.data_ptr(), K, w.data_ptr(),
                                 None if gx is None else gx.data_ptr(), K, dw.data_ptr(), db.data_ptr(), M, N, K,
                                 _st(x)), "tdx_linear_bwd")
        return gx, dw, db


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, b, eps):
        x, g, b = _prep(x), _prep(g), _prep(b)
        M, N = x.shape
        out = torch.empty_like(x)
        mean = torch.empty(M, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        check(lib.tdx_layernorm_fwd(x.data_ptr(), g.data_ptr(), b.data_ptr(), out.data_ptr(), mean.data_ptr(),
                                    rstd.data_ptr(), M, N, eps, _st(x)), "tdx_layernorm_fwd")
        ctx.save_for_backward(x, g, mean, rstd)
        return out

    @staticmethod
    def backward(ctx, gy):
        x, g, mean, rstd = ctx.saved_tensors
        gy = _prep(gy)
        M, N = x.shape
        gx, dg, db = torch.empty_like(x), torch.empty_like(g), torch.empty_like(g)
        check(lib.tdx_layernorm_bwd(gy.data_ptr(), x.data_ptr(), g.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                    gx.data_ptr(), dg.data_ptr(), db.data_ptr(), M, N, _st(x)), "tdx_layernorm_bwd")
        return gx, dg, db, None


class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kind):
        x = _prep(x)
        out = torch.empty_like(x)
        check(lib.tdx_act_fwd(x.data_ptr(), out.data_ptr(), x.numel(), kind, _st(x)), "tdx_act_fwd")
        ctx.save_for_backward(x)
        ctx.kind = kind
        return out

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        gy = _prep(gy)
        gx = torch.empty_like(x)
        check(lib.tdx_act_bwd(gy.data_ptr(), x.data_ptr(), gx.data_ptr(), x.numel(), ctx.kind, _st(x)), "tdx_act_bwd")
        return gx, None


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, group, seed, offset):
        x = _prep(x)
        out = torch.empty_like(x)
        check(lib.tdx_dropout(x.data_ptr(), out.data_ptr(), x.numel(), group, p, seed, offset, _st(x)), "tdx_dropout")
        ctx.args = (p, group, seed, offset)
        return out

    @staticmethod
    def backward(ctx, gy):
        gy = _prep(gy)
        gx = torch.empty_like(gy)
        p, group, seed, offset = ctx.args
        check(lib.tdx_dropout(gy.data_ptr(), gx.data_ptr(), gy.numel(), group, p, seed, offset, _st(gy)), "tdx_dropout")
        return gx, None, None, None, None


class _Add(torch.autograd.Function):
    """a (M,N) + b, b either (M,N) or a broadcast row (N,)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _prep(a), _prep(b)
        out = torch.empty_like(a)
        period = 0 if b.numel() == a.numel() else b.numel()
        check(lib.tdx_add(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), period, _st(a)), "tdx_add")
        ctx.row = period != 0
        ctx.b_shape = b.shape
        return out

    @staticmethod
    def backward(ctx, g):
        if not ctx.row:
            return g, g
        g = _prep(g)
        n = g.shape[-1]
        gb = torch.empty(n, dtype=torch.float32, device=g.device)
        # column sums through the Linear backward's bias-gradient path
        check(lib.tdx_linear_bwd(g.data_ptr(), n, None, 0, None, None, 0, None, gb.data_ptr(), g.numel() // n, n, 1,
                                 _st(g)), "tdx_linear_bwd(db)")
        return g, gb.view(ctx.b_shape)


class _Embedding(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, idx):
        w = _prep(w)
        idx = idx.contiguous().to(torch.int64)
        M, N = idx.shape[0], w.shape[1]
        out = torch.empty(M, N, dtype=torch.float32, device=w.device)
        check(lib.tdx_embedding_fwd(w.data_ptr(), idx.data_ptr(), out.data_ptr(), M, N, _st(w)), "tdx_embedding_fwd")
        ctx.save_for_backward(idx)
        ctx.num = w.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        g = _prep(g)
        M, N = g.shape
        dw = torch.empty(ctx.num, N, dtype=torch.float32, device=g.device)
        check(lib.tdx_embedding_bwd(g.data_ptr(), idx.data_ptr(), dw.data_ptr(), M, N, ctx.num, _st(g)),
              "tdx_embedding_bwd")
        return dw, None


def linear(x, w, b):
    return _Linear.apply(x, w, b)


def layer_norm(x, g, b, eps=1e-5):
    return _LayerNorm.apply(x, g, b, eps)


def silu(x):
    return _Act.apply(x, 0)


def gelu(x):
    return _Act.apply(x, 1)


def dropout(x, p, group, seed, offset):
    return _Dropout.apply(x, float(p), int(group), int(seed), int(offset))


def add(a, b):
    return _Add.apply(a, b)


def embedding(w, idx):
    return _Embedding.apply(w, idx)

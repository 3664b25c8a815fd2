"""Per-operator Python entry points over the C ABI (include/omnitok.h).

Every function takes torch tensors that live on the MI355X ("cuda" device of PyTorch-ROCm), passes
raw device pointers + the current HIP stream to libomnitok.so and returns torch tensors.  PyTorch
is used for memory and streams only; all arithmetic is in the hand-written HIP kernels.  There is
no CPU path: tensors on the CPU raise.

The operators are also registered as PyTorch custom ops in the `omnitok::` namespace
(torch.ops.omnitok.vq_argmin, .attn_spatial, .attn_window, .attn_temporal, .linear, .peg3d,
.layernorm) so that callers who compose their own graphs get schema-checked ops.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib
from ._lib import check

GEMM_BIAS, GEMM_RESIDUAL, GEMM_GEGLU, GEMM_LEAKY = 1, 2, 4, 8


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _req(t: torch.Tensor, name: str, dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor")
    if t.device.type != "cuda":
        raise RuntimeError(f"{name}: tensor is on {t.device}; omnitokenizer_amd has no CPU path "
                           "(the HIP kernels are the product)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    if t.device.index is not None and t.device.index != torch.cuda.current_device():
        raise RuntimeError(f"{name}: tensor is on {t.device} but the current device is cuda:{torch.cuda.current_device()}; "
                           "the kernels launch on the current device's stream -- wrap the call in "
                           "`with torch.cuda.device(tensor.device):`")
    return t


def _opt(t: Optional[torch.Tensor], name: str, numel: Optional[int] = None, dtype=torch.float32):
    """Optional operand: validated like a required one when present."""
    if t is None:
        return None
    _req(t, name, dtype)
    if numel is not None and t.numel() != numel:
        raise ValueError(f"{name}: expected {numel} elements, got {t.numel()}")
    return t


def layernorm(x, gamma, beta=None, eps=1e-5):
    x = _req(x, "x")
    rows, dim = x.numel() // x.shape[-1], x.shape[-1]
    y = torch.empty_like(x)
    _opt(beta, "beta", dim)
    check(_lib.load().omnitok_layernorm(_p(x), _p(_req(gamma, "gamma")), _p(beta), _p(y), rows, dim, eps,
                                        0, 0, 0, _stream()), "layernorm")
    return y


def layernorm_transposed(x, gamma, beta, n, a, c, eps=1e-5):
    """LayerNorm of rows ordered (n, a, c) stored in (n, c, a) order (the token transpose fused into the store)."""
    x = _req(x, "x")
    dim = x.shape[-1]
    assert x.numel() == n * a * c * dim
    y = torch.empty_like(x)
    _opt(beta, "beta", dim)
    check(_lib.load().omnitok_layernorm_transposed(_p(x), _p(_req(gamma, "gamma")), _p(beta), _p(y), n, a, c, dim, eps,
                                                   _stream()), "layernorm_transposed")
    return y


def layernorm_prevq(x, gamma, beta, w, b, n, a, c, transpose=True, l2=True, eps=1e-5):
    """z [n*a*c, 8] = l2norm(LayerNorm(x) @ w.T + b), rows in (n, c, a) order when transpose (the encoder's last norm_out
    fused with pre_vq, reference omnitokenizer.py:143-148, 251-252); bit-identical to layernorm[_transposed] + pre_vq."""
    x = _req(x, "x")
    dim = x.shape[-1]
    assert x.numel() == n * a * c * dim
    z = torch.empty(n * a * c, 8, device=x.device, dtype=torch.float32)
    _opt(beta, "beta", dim)
    check(_lib.load().omnitok_layernorm_prevq(_p(x), _p(_req(gamma, "gamma")), _p(beta), _p(_req(w, "w")), _p(_req(b, "b")),
                                              _p(z), n, a, c, dim, eps, int(transpose), int(l2), _stream()), "layernorm_prevq")
    return z


def linear(x, weight, bias=None, residual=None, leaky=False):
    """y = x @ weight.T (+bias) (+leaky_relu 0.1) (+residual); weight [N, K], K % 32 == 0."""
    x = _req(x, "x")
    weight = _req(weight, "weight")
    K = x.shape[-1]
    M = x.numel() // K
    N = weight.shape[0]
    out = torch.empty(*x.shape[:-1], N, device=x.device, dtype=torch.float32)
    _opt(bias, "bias", N)
    _opt(residual, "residual", M * N)
    flags = (GEMM_BIAS if bias is not None else 0) | (GEMM_RESIDUAL if residual is not None else 0) \
        | (GEMM_LEAKY if leaky else 0)
    check(_lib.load().omnitok_gemm(_p(x), K, _p(weight), weight.shape[1], _p(bias), _p(residual), N, _p(out), N,
                                   M, N, K, flags, 0, 0, 0, _stream()), "gemm")
    return out


def row_stats(x, eps=1e-5, bounds=None, rows_per_clip=0):
    """[rows, 2] = (mean, rstd) per row of x[..., dim] (the statistics a fused-LN GEMM consumes).
    bounds: optional zeroed float32[n_clips, 2] that receives, per clip of rows_per_clip rows, upper bounds of
    max|x| and max ||x_row||_2."""
    x = _req(x, "x")
    rows, dim = x.numel() // x.shape[-1], x.shape[-1]
    st = torch.empty(rows, 2, device=x.device, dtype=torch.float32)
    if bounds is not None:
        _req(bounds, "bounds")
    check(_lib.load().omnitok_row_stats(_p(x), rows, dim, eps, _p(st), _p(bounds), int(rows_per_clip), _stream()),
          "row_stats")
    return st


def linear_x3(x, weight, bias=None, residual=None, geglu=False, ln=None, ln_cols=None):
    """y = x @ weight.T (+epilogue) with fp32 operands split in-kernel into 3 bf16 planes each (six bf16
    MFMA products, fp32 accumulate; csrc/gemm_x3.hip).  geglu: weight packed by pack_geglu_weight.
    ln = (stats, gamma, beta|None): LayerNorm fused into the A operand for output columns [0, ln_cols)."""
    x = _req(x, "x")
    weight = _req(weight, "weight")
    K = x.shape[-1]
    M = x.numel() // K
    N = weight.shape[0]
    ncol = N // 2 if geglu else N
    out = torch.empty(*x.shape[:-1], ncol, device=x.device, dtype=torch.float32)
    _opt(bias, "bias", ncol)
    _opt(residual, "residual", M * ncol)
    flags = GEMM_GEGLU if geglu else ((GEMM_BIAS if bias is not None else 0) |
                                      (GEMM_RESIDUAL if residual is not None else 0))
    st = g = b = None
    if ln is not None:
        st, g, b = ln
        _req(st, "ln stats")
        _req(g, "ln gamma")
        _opt(b, "ln beta", K)
    check(_lib.load().omnitok_gemm_x3(_p(x), K, _p(weight), weight.shape[1], _p(bias), _p(residual), ncol, _p(out),
                                      ncol, M, N, K, flags, 0, 0, 0, _p(st), _p(g), _p(b),
                                      int(ln_cols if ln_cols is not None else N), None, 0, 0, _stream()), "gemm_x3")
    return out


def h2_pack_weight(weight):
    """fp32 [N, K] -> (planes fp16 [ceil(N/64), K/32, 2, 4, 64, 8] = hi|lo of the row-scaled weight in blocks of
    64 rows x 32 k (the LDS image of the K32 kernel), scale [N])."""
    weight = _req(weight, "weight")
    N, K = weight.shape
    if K % 32:
        raise ValueError("h2_pack_weight: K must be a multiple of 32")
    planes = torch.empty((N + 63) // 64, K // 32, 2, 4, 64, 8, device=weight.device, dtype=torch.float16)
    scale = torch.empty(N, device=weight.device, dtype=torch.float32)
    check(_lib.load().omnitok_h2_pack_weight(_p(weight), K, N, K, _p(planes), _p(scale), _stream()), "h2_pack_weight")
    return planes, scale


def linear_h2(x, packed, a_bound, bias=None, residual=None, geglu=False, a_bound_dev=None, ln=None, ln_cols=None,
              ln_bound=0.0, a_bound_stride=1, rows_per_clip=0):
    """y = x @ weight.T (+epilogue) from 2-way fp16 splits of both operands (three fp16 MFMA products, fp32
    accumulate; csrc/gemm_h2.hip).  packed = h2_pack_weight(weight); a_bound >= max|x| (times *a_bound_dev)."""
    x = _req(x, "x")
    planes, scale = packed
    K = x.shape[-1]
    M = x.numel() // K
    N = scale.shape[0]
    ncol = N // 2 if geglu else N
    out = torch.empty(*x.shape[:-1], ncol, device=x.device, dtype=torch.float32)
    _opt(bias, "bias", ncol)
    _opt(residual, "residual", M * ncol)
    _opt(a_bound_dev, "a_bound_dev")
    flags = GEMM_GEGLU if geglu else ((GEMM_BIAS if bias is not None else 0) |
                                      (GEMM_RESIDUAL if residual is not None else 0))
    st = g = b = None
    if ln is not None:
        st, g, b = ln
        _req(st, "ln stats")
        _req(g, "ln gamma")
        _opt(b, "ln beta", K)
    check(_lib.load().omnitok_gemm_h2(_p(x), K, _p(planes), _p(scale), _p(bias), _p(residual), ncol, _p(out), ncol,
                                      M, N, K, flags, 0, 0, 0, float(a_bound), _p(a_bound_dev), int(a_bound_stride),
                                      int(rows_per_clip), _p(st), _p(g), _p(b),
                                      int(ln_cols if ln_cols is not None else N), float(ln_bound), None, 0, 0,
                                      _stream()), "gemm_h2")
    return out


def linear_h2_vpack(x, packed, a_bound, ln, ln_cols, ln_bound, v_col0, n_tokens, heads, v_bound, a_bound_dev=None,
                    a_bound_stride=1, rows_per_clip=0, v_bound_dev=None, v_bound_stride=1):
    """linear_h2 (fused LayerNorm on the first ln_cols columns) whose columns [v_col0, N) go straight into the packed
    fp16 hi|lo V planes of attn_spatial_h2 (omnitok_gemm_h2_vpack).  Returns (out[M, v_col0], v_planes)."""
    x = _req(x, "x")
    planes, scale = packed
    K = x.shape[-1]
    M = x.numel() // K
    N = scale.shape[0]
    out = torch.empty(M, v_col0, device=x.device, dtype=torch.float32)
    vp = torch.empty(M * heads * 64, device=x.device, dtype=torch.int32)
    st, g, b = ln
    _req(st, "ln stats")
    _req(g, "ln gamma")
    _opt(b, "ln beta", K)
    _opt(a_bound_dev, "a_bound_dev")
    _opt(v_bound_dev, "v_bound_dev")
    check(_lib.load().omnitok_gemm_h2_vpack(_p(x), K, _p(planes), _p(scale), None, None, 0, _p(out), v_col0, M, N, K, 0, 0,
                                            0, 0, float(a_bound), _p(a_bound_dev), int(a_bound_stride),
                                            int(rows_per_clip), _p(st), _p(g), _p(b), int(ln_cols), float(ln_bound), None,
                                            0, 0, _p(vp), int(v_col0), int(n_tokens), int(heads), float(v_bound),
                                            _p(v_bound_dev), int(v_bound_stride), _stream()), "gemm_h2_vpack")
    return out, vp


# ---- plane x plane GEMM (csrc/gemm_pl.h): both operands as fp16 hi|lo planes in 64-row x 32-k blocks ------------------

def _pad256(n: int) -> int:
    return (n + 255) // 256 * 256


def pl_pack_weight(weight):
    """nn.Linear weight [N, K] -> (planes, row scales) for linear_pl (rows permuted inside groups of 32, padded to 256)."""
    w = _req(weight, "weight")
    N, K = w.shape
    planes = torch.empty(_pad256(N) * K, device=w.device, dtype=torch.int32)
    scale = torch.empty(N, device=w.device, dtype=torch.float32)
    check(_lib.load().omnitok_pl_pack_weight(_p(w), K, N, K, _pad256(N), _p(planes), _p(scale), _stream()), "pl_pack_weight")
    return planes, scale


def pl_pack_rows(x, static_bound=None):
    """Activation rows [M, K] -> (planes, per-row scales or None).  static_bound: one power-of-two scale from an upper
    bound of |x| (the consumer then takes a_scale_const = pl_unscale(bound)); default: one scale per row."""
    x = _req(x, "x")
    M, K = x.shape
    planes = torch.empty(_pad256(M) * K, device=x.device, dtype=torch.int32)
    scales = None if static_bound else torch.empty(max(M, 1), device=x.device, dtype=torch.float32)
    check(_lib.load().omnitok_pl_pack_rows(_p(x), K, M, K, _pad256(M), _p(planes), _p(scales), float(static_bound or 0.0),
                                           _stream()), "pl_pack_rows")
    return planes, scales


def pl_unscale(bound: float) -> float:
    return float(_lib.load().omnitok_pl_unscale(float(bound)))


def pl_unpack_planes(planes, M, K):
    """hi + lo of a plane buffer as fp64 [M, K] (test helper; plain torch indexing on the GPU tensor)."""
    h = planes.view(torch.float16).view(-1, K // 32, 2, 4, 64, 8)  # [row block][k block][plane][k group][row][8]
    v = h[:, :, 0].double() + h[:, :, 1].double()                  # [rb][kb][kg][row][8]
    v = v.permute(0, 3, 1, 2, 4).reshape(-1, K)                    # [rb, row][kb, kg, 8]
    return v[:M]


def stats_pack(x, eps=1e-5, center=False):
    """Row statistics and planes of x [M, K] in one pass (K % 256 == 0): (planes, row scales, stats [M, 2] = mean, rstd).
    center: the planes hold x - mean (the operand of linear_pl(fold=...))."""
    x = _req(x, "x")
    M, K = x.shape
    planes = torch.empty(_pad256(M) * K, device=x.device, dtype=torch.int32)
    scales = torch.empty(max(M, 1), device=x.device, dtype=torch.float32)
    stats = torch.empty(max(M, 1), 2, device=x.device, dtype=torch.float32)
    check(_lib.load().omnitok_stats_pack(_p(x), M, K, eps, int(bool(center)), _p(planes), _pad256(M), _p(scales), _p(stats), None,
                                         0, _stream()), "stats_pack")
    return planes, scales, stats


def stats_pack_temporal(x, nseq, eps=1e-5):
    """stats_pack (centred) of x [nseq * 5, K] with the OUTPUT rows in the fused temporal stage's order: row of (sequence s,
    step t) = ((s // 64) * 2 + (s % 64) // 32) * 160 + t * 32 + s % 32; ceil(nseq / 64) * 320 rows (planes, scales, stats)."""
    x = _req(x, "x")
    M, K = x.shape
    assert M == nseq * 5
    rows = (nseq + 63) // 64 * 320
    planes = torch.empty(rows * K, device=x.device, dtype=torch.int32)
    scales = torch.empty(rows, device=x.device, dtype=torch.float32)
    stats = torch.empty(rows, 2, device=x.device, dtype=torch.float32)
    check(_lib.load().omnitok_stats_pack_temporal(_p(x), nseq, K, eps, _p(planes), _p(scales), _p(stats), None, 0, _stream()),
          "stats_pack_temporal")
    return planes, scales, stats


def temporal_fused(planes, scales, stats, nseq, heads, wqk_packed, wv_packed, fold_qk, fold_u_v, q_scale, k_scale, scale, v_bound,
                   alibi=None):
    """The fused temporal stage on stats_pack_temporal's operand (omnitok_pl_gemm epilogues 6 and 7): wqk_packed = packed
    [q_h | k_h] x heads rows (q rows LayerNorm-folded), fold_qk = (fold_b, fold_u) in that order, wv_packed the V rows,
    fold_u_v their row sums.  Returns (P [nseq, heads, 40], attention output planes in token order [nseq * 5, heads * 64],
    out_scale [nseq * 5])."""
    dev = planes.device
    D = heads * 64
    M = (nseq + 63) // 64 * 320
    P = torch.zeros(nseq, heads, 40, device=dev, dtype=torch.float32)
    out_planes = torch.zeros(_pad256(nseq * 5) * D, device=dev, dtype=torch.int32)
    out_scale = torch.zeros(nseq * 5, device=dev, dtype=torch.float32)
    for epi in (6, 7):
        g = _lib.OmnitokPlGemm()
        g.a, g.a_scale, g.fold_stats = planes.data_ptr(), scales.data_ptr(), stats.data_ptr()
        g.M, g.K = M, D
        g.tp, g.t_nseq, g.t_heads = P.data_ptr(), nseq, heads
        g.t_alibi = alibi.data_ptr() if alibi is not None else None
        g.epilogue = epi
        if epi == 6:
            g.w, g.w_scale = wqk_packed[0].data_ptr(), wqk_packed[1].data_ptr()
            g.fold_b, g.fold_u = fold_qk[0].data_ptr(), fold_qk[1].data_ptr()
            g.N = 2 * D
            g.q_scale, g.k_scale, g.q_mul = q_scale.data_ptr(), k_scale.data_ptr(), float(scale)
        else:
            g.w, g.w_scale = wv_packed[0].data_ptr(), wv_packed[1].data_ptr()
            g.fold_u = fold_u_v.data_ptr()
            g.N = D
            g.out_planes, g.out_planes_k = out_planes.data_ptr(), D
            g.t_out_scale, g.v_bound, g.t_seqs_per_clip = out_scale.data_ptr(), float(v_bound), nseq
        check(_lib.load().omnitok_gemm_pl(ctypes.byref(g), _stream()), "gemm_pl")
    return P, out_planes, out_scale


def stats_pack_windows(x, gh, gw, ws=8, eps=1e-5, center=True):
    """stats_pack with the OUTPUT rows (planes, scales, stats) in window-major order (frame, window, position): the operand of the
    window-attention q|k|v plane GEMM (reference attention.py:170-188 window_partition).  x [frames * gh * gw, K] in token order."""
    x = _req(x, "x")
    M, K = x.shape
    planes = torch.empty(_pad256(M) * K, device=x.device, dtype=torch.int32)
    scales = torch.empty(max(M, 1), device=x.device, dtype=torch.float32)
    stats = torch.empty(max(M, 1), 2, device=x.device, dtype=torch.float32)
    check(_lib.load().omnitok_stats_pack_windows(_p(x), M, K, eps, int(bool(center)), _p(planes), _pad256(M), _p(scales), _p(stats),
                                                 gh, gw, ws, _stream()), "stats_pack_windows")
    return planes, scales, stats


def layernorm_planes(x, gamma, beta, bound, eps=1e-5):
    """LayerNorm(x) [M, K] as hi|lo operand planes scaled by the power of two of `bound` (consumer: a_scale_const = pl_unscale(bound))."""
    x = _req(x, "x")
    M, K = x.shape
    planes = torch.empty(_pad256(M) * K, device=x.device, dtype=torch.int32)
    check(_lib.load().omnitok_layernorm_planes(_p(x), M, K, eps, _p(_req(gamma, "gamma")), _p(beta), float(bound), _p(planes),
                                               _pad256(M), _stream()), "layernorm_planes")
    return planes


def attn_window_h2(qp, kp, vp, bias_dense, q_bound, k_bound, v_bound, Bn, gh, gw, heads, planes=False):
    """Window attention from packed operands (omnitok_attn_window_h2): fp32 [Bn * gh * gw, heads * 64] in token order, or with
    planes=True the hi|lo planes of it (scaled by the power of two of v_bound)."""
    rows = Bn * gh * gw
    if planes:
        out = torch.empty(_pad256(rows) * heads * 64, device=qp.device, dtype=torch.int32)
        check(_lib.load().omnitok_attn_window_h2(_p(qp), _p(kp), _p(vp), _p(_req(bias_dense, "bias_dense")), None, 0, _p(out),
                                                 float(q_bound), float(k_bound), float(v_bound), Bn, gh, gw, heads, _stream()),
              "attn_window_h2")
        return out
    out = torch.empty(rows, heads * 64, device=qp.device, dtype=torch.float32)
    check(_lib.load().omnitok_attn_window_h2(_p(qp), _p(kp), _p(vp), _p(_req(bias_dense, "bias_dense")), _p(out), heads * 64, None,
                                             float(q_bound), float(k_bound), float(v_bound), Bn, gh, gw, heads, _stream()),
          "attn_window_h2")
    return out


def fold_layernorm_weight(weight, gamma, beta=None, rows_fold=None):
    """(w', b, u) for linear_pl(fold=...) on a centred operand: rows < rows_fold of w multiplied by gamma with b = w beta,
    u = row sums of the other rows (what the engine prepares once per layer)."""
    rows_fold = weight.shape[0] if rows_fold is None else rows_fold
    w2 = weight.clone()
    w2[:rows_fold] = weight[:rows_fold] * gamma[None, :]
    b = torch.zeros(weight.shape[0], device=weight.device)
    if beta is not None:
        b[:rows_fold] = (weight[:rows_fold].double() @ beta.double()).float()
    u = torch.zeros(weight.shape[0], device=weight.device)
    u[rows_fold:] = weight[rows_fold:].double().sum(1).float()
    return w2.contiguous(), b.contiguous(), u.contiguous()


def linear_pl(a_planes, w_packed, M, N, K, a_scale=None, a_scale_const=0.0, bias=None, residual=None, epilogue=0,
              out_bound=0.0, ln=None, a2=None, a_split_n=0, c_split_n=0, cfg=0, fold=None, attn=None, k_valid=0, a_rows=None,
              unpatch=None):
    """c = a . w^T from plane operands.  epilogue 0: fp32 [M, N] (+ bias, + residual; c_split_n: two outputs);
    1: GEGLU -> hidden planes [M, N / 2]; 2: (fp32, planes of LayerNorm(c)) with ln = (gamma, beta or None, eps).
    a2 = (planes, scales or None, const): second activation operand for output columns >= a_split_n.
    fold = (stats, b or None, u or None, fold_cols): LayerNorm folded into the weight, operand = centred planes of
    stats_pack(center=True): columns < fold_cols get rstd * acc + b, the others acc + mean * u (epilogues 0, 3, 4).
    epilogue 4 (packed Q | K) and 3 (packed V, N = heads * 64) take attn = dict(n_tokens, heads, q_scale, k_scale, cos, sin,
    q_mul, q_bound, k_bound, v_bound) and return int32 buffers in the layout of attn_pack.
    k_valid: leading k that are not zero padding (the K loop stops there).  a_rows = (rpg, gstride, goff): GEMM row m reads plane
    row (m // rpg) * gstride + goff + m % rpg.  epilogue 5 takes unpatch = dict(video, f0, t, pt, p): the fp32 result (+ bias) is
    scattered into `video` [B, C, F, H, W] in place (un-patchify store) and the video is returned."""
    g = _lib.OmnitokPlGemm()
    dev = a_planes.device
    g.a = a_planes.data_ptr()
    g.a_scale = a_scale.data_ptr() if a_scale is not None else None
    g.a_scale_const = float(a_scale_const)
    if a2 is not None:
        g.a2 = a2[0].data_ptr()
        g.a2_scale = a2[1].data_ptr() if a2[1] is not None else None
        g.a2_scale_const = float(a2[2])
        g.a_split_n = int(a_split_n)
    g.w = w_packed[0].data_ptr()
    g.w_scale = w_packed[1].data_ptr()
    g.bias = bias.data_ptr() if bias is not None else None
    outs = []
    if epilogue in (0, 2):
        if c_split_n:
            c = torch.empty(M, c_split_n, device=dev, dtype=torch.float32)
            c2 = torch.empty(M, N - c_split_n, device=dev, dtype=torch.float32)
            g.c, g.ldc, g.c2, g.ldc2, g.c_split_n = c.data_ptr(), c_split_n, c2.data_ptr(), N - c_split_n, c_split_n
            outs = [c, c2]
        else:
            c = torch.empty(M, N, device=dev, dtype=torch.float32)
            g.c, g.ldc = c.data_ptr(), N
            outs = [c]
        if residual is not None:
            g.residual, g.ldr = residual.data_ptr(), residual.stride(0)
    if epilogue in (1, 2):
        ko = N // 2 if epilogue == 1 else N
        op = torch.empty(_pad256(M) * ko, device=dev, dtype=torch.int32)
        g.out_planes, g.out_planes_k, g.out_bound = op.data_ptr(), ko, float(out_bound)
        outs.append(op)
    if fold is not None:
        g.fold_stats = fold[0].data_ptr()
        g.fold_b = fold[1].data_ptr() if fold[1] is not None else None
        g.fold_u = fold[2].data_ptr() if fold[2] is not None else None
        g.fold_cols = int(fold[3])
    if epilogue in (3, 4):
        heads, nt = attn["heads"], attn["n_tokens"]
        g.n_tokens, g.heads = nt, heads
        if epilogue == 4:
            qp = torch.empty(M * heads * 64, device=dev, dtype=torch.int32)
            kp = torch.empty(M * heads * 64, device=dev, dtype=torch.int32)
            g.qp, g.kp, g.qk_k0 = qp.data_ptr(), kp.data_ptr(), heads * 64
            if attn.get("q_scale") is not None:   # None: no l2 normalisation, no learned scale (window attention)
                g.q_scale, g.k_scale = attn["q_scale"].data_ptr(), attn["k_scale"].data_ptr()
            if attn.get("cos") is not None:
                g.rope_cos, g.rope_sin = attn["cos"].data_ptr(), attn["sin"].data_ptr()
            g.q_mul, g.q_bound, g.k_bound = float(attn["q_mul"]), float(attn["q_bound"]), float(attn["k_bound"])
            outs = [qp, kp]
        else:
            vp = torch.empty(M * heads * 64, device=dev, dtype=torch.int32)
            g.vp, g.v_bound = vp.data_ptr(), float(attn["v_bound"])
            outs = [vp]
    if epilogue == 2:
        g.ln_gamma = ln[0].data_ptr()
        g.ln_beta = ln[1].data_ptr() if ln[1] is not None else None
        g.ln_eps = float(ln[2])
    g.k_valid = int(k_valid)
    if a_rows is not None:
        g.a_rpg, g.a_gstride, g.a_goff = (int(v) for v in a_rows)
    if epilogue == 5:
        video = _req(unpatch["video"], "video")
        g.c = video.data_ptr()
        g.up_C, g.up_F, g.up_H, g.up_W = video.shape[1], video.shape[2], video.shape[3], video.shape[4]
        g.up_f0, g.up_t, g.up_pt, g.up_p = int(unpatch["f0"]), int(unpatch["t"]), int(unpatch["pt"]), int(unpatch["p"])
        outs = [video]
    g.epilogue, g.M, g.N, g.K, g.cfg = epilogue, M, N, K, cfg
    check(_lib.load().omnitok_gemm_pl(ctypes.byref(g), _stream()), "gemm_pl")
    return outs[0] if len(outs) == 1 else tuple(outs)


def attn_spatial_h2_planes(packed, bounds, Bn, N, heads, bias_table=None, gh=0, gw=0, v_bound_dev=None, v_bound_stride=1,
                           seq_per_clip=0):
    """attn_spatial_h2 with the output as (planes, row scales) of K = heads * 64 (the to_out GEMM's operand)."""
    planes = torch.empty(_pad256(Bn * N) * heads * 64, device=packed.device, dtype=torch.int32)
    scales = torch.empty(Bn * N, device=packed.device, dtype=torch.float32)
    check(_lib.load().omnitok_attn_spatial_h2_planes(_p(packed[0]), _p(packed[1]), _p(packed[2]), None, 0, _p(planes),
                                                     _p(scales), Bn, N, heads, bounds[0], bounds[1], bounds[2],
                                                     _p(v_bound_dev), v_bound_stride, seq_per_clip, _p(bias_table), gh, gw,
                                                     _stream()), "attn_spatial_h2_planes")
    return planes, scales


def attn_window_planes(qkv, bias_dense, Bn, gh, gw, heads, out_bound):
    qkv = _req(qkv, "qkv")
    planes = torch.empty(_pad256(qkv.shape[0]) * heads * 64, device=qkv.device, dtype=torch.int32)
    check(_lib.load().omnitok_attn_window_planes(_p(qkv), qkv.shape[1], _p(_req(bias_dense, "bias_dense")), None, 0,
                                                 _p(planes), float(out_bound), Bn, gh, gw, heads, _stream()),
          "attn_window_planes")
    return planes


def attn_temporal_planes(q, k, v, cols, T, heads, q_scale, k_scale, causal, v_bound, alibi=None, scale=8.0,
                         v_bound_dev=None, v_bound_stride=1, cols_per_clip=0):
    assert k.stride(0) == v.stride(0)
    planes = torch.empty(_pad256(q.shape[0]) * heads * 64, device=q.device, dtype=torch.int32)
    scales = torch.empty(q.shape[0], device=q.device, dtype=torch.float32)
    check(_lib.load().omnitok_attn_temporal_planes(_p(q), q.stride(0), _p(k), _p(v), k.stride(0), None, 0, _p(planes),
                                                   _p(scales), float(v_bound), _p(v_bound_dev), v_bound_stride,
                                                   cols_per_clip, cols, T, heads, _p(q_scale), _p(k_scale), scale,
                                                   int(bool(causal)), _p(alibi), _stream()), "attn_temporal_planes")
    return planes, scales


def pack_geglu_weight(w1, inner_pad):
    w1 = _req(w1, "w1")
    inner, K = w1.shape[0] // 2, w1.shape[1]
    out = torch.empty(2 * inner_pad, K, device=w1.device, dtype=torch.float32)
    check(_lib.load().omnitok_pack_geglu_weight(_p(w1), inner, K, inner_pad, _p(out), _stream()), "pack_geglu")
    return out


def linear_geglu(x, w1_packed):
    """gelu(gate) * value of x @ w1.T with w1 packed by pack_geglu_weight -> [.., inner_pad]."""
    x = _req(x, "x")
    K = x.shape[-1]
    M = x.numel() // K
    Np = w1_packed.shape[0]
    out = torch.empty(*x.shape[:-1], Np // 2, device=x.device, dtype=torch.float32)
    check(_lib.load().omnitok_gemm(_p(x), K, _p(_req(w1_packed, "w1_packed")), K, None, None, 0, _p(out), Np // 2,
                                   M, Np, K, GEMM_GEGLU, 0, 0, 0, _stream()), "gemm_geglu")
    return out


def patchify_ln(video, f0, t, pt, p, gamma=None, beta=None, eps=1e-5, ldo=0):
    """gamma/beta None: plain im2col rows (no LayerNorm).  ldo: output row stride (0 = dense);
    the padding columns are zero."""
    video = _req(video, "video")
    B, C, F, H, W = video.shape
    dim = C * pt * p * p
    out = torch.empty(B * t * (H // p) * (W // p), ldo or dim, device=video.device, dtype=torch.float32)
    check(_lib.load().omnitok_patchify_ln(_p(video), B, C, F, H, W, f0, t, pt, p,
                                          None if gamma is None else _p(_req(gamma, "gamma")),
                                          None if beta is None else _p(_req(beta, "beta")), eps, _p(out), ldo,
                                          _stream()), "patchify_ln")
    return out


def unpatchify(tok, video, f0, t, pt, p):
    """Scatters tok rows into video[B,C,F,H,W] (in place) and returns video."""
    B, C, F, H, W = video.shape
    check(_lib.load().omnitok_unpatchify(_p(_req(tok, "tok")), B, C, F, H, W, f0, t, pt, p, _p(_req(video, "video")),
                                         _stream()), "unpatchify")
    return video


def pack_peg_weight(w):
    w = _req(w, "w")
    D = w.shape[0]
    out = torch.empty(27, D, device=w.device, dtype=torch.float32)
    check(_lib.load().omnitok_pack_peg_weight(_p(w), D, _p(out), _stream()), "pack_peg")
    return out


def peg3d(x, w27, bias, shape, causal):
    """PEG(x) + x on the raw buffer viewed as [B,T,H,W,D]."""
    x = _req(x, "x")
    B, T, H, W = shape
    D = x.shape[-1]
    assert x.numel() == B * T * H * W * D
    y = torch.empty_like(x)
    check(_lib.load().omnitok_peg3d(_p(x), _p(_req(w27, "w27")), _p(_req(bias, "bias")), _p(y), B, T, H, W, D,
                                    int(bool(causal)), _stream()), "peg3d")
    return y


def transpose_tokens(x, B, A, C):
    x = _req(x, "x")
    D = x.shape[-1]
    y = torch.empty_like(x)
    check(_lib.load().omnitok_transpose_tokens(_p(x), _p(y), B, A, C, D, _stream()), "transpose_tokens")
    return y


def rope_table(n_tokens, dim_head=64, theta=10000.0):
    cos = torch.empty(n_tokens, dim_head // 2, dtype=torch.float32)
    sin = torch.empty_like(cos)
    check(_lib.load().omnitok_rope_table(n_tokens, dim_head, theta, ctypes.c_void_p(cos.data_ptr()),
                                         ctypes.c_void_p(sin.data_ptr())), "rope_table")
    return cos, sin


def qk_prep_(q, k, n_tokens, heads, q_scale, k_scale, cos=None, sin=None, scale=8.0):
    """In place on q [rows, heads*64] and k (a [rows, heads*64] view with row stride k.stride(0))."""
    assert q.dim() == 2 and k.dim() == 2 and q.stride(1) == 1 and k.stride(1) == 1
    _opt(cos, "cos", n_tokens * 32)
    _opt(sin, "sin", n_tokens * 32)
    check(_lib.load().omnitok_qk_prep(_p(q), q.stride(0), _p(k), k.stride(0), q.shape[0], n_tokens, heads, _p(cos),
                                      _p(sin), _p(_req(q_scale, "q_scale")), _p(_req(k_scale, "k_scale")), scale,
                                      _stream()), "qk_prep")


def attn_spatial(q, k, v, Bn, N, heads, bias_table=None, gh=0, gw=0):
    """q [Bn*N, heads*64]; k, v: [Bn*N, heads*64] views sharing one row stride."""
    assert k.stride(0) == v.stride(0) and q.stride(1) == 1
    out = torch.empty(q.shape[0], heads * 64, device=q.device, dtype=torch.float32)
    check(_lib.load().omnitok_attn_spatial(_p(q), q.stride(0), _p(k), _p(v), k.stride(0), _p(out), heads * 64, Bn, N,
                                           heads, _p(bias_table), gh, gw, _stream()), "attn_spatial")
    return out


def attn_pack(q, k, v, n_tokens, heads, q_scale, k_scale, cos=None, sin=None, scale=8.0, v_bound=None,
              v_bound_dev=None, v_bound_stride=1, rows_per_clip=0):
    """fp16-split attention operands (csrc/attn_h2.hip): RoPE + l2norm + scales as qk_prep_, then q, k, v as hi|lo
    planes in MFMA fragment order.  Returns (packed, bounds) for attn_spatial_h2.  v_bound: upper bound of |v|
    (default: measured max|v|), multiplied per clip by v_bound_dev[v_bound_stride * clip] when given."""
    assert q.dim() == 2 and k.dim() == 2 and q.stride(1) == 1 and k.stride(1) == 1 and k.stride(0) == v.stride(0)
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        if t.device.type != "cuda" or t.dtype != torch.float32:
            raise RuntimeError(f"{n}: expected a float32 CUDA tensor (no CPU path)")
    q_scale, k_scale = _req(q_scale, "q_scale"), _req(k_scale, "k_scale")
    rows = q.shape[0]
    q_bound = 1.01 * scale * float(q_scale.abs().max())
    k_bound = 1.01 * float(k_scale.abs().max())
    if v_bound is None:
        v_bound = 1.01 * float(v.abs().max()) if v_bound_dev is None else 1.01
    packed = torch.empty(3, rows * heads * 64, device=q.device, dtype=torch.int32)
    _opt(v_bound_dev, "v_bound_dev")
    _opt(cos, "cos", n_tokens * 32)
    _opt(sin, "sin", n_tokens * 32)
    check(_lib.load().omnitok_attn_pack(_p(q), q.stride(0), _p(k), _p(v), k.stride(0), rows, n_tokens, heads, _p(cos),
                                        _p(sin), _p(q_scale), _p(k_scale), scale, q_bound, k_bound, v_bound,
                                        _p(v_bound_dev), v_bound_stride, rows_per_clip, _p(packed[0]), _p(packed[1]),
                                        _p(packed[2]), _stream()), "attn_pack")
    return packed, (q_bound, k_bound, v_bound)


def attn_spatial_h2(packed, bounds, Bn, N, heads, bias_table=None, gh=0, gw=0, v_bound_dev=None, v_bound_stride=1,
                    seq_per_clip=0):
    """softmax(q k^T [+ bias]) v on the fp16 matrix cores from the operands of attn_pack -> [Bn*N, heads*64]."""
    out = torch.empty(Bn * N, heads * 64, device=packed.device, dtype=torch.float32)
    check(_lib.load().omnitok_attn_spatial_h2(_p(packed[0]), _p(packed[1]), _p(packed[2]), _p(out), heads * 64, Bn, N,
                                              heads, bounds[0], bounds[1], bounds[2], _p(v_bound_dev), v_bound_stride,
                                              seq_per_clip, _p(bias_table), gh, gw, _stream()), "attn_spatial_h2")
    return out


def attn_window(qkv, bias_dense, Bn, gh, gw, heads):
    qkv = _req(qkv, "qkv")
    out = torch.empty(qkv.shape[0], heads * 64, device=qkv.device, dtype=torch.float32)
    check(_lib.load().omnitok_attn_window(_p(qkv), qkv.shape[1], _p(_req(bias_dense, "bias_dense")), _p(out),
                                          heads * 64, Bn, gh, gw, heads, _stream()), "attn_window")
    return out


def attn_temporal(q, k, v, cols, T, heads, q_scale, k_scale, causal, alibi=None, scale=8.0):
    assert k.stride(0) == v.stride(0)
    out = torch.empty(q.shape[0], heads * 64, device=q.device, dtype=torch.float32)
    check(_lib.load().omnitok_attn_temporal(_p(q), q.stride(0), _p(k), _p(v), k.stride(0), _p(out), heads * 64, cols,
                                            T, heads, _p(q_scale), _p(k_scale), scale, int(bool(causal)), _p(alibi),
                                            _stream()), "attn_temporal")
    return out


def pre_vq(x, w, b, l2=True):
    x = _req(x, "x")
    n, D = x.numel() // x.shape[-1], x.shape[-1]
    z = torch.empty(*x.shape[:-1], 8, device=x.device, dtype=torch.float32)
    check(_lib.load().omnitok_pre_vq(_p(x), _p(_req(w, "w")), _p(_req(b, "b")), _p(z), n, D, 8, int(l2), _stream()),
          "pre_vq")
    return z


def vq_prepare(codebook):
    codebook = _req(codebook, "codebook")
    n_codes, cdim = codebook.shape
    packed = torch.empty(n_codes * 8, device=codebook.device, dtype=torch.float32)
    ee = torch.empty(n_codes, device=codebook.device, dtype=torch.float32)
    check(_lib.load().omnitok_vq_prepare(_p(codebook), n_codes, cdim, _p(packed), _p(ee), _stream()), "vq_prepare")
    return packed, ee


def vq_screen_prepare(codebook, ee):
    """fp16 screening fragments + norm bounds for vq_argmin_screened (include/omnitok.h omnitok_vq_screen_prepare)."""
    codebook = _req(codebook, "codebook")
    n_codes, cdim = codebook.shape
    screen = torch.empty(n_codes * 8 + 4, device=codebook.device, dtype=torch.float32)
    check(_lib.load().omnitok_vq_screen_prepare(_p(codebook), _p(_req(ee, "ee")), n_codes, cdim, _p(screen), _stream()),
          "vq_screen_prepare")
    return screen


def vq_argmin_screened(z, codebook, prepared=None, screen=None):
    """ids[n] = nearest code of z[n] (int64) through the fp16-screened search: bit-identical to vq_argmin."""
    z = _req(z, "z")
    packed, ee = prepared if prepared is not None else vq_prepare(codebook)
    if screen is None:
        screen = vq_screen_prepare(codebook, ee)
    n = z.numel() // 8
    ids = torch.empty(z.shape[:-1], device=z.device, dtype=torch.int64)
    check(_lib.load().omnitok_vq_argmin_screened(_p(z), _p(packed), _p(ee), _p(screen), n, codebook.shape[0], _p(ids),
                                                 _stream()), "vq_argmin_screened")
    return ids


def vq_argmin(z, codebook, prepared=None):
    """ids[n] = nearest code of z[n] (int64), bit-exact with reference modules/codebook.py:82-86."""
    z = _req(z, "z")
    packed, ee = prepared if prepared is not None else vq_prepare(codebook)
    n = z.numel() // 8
    ids = torch.empty(z.shape[:-1], device=z.device, dtype=torch.int64)
    check(_lib.load().omnitok_vq_argmin(_p(z), _p(packed), _p(ee), n, ee.numel(), _p(ids), _stream()), "vq_argmin")
    return ids


def vq_argmax_cos(z, embed, prepared=None):
    """ids[n] = first argmax_c z[n] . embed[c] (int64): the external cosine-similarity codebook, bit-exact with
    reference quantizer/vector_quantize_pytorch.py:646-650.  z, embed rows unit-norm."""
    z = _req(z, "z")
    packed, _ = prepared if prepared is not None else vq_prepare(embed)
    n = z.numel() // 8
    ids = torch.empty(z.shape[:-1], device=z.device, dtype=torch.int64)
    check(_lib.load().omnitok_vq_argmax_cos(_p(z), _p(packed), n, embed.shape[0], _p(ids), _stream()), "vq_argmax_cos")
    return ids


def vq_argmin_cdist(z, embed, prepared=None):
    """ids[n] = first argmax_c of -cdist(z[n], embed[c]) (int64): the external EuclideanCodebook, bit-exact with
    reference quantizer/vector_quantize_pytorch.py:29-33, 463."""
    z = _req(z, "z")
    packed, ee = prepared if prepared is not None else vq_prepare(embed)
    n = z.numel() // 8
    ids = torch.empty(z.shape[:-1], device=z.device, dtype=torch.int64)
    check(_lib.load().omnitok_vq_argmin_cdist(_p(z), _p(packed), _p(ee), n, embed.shape[0], _p(ids), _stream()),
          "vq_argmin_cdist")
    return ids


def dequant_post_vq(ids, codebook, w, b):
    ids = _req(ids, "ids", torch.int64)
    D = w.shape[0]
    n = ids.numel()
    tok = torch.empty(*ids.shape, D, device=ids.device, dtype=torch.float32)
    err = torch.zeros(1, device=ids.device, dtype=torch.int32)
    check(_lib.load().omnitok_dequant_post_vq(_p(ids), _p(_req(codebook, "codebook")), codebook.shape[0], 8,
                                              _p(_req(w, "w")), _p(_req(b, "b")), _p(tok), n, D, _p(err), _stream()),
          "dequant_post_vq")
    if int(err.item()):
        raise IndexError("token id out of range")
    return tok



def dequant_table(codebook, w, b):
    """table[n_codes, D] = codebook @ w.T + b with the dequant_post_vq kernel (rows bit-identical to it)."""
    codebook, w, b = _req(codebook, "codebook"), _req(w, "w"), _req(b, "b")
    n_codes, D = codebook.shape[0], w.shape[0]
    table = torch.empty(n_codes, D, device=w.device, dtype=torch.float32)
    scratch = torch.empty(n_codes, device=w.device, dtype=torch.int64)
    check(_lib.load().omnitok_dequant_table(_p(codebook), n_codes, 8, _p(w), _p(b), _p(table), D, _p(scratch),
                                            _stream()), "dequant_table")
    return table


def gather_rows(ids, table, transpose=None):
    """tok[..., :] = table[ids[...], :]; out-of-range ids raise IndexError like F.embedding.
    transpose=(a, c): ids are ordered (n, a, c), the rows are stored in (n, c, a) order."""
    ids, table = _req(ids, "ids", torch.int64), _req(table, "table")
    D = table.shape[1]
    tok = torch.empty(*ids.shape, D, device=ids.device, dtype=torch.float32)
    err = torch.zeros(1, device=ids.device, dtype=torch.int32)
    a, c = transpose if transpose else (0, 0)
    check(_lib.load().omnitok_gather_rows_transposed(_p(ids), _p(table), table.shape[0], _p(tok), ids.numel(), a, c, D,
                                                     _p(err), _stream()), "gather_rows")
    if int(err.item()):
        raise IndexError("token id out of range")
    return tok


def token_resample(x, mode):
    """Token-grid resampling used by the pooling blocks and the deferred pools.  mode:
    "avg2d" / "max2d": x [n, gh, gw, D] -> [n, gh/2, gw/2, D]   (Pooling 'a' / 'm', attention.py:83-106)
    "up2d":            x [n, gh, gw, D] -> [n, 2gh, 2gw, D]     (nearest, omnitokenizer.py:1001)
    "avg_t":           x [B, T, S, D]   -> [B, 1+(T-1)//2, S, D] (frame 0 kept, omnitokenizer.py:909-914)
    "up_t":            x [B, T, S, D]   -> [B, 1+(T-1)*2, S, D]  (frame 0 kept, omnitokenizer.py:1103-1107)"""
    x = _req(x, "x")
    m = {"avg2d": 0, "max2d": 1, "up2d": 2, "avg_t": 3, "up_t": 4}[mode]
    D = x.shape[-1]
    if m <= 2:
        n, gh, gw = x.shape[:3]
        T = 1
        oshape = (n, gh // 2, gw // 2, D) if m < 2 else (n, 2 * gh, 2 * gw, D)
    else:
        n, T, S = x.shape[:3]
        gh, gw = S, 1
        oshape = (n, 1 + (T - 1) // 2, S, D) if m == 3 else (n, 1 + (T - 1) * 2, S, D)
    out = torch.empty(oshape, device=x.device, dtype=torch.float32)
    check(_lib.load().omnitok_token_resample(_p(x), _p(out), m, n, T, gh, gw, D, _stream()), "token_resample")
    return out


def vae_sample(x, w, b, noise=None, return_moments=False):
    """x [B, thw, D] tokens -> z [B, 8, thw] = mean + exp(0.5*clamp(logvar,-30,20))*noise (reference
    modules/vae.py:4-17; w [16, D] = mean | logvar rows).  noise [B, 8, thw] or None (mode)."""
    x = _req(x, "x")
    B, thw, D = x.shape
    z = torch.empty(B, 8, thw, device=x.device, dtype=torch.float32)
    mom = torch.empty(B, 16, thw, device=x.device, dtype=torch.float32) if return_moments else None
    check(_lib.load().omnitok_vae_sample(_p(x), _p(_req(w, "w")), _p(_req(b, "b")),
                                         None if noise is None else _p(_req(noise, "noise")), _p(z),
                                         None if mom is None else _p(mom), B, thw, D, 8, _stream()), "vae_sample")
    return (z, mom) if return_moments else z


def post_vq(z, w, b, channel_first=False):
    """tok = z . w^T + b on continuous latents: z [B, thw, 8] or (channel_first) [B, 8, thw]."""
    z = _req(z, "z")
    B = z.shape[0]
    thw = z.shape[2] if channel_first else z.shape[1]
    D = w.shape[0]
    tok = torch.empty(B, thw, D, device=z.device, dtype=torch.float32)
    check(_lib.load().omnitok_post_vq(_p(z), int(channel_first), B, thw, 8, _p(_req(w, "w")), _p(_req(b, "b")),
                                      _p(tok), D, _stream()), "post_vq")
    return tok


def vq_stats(ids, n_codes, codebook_usage, first_call, usage_sigma=0.99):
    """(batch_usage [n_codes], perplexity, avg_usage) of reference Codebook.forward; updates the
    EMA buffer codebook_usage in place."""
    ids = _req(ids, "ids", torch.int64)
    counts = torch.empty(n_codes, device=ids.device, dtype=torch.int32)
    usage = torch.empty(n_codes, device=ids.device, dtype=torch.float32)
    out2 = torch.empty(2, device=ids.device, dtype=torch.float32)
    check(_lib.load().omnitok_vq_stats(_p(ids), ids.numel(), n_codes, _p(counts), _p(usage),
                                       _p(_req(codebook_usage, "codebook_usage")), int(bool(first_call)),
                                       usage_sigma, _p(out2), _stream()), "vq_stats")
    return usage, out2[0], out2[1]


# ------------------------------------------------------------------------------------------------
# PyTorch custom-op registration (omnitok:: namespace)
# ------------------------------------------------------------------------------------------------
def _register():
    from torch.library import custom_op

    @custom_op("omnitok::vq_argmin", mutates_args=(), device_types="cuda")
    def _vq_argmin(z: torch.Tensor, codebook: torch.Tensor) -> torch.Tensor:
        return vq_argmin(z.contiguous(), codebook.contiguous())

    @_vq_argmin.register_fake
    def _(z, codebook):
        return z.new_empty(z.shape[:-1], dtype=torch.int64)

    @custom_op("omnitok::linear", mutates_args=(), device_types="cuda")
    def _linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        return linear(x.contiguous(), weight.contiguous(), bias)

    @_linear.register_fake
    def _(x, weight, bias=None):
        return x.new_empty(*x.shape[:-1], weight.shape[0])

    @custom_op("omnitok::layernorm", mutates_args=(), device_types="cuda")
    def _layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: Optional[torch.Tensor] = None) -> torch.Tensor:
        return layernorm(x.contiguous(), gamma, beta)

    @_layernorm.register_fake
    def _(x, gamma, beta=None):
        return torch.empty_like(x)

    @custom_op("omnitok::attn_spatial", mutates_args=(), device_types="cuda")
    def _attn_spatial(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, n_tokens: int, heads: int) -> torch.Tensor:
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        return attn_spatial(q, k, v, q.shape[0] // n_tokens, n_tokens, heads)

    @_attn_spatial.register_fake
    def _(q, k, v, n_tokens, heads):
        return torch.empty_like(q)

    @custom_op("omnitok::attn_window", mutates_args=(), device_types="cuda")
    def _attn_window(qkv: torch.Tensor, bias_dense: torch.Tensor, images: int, gh: int, gw: int,
                     heads: int) -> torch.Tensor:
        return attn_window(qkv.contiguous(), bias_dense.contiguous(), images, gh, gw, heads)

    @_attn_window.register_fake
    def _(qkv, bias_dense, images, gh, gw, heads):
        return qkv.new_empty(qkv.shape[0], heads * 64)

    @custom_op("omnitok::attn_temporal", mutates_args=(), device_types="cuda")
    def _attn_temporal(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, q_scale: torch.Tensor,
                       k_scale: torch.Tensor, T: int, heads: int, causal: bool) -> torch.Tensor:
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        return attn_temporal(q, k, v, q.shape[0] // T, T, heads, q_scale, k_scale, causal)

    @_attn_temporal.register_fake
    def _(q, k, v, q_scale, k_scale, T, heads, causal):
        return torch.empty_like(q)

    @custom_op("omnitok::linear_geglu", mutates_args=(), device_types="cuda")
    def _linear_geglu(x: torch.Tensor, w1_packed: torch.Tensor) -> torch.Tensor:
        return linear_geglu(x.contiguous(), w1_packed.contiguous())

    @_linear_geglu.register_fake
    def _(x, w1_packed):
        return x.new_empty(*x.shape[:-1], w1_packed.shape[0] // 2)

    @custom_op("omnitok::patchify_ln", mutates_args=(), device_types="cuda")
    def _patchify_ln(video: torch.Tensor, f0: int, t: int, pt: int, p: int, gamma: Optional[torch.Tensor] = None,
                     beta: Optional[torch.Tensor] = None) -> torch.Tensor:
        return patchify_ln(video.contiguous(), f0, t, pt, p, gamma, beta)

    @_patchify_ln.register_fake
    def _(video, f0, t, pt, p, gamma=None, beta=None):
        B, C, F, H, W = video.shape
        return video.new_empty(B * t * (H // p) * (W // p), C * pt * p * p)

    @custom_op("omnitok::unpatchify", mutates_args=("video",), device_types="cuda")
    def _unpatchify(tok: torch.Tensor, video: torch.Tensor, f0: int, t: int, pt: int, p: int) -> None:
        unpatchify(tok.contiguous(), video, f0, t, pt, p)

    @custom_op("omnitok::qk_prep", mutates_args=("q", "k"), device_types="cuda")
    def _qk_prep(q: torch.Tensor, k: torch.Tensor, n_tokens: int, heads: int, q_scale: torch.Tensor,
                 k_scale: torch.Tensor, cos: Optional[torch.Tensor] = None, sin: Optional[torch.Tensor] = None,
                 scale: float = 8.0) -> None:
        qk_prep_(q, k, n_tokens, heads, q_scale, k_scale, cos, sin, scale)

    @custom_op("omnitok::attn_spatial_h2", mutates_args=(), device_types="cuda")
    def _attn_spatial_h2(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, n_tokens: int, heads: int,
                         q_scale: torch.Tensor, k_scale: torch.Tensor, cos: Optional[torch.Tensor] = None,
                         sin: Optional[torch.Tensor] = None, scale: float = 8.0) -> torch.Tensor:
        """RoPE + l2norm + scales + softmax(q k^T) v on the fp16 matrix cores from RAW q, k, v (attn_pack +
        attn_spatial_h2 of csrc/attn_h2.hip): the engine's spatial attention as one functional operator."""
        packed, bounds = attn_pack(q, k, v, n_tokens, heads, q_scale, k_scale, cos, sin, scale)
        return attn_spatial_h2(packed, bounds, q.shape[0] // n_tokens, n_tokens, heads)

    @_attn_spatial_h2.register_fake
    def _(q, k, v, n_tokens, heads, q_scale, k_scale, cos=None, sin=None, scale=8.0):
        return q.new_empty(q.shape[0], heads * 64)

    @custom_op("omnitok::pre_vq", mutates_args=(), device_types="cuda")
    def _pre_vq(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, l2: bool = True) -> torch.Tensor:
        return pre_vq(x.contiguous(), w.contiguous(), b, l2)

    @_pre_vq.register_fake
    def _(x, w, b, l2=True):
        return x.new_empty(*x.shape[:-1], w.shape[0])

    @custom_op("omnitok::dequant_post_vq", mutates_args=(), device_types="cuda")
    def _dequant(ids: torch.Tensor, codebook: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        return dequant_post_vq(ids.contiguous(), codebook.contiguous(), w.contiguous(), b)

    @_dequant.register_fake
    def _(ids, codebook, w, b):
        return w.new_empty(*ids.shape, w.shape[0])

    @custom_op("omnitok::gather_rows", mutates_args=(), device_types="cuda")
    def _gather_rows(ids: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
        return gather_rows(ids.contiguous(), table.contiguous())

    @_gather_rows.register_fake
    def _(ids, table):
        return table.new_empty(*ids.shape, table.shape[1])

    @custom_op("omnitok::peg3d", mutates_args=(), device_types="cuda")
    def _peg3d(x: torch.Tensor, w27: torch.Tensor, bias: torch.Tensor, B: int, T: int, H: int, W: int,
               causal: bool) -> torch.Tensor:
        return peg3d(x.contiguous(), w27, bias, (B, T, H, W), causal)

    @_peg3d.register_fake
    def _(x, w27, bias, B, T, H, W, causal):
        return torch.empty_like(x)


# the operators are part of the boundary: a registration failure is an import error
_register()
CUSTOM_OPS_REGISTERED = True

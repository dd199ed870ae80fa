"""Reverse mode for the HIP operators: `torch.autograd.Function` wrappers around the C ABI.

The reference trains through `tf.GradientTape` over the whole Keras model (`emulator.py:457-484`); here every HIP
operator of the forward carries its own backward, so `loss.backward()` on an `Emulator` output reaches all parameters:

    DenseFn        keras Dense                     dX = row GEMM with the transposed kernel (HIP), dW = X^T dZ (plain GEMM)
    Conv1DFn       causal dilated Conv1D           dX = the same kernels looking AHEAD (dilation < 0) with transposed taps
    SpmmFn         NodeEdge on its support / GCN   dX = spmm on the transposed pattern, dval = uds_csr_sddmm
    GatFn          MixedGAT(GATConv)               uds_gat_backward (softmax / leaky-relu / aggregation), then Dense rules
    CumsumActFn    relu(cumsum_T(x) + res)         reverse cumulative sum
    FlowBalanceFn  post_proc_tf incidence sums     gather along the link end nodes
    SpatialLayerFn the fused spatial layer forward, backward through the unfused chain above (intermediates recomputed)

    RecurrentFn    GRU / LSTM time recurrence     back-propagation through time in one launch (uds_recurrent_backward)
    RemainderFn    dense off-support NodeEdge bias  forward and d x on the split-bf16 MFMA GEMM (uds_remainder_forward)

Weight gradients are GEMMs with a huge reduction dimension (rows) and tiny outputs: the split-K MFMA kernel uds_wgrad
(`weight_grad`; shapes it does not take fall back to a library GEMM through `torch.mm`).  Activations are differentiated
from their outputs.
"""
import torch

from . import _lib


def act_grad(y, gy, act):
    """dL/dz from dL/dy for y = act(z), using only y (every activation here is invertible enough for that)."""
    act = act or 'linear'
    if act == 'linear':
        return gy
    if act == 'relu':
        return torch.ops.aten.threshold_backward(gy, y, 0.0)        # gy where y > 0 else 0, one kernel (no mask tensor)
    if act == 'tanh':
        return gy * (1.0 - y * y)
    if act == 'sigmoid':
        return gy * y * (1.0 - y)
    if act == 'hard_sigmoid':
        return gy * 0.2 * ((y > 0) & (y < 1)).to(gy.dtype)
    raise ValueError('unknown activation %r' % (act,))


def apply_activation(x, act):
    """The layer activations as differentiable tensor ops (used where no fused kernel applies them)."""
    act = act or 'linear'
    if act == 'linear':
        return x
    if act == 'relu':
        return torch.relu(x)
    if act == 'tanh':
        return torch.tanh(x)
    if act == 'sigmoid':
        return torch.sigmoid(x)
    if act == 'hard_sigmoid':
        return torch.clamp(0.2 * x + 0.5, 0.0, 1.0)
    raise ValueError('unknown activation %r' % (act,))


def rows_matmul(x, w, precision='bf16x3'):
    """x (..., K) @ w (K, M) on the HIP row-GEMM kernels (matrix cores where the shape allows, in <= 64-column pieces)."""
    x = x.contiguous()
    K, M = w.shape
    lead = x.shape[:-1]
    x3 = x.reshape(1, -1, K)
    if precision == 'bf16x3' and K % 32 == 0:
        outs = []
        for c0 in range(0, M, 64):
            wc = w[:, c0:c0 + 64].contiguous()
            if not _lib.rowgemm_supported(K, K, wc.shape[1]):
                outs = None
                break
            outs.append(_lib.rowgemm_forward(x3, _lib.rowgemm_pack(wc), None, wc.shape[1], 'linear'))
        if outs is not None:
            out = outs[0] if len(outs) == 1 else torch.cat(outs, dim=-1)
            return out.reshape(lead + (M,))
    return _lib.dense_act(x3, w.contiguous(), None, 'linear').reshape(lead + (M,))


def weight_grad(x, gz, precision, want_bias, shift=0):
    """(dW, db) of y = x @ W + b summed over all leading axes: the HIP split-K MFMA kernel ('bf16x3'), else rocBLAS."""
    F, H = x.shape[-1], gz.shape[-1]
    if precision == 'bf16x3' and _lib.wgrad_supported(F, H, want_bias):
        x4 = x.contiguous().reshape((1, 1, -1, F)) if shift == 0 else x.contiguous()
        g4 = gz.contiguous().reshape((1, 1, -1, H)) if shift == 0 else gz.contiguous()
        return _lib.wgrad(x4, g4, shift, want_bias)
    if shift:
        T = x.shape[1]
        dw = x[:, :T - shift].reshape(-1, F).t().mm(gz[:, shift:].reshape(-1, H)) if shift < T else torch.zeros(F, H, device=x.device)
    else:
        dw = x.reshape(-1, F).t().mm(gz.reshape(-1, H))
    return dw, (gz.reshape(-1, H).sum(0) if want_bias else None)


class DenseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kernel, bias, module, act):
        xc = x.contiguous()
        fi = xc.shape[-1]
        if module.precision == 'bf16x3' and _lib.rowgemm_supported(fi, fi, module.units):
            from .layers import _packed_kernel
            y = _lib.rowgemm_forward(xc, _packed_kernel(module, kernel), bias, module.units, act)
        else:
            y = _lib.dense_act(xc, kernel, bias, act)
        ctx.save_for_backward(xc, kernel, y)
        ctx.act, ctx.precision, ctx.has_bias = act, module.precision, bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, kernel, y = ctx.saved_tensors
        gz = act_grad(y, gy.contiguous(), ctx.act)
        g2 = gz.reshape(-1, gz.shape[-1])
        dx = rows_matmul(gz, kernel.t(), ctx.precision) if ctx.needs_input_grad[0] else None
        dw, db = weight_grad(x, gz, ctx.precision, ctx.has_bias and ctx.needs_input_grad[2]) if ctx.needs_input_grad[1] else (None, None)
        if dw is None and ctx.has_bias and ctx.needs_input_grad[2]:
            db = g2.sum(0)
        return dx, dw, db, None, None


class RemainderFn(torch.autograd.Function):
    """`rest @ xs`, rest (R, M) = a dense NodeEdge bias off the incidence support (emulator.py:36-39,44), xs (S, M, h): the two
    N x E x (S h) products of a training step -- the forward and d xs = rest^T @ g -- run on the split-bf16 MFMA GEMM
    (uds_remainder_forward, the kernel of the inference path).  d rest = sum_s g_s xs_s^T has the parameter's own (R, M) shape:
    a plain library GEMM (it exists only for networks small enough to hold dense (R, M) parameters at all)."""

    @staticmethod
    def forward(ctx, rest, xs):
        xs = xs.contiguous()
        ctx.save_for_backward(rest, xs)
        return _lib.remainder_forward(_lib.remainder_pack(rest.contiguous()), tuple(rest.shape), xs)

    @staticmethod
    def backward(ctx, g):
        rest, xs = ctx.saved_tensors
        g = g.contiguous()
        dxs = drest = None
        if ctx.needs_input_grad[1]:
            rt = rest.t().contiguous()
            dxs = _lib.remainder_forward(_lib.remainder_pack(rt), tuple(rt.shape), g)
        if ctx.needs_input_grad[0]:
            R, M = rest.shape
            drest = torch.matmul(g.permute(1, 0, 2).reshape(R, -1), xs.permute(1, 0, 2).reshape(M, -1).t())
        return drest, dxs


class RecurrentFn(torch.autograd.Function):
    """The time recurrence of a 64-unit keras GRU / LSTM on the input projection xp (B, T, R, G*64): forward exact fp32
    (uds_recurrent_forward_train), backward = back-propagation through time in one launch on the matrix cores
    (uds_recurrent_backward: gates recomputed from the saved xp and h), the recurrent kernel's gradient one split-K weight-
    gradient call per gate with a time shift of one, the recurrent bias its bias row.  The Dense that made xp carries the
    rest (kernel, input bias, dx).  Reference: emulator.py:158-161 inside the GradientTape of fit_eval (:457-484)."""

    @staticmethod
    def forward(ctx, xp, recurrent_kernel, recurrent_bias, kind, precision):
        xp = xp.contiguous()
        h, c = _lib.recurrent_forward_train(xp, recurrent_kernel, recurrent_bias, kind)
        ctx.save_for_backward(xp, recurrent_kernel, recurrent_bias, h, c)
        ctx.kind, ctx.precision = kind, precision
        return h

    @staticmethod
    def backward(ctx, gh):
        xp, U, rb, h, c = ctx.saved_tensors
        G = U.shape[1] // 64
        dxp, darec = _lib.recurrent_backward(xp, _lib.recurrent_pack_bwd(U), rb, h, c, gh.contiguous(), ctx.kind)
        dU = drb = None
        if ctx.needs_input_grad[1] or (rb is not None and ctx.needs_input_grad[2]):
            want_b = rb is not None and ctx.needs_input_grad[2]
            parts = [weight_grad(h, darec[g], ctx.precision, want_b, shift=1) for g in range(G)]      # dU_g = sum_t h[t-1]^T darec_g[t]
            dU = torch.cat([p[0] for p in parts], dim=1)
            if want_b:
                drb = torch.cat([p[1] if p[1] is not None else darec[g].reshape(-1, 64).sum(0) for g, p in enumerate(parts)])
        return (dxp if ctx.needs_input_grad[0] else None), dU, drb, None, None


class Conv1DFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kernel, bias, module):
        xc = x.contiguous()
        k, f, h = kernel.shape
        if module.precision == 'bf16x3' and _lib.rowgemm_supported(k * f, f, h):
            from .layers import _packed_kernel
            y = _lib.rowgemm_forward(xc, _packed_kernel(module, kernel.reshape(k * f, h)), bias, h, module.activation, taps=k,
                                     dilation=module.dilation_rate)
        else:
            y = _lib.conv1d_causal(xc, kernel, bias, module.dilation_rate, module.activation)
        ctx.save_for_backward(xc, kernel, y)
        ctx.act, ctx.precision, ctx.dil = module.activation, module.precision, module.dilation_rate
        return y

    @staticmethod
    def backward(ctx, gy):
        x, kernel, y = ctx.saved_tensors
        k, f, h = kernel.shape
        B, T, R, _ = x.shape
        gz = act_grad(y, gy.contiguous(), ctx.act).contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            kt = kernel.transpose(1, 2).contiguous()                 # (k, h, f): tap j -> W_j^T
            if ctx.precision == 'bf16x3' and _lib.rowgemm_supported(k * h, h, f):
                dx = _lib.rowgemm_forward(gz, _lib.rowgemm_pack(kt.reshape(k * h, f)), None, f, 'linear', taps=k, dilation=-ctx.dil)
            else:
                dx = _lib.conv1d_causal(gz, kt, None, -ctx.dil, 'linear')
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(kernel)
            for j in range(k):
                dw[j], dbj = weight_grad(x, gz, ctx.precision, j == k - 1 and ctx.needs_input_grad[2], shift=(k - 1 - j) * ctx.dil)
                db = dbj if dbj is not None else db
        if db is None and ctx.needs_input_grad[2]:
            db = gz.reshape(-1, h).sum(0)
        return dx, dw, db, None


class SpmmFn(torch.autograd.Function):
    """out = A(val) @ x on a CSR pattern (no bias / activation: NodeEdge on its support)."""

    @staticmethod
    def forward(ctx, val, x, handle):
        xc = x.contiguous()
        ctx.save_for_backward(val, xc)
        ctx.handle = handle
        return _lib.csr_spmm(handle, val, xc)

    @staticmethod
    def backward(ctx, g):
        val, x = ctx.saved_tensors
        g = g.contiguous()
        dval = dx = None
        if ctx.needs_input_grad[1]:
            ht, perm = ctx.handle.transposed(g.device)
            dx = _lib.csr_spmm(ht, val[perm.long()].contiguous(), g)
        if ctx.needs_input_grad[0]:
            dval = _lib.csr_sddmm(ctx.handle, g, x)
        return dval, dx, None


class GatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xa, xb, kernel, a_self, a_nbr, bias, act, handle, precision, coef=None):
        """coef (S, nnz) or None: multiplier of the normalised attention coefficients (Spektral's attention dropout)."""
        xa = xa.contiguous()
        xb = None if xb is None else xb.contiguous()
        fin, d = xa.shape[-1] + (0 if xb is None else xb.shape[-1]), kernel.shape[-1]
        w2 = kernel.reshape(fin, d)
        fast = (precision == 'bf16x3' and xa.shape[-1] % 32 == 0 and (xb is None or xb.shape[-1] % 32 == 0) and d % 16 == 0 and
                all(_lib.rowgemm_supported(fin, 32, min(64, d - c0)) for c0 in range(0, d, 64)) and xa.shape[0] * xa.shape[1] >= 4096)
        if fast:
            # linear part on the matrix cores: 64-column blocks of the kernel written into one hx (rows read from their two
            # tensors), the attention projections as one narrow row GEMM with W a_self, W a_nbr, then the aggregation kernel
            hx = torch.empty(xa.shape[:-1] + (d,), device=xa.device, dtype=torch.float32)
            for c0 in range(0, d, 64):
                _lib.rowgemm_cat(xa, xb, _lib.rowgemm_pack(w2[:, c0:c0 + 64].contiguous()), None, min(64, d - c0), 'linear', out=hx, col0=c0)
            wa = torch.stack([w2 @ a_self.reshape(-1), w2 @ a_nbr.reshape(-1)], dim=1).contiguous()
            s2 = _lib.rowgemm_cat(xa, xb, _lib.rowgemm_pack(wa), None, 2, 'linear')
            s_self, s_nbr = s2[..., 0].contiguous(), s2[..., 1].contiguous()
            out = _lib.gat_aggregate(handle, hx, s_self, s_nbr, bias, act, coef=coef)
        else:
            out, (hx, s_self, s_nbr) = _lib.gat_forward(handle, xa, kernel, a_self, a_nbr, bias, act, xb, return_workspace=True)
            if coef is not None:
                out = _lib.gat_aggregate(handle, hx, s_self, s_nbr, bias, act, coef=coef)
        ctx.save_for_backward(xa, xb, kernel, a_self, a_nbr, out, hx, s_self, s_nbr)
        ctx.act, ctx.handle, ctx.precision, ctx.has_bias, ctx.coef = act, handle, precision, bias is not None, coef
        return out

    @staticmethod
    def backward(ctx, gout):
        xa, xb, kernel, a_self, a_nbr, out, hx, s_self, s_nbr = ctx.saved_tensors
        S, n, d = out.shape
        g = act_grad(out, gout.contiguous(), ctx.act).contiguous()
        ht, perm = ctx.handle.transposed(g.device)
        d_hx, ds_self, ds_nbr = _lib.gat_backward(ctx.handle, ht, perm, g, hx, s_self, s_nbr, a_self.reshape(-1).contiguous(),
                                                  a_nbr.reshape(-1).contiguous(), coef=ctx.coef)
        fa = xa.shape[-1]
        w2 = kernel.reshape(-1, d)
        dxa = dxb = dk = das = dan = db = None
        # d[xa | xb] = d_hx W^T, each piece straight into its own tensor (no concatenated intermediate)
        if ctx.needs_input_grad[0]:
            dxa = rows_matmul(d_hx, w2[:fa].t(), ctx.precision)
        if xb is not None and ctx.needs_input_grad[1]:
            dxb = rows_matmul(d_hx, w2[fa:].t(), ctx.precision)
        if ctx.needs_input_grad[2]:
            dk = weight_grad(xa, d_hx, ctx.precision, False)[0]
            if xb is not None:
                dk = torch.cat([dk, weight_grad(xb, d_hx, ctx.precision, False)[0]], dim=0)
            dk = dk.reshape(kernel.shape)
        if ctx.needs_input_grad[3] or ctx.needs_input_grad[4]:
            # d a_self = sum_r ds_self[r] hx[r, :], d a_nbr likewise: one (rows, 2)^T (rows, d) reduction
            da = weight_grad(torch.stack([ds_self, ds_nbr], dim=-1), hx, ctx.precision, False)[0]
            das, dan = da[0].reshape(a_self.shape), da[1].reshape(a_nbr.shape)
        if ctx.has_bias and ctx.needs_input_grad[5]:
            db = g.reshape(-1, d).sum(0)
        return dxa, dxb, dk, das, dan, db, None, None, None, None


class CumsumActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res, act):
        y = _lib.cumsum_act(x.contiguous(), None if res is None else res.contiguous(), act)
        ctx.save_for_backward(y)
        ctx.act, ctx.has_res = act, res is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        gz = act_grad(y, gy, ctx.act)
        dx = torch.flip(torch.cumsum(torch.flip(gz, dims=[1]), dim=1), dims=[1]) if ctx.needs_input_grad[0] else None
        dres = gz.sum(dim=1, keepdim=True) if ctx.has_res and ctx.needs_input_grad[1] else None
        return dx, dres, None


class DropoutFn(torch.autograd.Function):
    """keras Dropout in training mode (uds_dropout): the backward pass is the same kernel on the gradient with the same
    (seed, offset) -- the mask is recomputed, not stored."""

    @staticmethod
    def forward(ctx, x, rate, seed, offset):
        ctx.cfg = (rate, seed, offset)
        return _lib.dropout(x, rate, seed, offset)

    @staticmethod
    def backward(ctx, gy):
        rate, seed, offset = ctx.cfg
        return _lib.dropout(gy.contiguous(), rate, seed, offset), None, None, None


class FlowBalanceFn(torch.autograd.Function):
    """q_in, q_out (S,N) from signed link flows (S,E) (`emulator.py:717-724`); edges (E,2) int64 = [from, to] per link."""

    @staticmethod
    def forward(ctx, flow, handle, sign, scale_in, scale_out, edges):
        flow = flow.contiguous()
        ctx.save_for_backward(flow, scale_in, scale_out, edges)
        return _lib.flow_balance(handle, sign, flow, scale_in, scale_out)

    @staticmethod
    def backward(ctx, g_in, g_out):
        flow, scale_in, scale_out, edges = ctx.saved_tensors
        gi, go = g_in * scale_in, g_out * scale_out                     # (S,N)
        u, v = edges[:, 0], edges[:, 1]
        # from-node u: q_out += max(f,0), q_in += max(-f,0); to-node v: q_in += max(f,0), q_out += max(-f,0)
        pos = go[:, u] + gi[:, v]
        neg = gi[:, u] + go[:, v]
        dflow = torch.where(flow > 0, pos, torch.zeros_like(pos)) - torch.where(flow < 0, neg, torch.zeros_like(neg))
        return dflow, None, None, None, None, None


def grad_on(*tensors):
    """True when autograd is recording and one of the tensors needs a gradient: the modules then take the Function path."""
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)

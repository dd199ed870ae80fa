"""CPU oracle of the training step (TEST INFRASTRUCTURE ONLY; PARITY UNPINNED, see oracle/__init__.py).

Restates `Emulator.fit_eval` and its losses (`surrogate/emulator.py:440-484`) and the constructor's loss weights
(`:82,99-106,116`) on torch CPU tensors; gradients come from torch autograd over `oracle.emulator_ref` (the reference
uses `tf.GradientTape` over the same graph).  Third-party arithmetic restated from its published definition, absent
from /root/reference: keras==2.10.0 (`requirements.txt:2`) `MeanSquaredError`, `BinaryCrossentropy` (reduction
SUM_OVER_BATCH_SIZE, probabilities clipped to [1e-7, 1-1e-7]) and `optimizers.Adam(clipnorm=1.0)` (optimizer_v2:
per-variable `tf.clip_by_norm`, `lr_t = lr sqrt(1-b2^t)/(1-b1^t)`, `var -= lr_t m / (sqrt(v) + 1e-7)`).
"""
import numpy as np
import torch

from . import emulator_ref as ER


def loss_weights(args, dtype=torch.float64):
    """nwei (N, 3+balance), ewei (E), poswei (N)  -- emulator.py:82,99-106,116."""
    c = ER.config(args)
    g = lambda k, d: np.asarray(getattr(args, k, d), dtype=np.float64)
    balance = bool(getattr(args, 'balance', False))
    nwei = np.repeat(g('nwei', np.ones(c.n_node))[:, None], 3 + int(balance), axis=-1).astype(np.float32).astype(np.float64)
    if c.hmin.max() > 0:
        wei = (c.hmax - c.hmin) * (1 - c.is_outfall) + (c.hmax - c.hmin).mean() * c.is_outfall
        wei = (c.hmax.max() - c.hmin.min()) / wei
        nwei = nwei * np.stack([wei] + (2 + int(balance)) * [np.ones_like(wei)], axis=-1)
    T = lambda a: torch.as_tensor(np.asarray(a), dtype=dtype)
    return dict(nwei=T(nwei), ewei=T(g('ewei', np.ones(c.n_edge))), poswei=T(g('poswei', np.ones(c.n_node)).astype(np.float32)))


def mse(y_true, y_pred, sample_weight=None):
    per = ((y_pred - y_true) ** 2).mean(dim=-1)
    if sample_weight is not None:
        per = per * sample_weight
    return per.mean()


def bce(y_true, y_pred, sample_weight):
    p = y_pred.clamp(1e-7, 1 - 1e-7)
    per = -(y_true * torch.log(p) + (1 - y_true) * torch.log(1 - p)).mean(dim=-1)
    return (per * sample_weight).mean()


def model(args, params, norms, x, a, b, ex):
    """`_model` (emulator.py:400-438) on normalised tensors: roll > 0 -> autoregressive chunks, else one forward."""
    c = ER.config(args)
    if c.roll:
        return ER.model_rollout(args, params, norms, x, a, b, ex)
    ae = ER.get_edge_action(c, a) if c.act else None
    y, ey = ER.forward(args, params, x, b, ex, ae)
    y, ey = ER.post_proc(args, norms, y, ey, a, b)
    return y.clamp(0, 1), ey


def losses(args, params, norms, x, a, b, y, ex, ey):
    """[node_loss, (flood_loss,) edge_loss] of fit_eval (emulator.py:459-468)."""
    c = ER.config(args)
    lw = loss_weights(args, x.dtype)
    balance = bool(getattr(args, 'balance', False))
    preds, edge_preds = model(args, params, norms, x, a, b, ex)
    if balance:                                                                  # :441-447
        q_w, pr = ER.constrain(args, ER.normalize(norms, preds, 'y', True), ER.normalize(norms, b, 'b', True)[..., :1])
        q_w = (q_w / norms['y'][0, :, -1]).unsqueeze(-1)
        pr = ER.normalize(norms, pr, 'y').clamp(0, 1)
        node = mse(torch.cat([y[..., :3], y[..., -1:]], dim=-1) * lw['nwei'], torch.cat([pr[..., :3], q_w], dim=-1) * lw['nwei'])
    else:                                                                        # :449
        node = mse(y[..., :3] * lw['nwei'], preds[..., :3] * lw['nwei'])
    out = [node]
    if c.if_flood and not balance:                                               # :452-455,466-467
        weight = lw['poswei'] * y[..., -2] + lw['nwei'][:, -1] * (1 - y[..., -2])
        out.append(bce(y[..., -2:-1], preds[..., -1:], weight))
    out.append(mse(ey, edge_preds, lw['ewei']))                                  # :468
    return out


def tree_leaves(p, prefix=''):
    """(name, tensor) of every parameter in a nested dict / list parameter tree, in a fixed order."""
    if isinstance(p, torch.Tensor):
        yield prefix, p
    elif isinstance(p, dict):
        for k in p:
            yield from tree_leaves(p[k], prefix + '.' + k if prefix else k)
    elif isinstance(p, (list, tuple)):
        for i, v in enumerate(p):
            yield from tree_leaves(v, '%s.%d' % (prefix, i))


def grads(args, params, norms, x, a, b, y, ex, ey):
    """Losses and d(sum of losses)/d(parameter) for every leaf of `params` (name -> tensor)."""
    leaves = list(tree_leaves(params))
    for _, t in leaves:
        t.requires_grad_(True)
        t.grad = None
    ls = losses(args, params, norms, x, a, b, y, ex, ey)
    sum(ls).backward()
    out = {n: (t.grad.clone() if t.grad is not None else torch.zeros_like(t)) for n, t in leaves}
    for _, t in leaves:
        t.requires_grad_(False)
        t.grad = None
    return [l.detach() for l in ls], out


class Adam:
    """keras.optimizers.Adam(learning_rate, clipnorm=1.0) of TF 2.10 on a name -> tensor dict."""

    def __init__(self, lr=1e-3, b1=0.9, b2=0.999, eps=1e-7, clipnorm=1.0):
        self.lr, self.b1, self.b2, self.eps, self.clipnorm, self.t = lr, b1, b2, eps, clipnorm, 0
        self.m, self.v = {}, {}

    def step(self, leaves, grads_):
        self.t += 1
        lr_t = self.lr * (1 - self.b2 ** self.t) ** 0.5 / (1 - self.b1 ** self.t)
        with torch.no_grad():
            for n, p in leaves:
                g = grads_[n]
                if self.clipnorm is not None:
                    g = g * (self.clipnorm / torch.clamp(g.norm(), min=self.clipnorm))
                m = self.m.setdefault(n, torch.zeros_like(p))
                v = self.v.setdefault(n, torch.zeros_like(p))
                m.mul_(self.b1).add_(g, alpha=1 - self.b1)
                v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
                p.sub_(lr_t * m / (v.sqrt() + self.eps))


def grad_norm_step(args, params, norms, x, a, b, y, ex, ey, ini_loss, alpha, opt):
    """`fit_grad_norm` + `_get_grad_norm` (emulator.py:486-519) on the oracle: alpha = tensor([alpha_reg, alpha_cls]) (updated
    in place: one Adam(1e-4) step of `opt`, then rescaled to sum 2); returns the alpha loss.  W = dense_resx kernel = params['res_x']."""
    c = ER.config(args)
    lw = loss_weights(args, x.dtype)
    W = params['res_x']['kernel']
    W.requires_grad_(True)
    preds, edge_preds = model(args, params, norms, x, a, b, ex)
    reg = mse(y[..., :3] * lw['nwei'], preds[..., :3] * lw['nwei']) + mse(ey, edge_preds, lw['ewei'])
    weight = lw['poswei'] * y[..., -2] + lw['nwei'][:, -1] * (1 - y[..., -2])
    fl = bce(y[..., -2:-1], preds[..., -1:], weight)
    g_reg, = torch.autograd.grad(reg, W, retain_graph=True)
    g_cls, = torch.autograd.grad(fl, W)
    W.requires_grad_(False)
    alpha.requires_grad_(True)
    nrm = torch.stack([(alpha[0] * g_reg).norm(), (alpha[1] * g_cls).norm()])
    r = torch.stack([reg.detach() / (ini_loss[0] + ini_loss[-1]), fl.detach() / ini_loss[1]])
    target = nrm.detach().mean() * (r / r.mean()) ** 0.5
    loss = (target - nrm).abs().mean()
    g, = torch.autograd.grad(loss, alpha)
    alpha.requires_grad_(False)
    opt.step([('alpha', alpha)], {'alpha': g})
    with torch.no_grad():
        alpha.mul_(2.0 / alpha.sum())
    return loss.detach()

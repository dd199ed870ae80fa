"""Model-predictive-control evaluation on the engine: what the reference's `mpc_problem_gr`
runs per candidate control sequence (`surrogate/mpc.py:551-612`) -- expand the decision vector to per-step settings,
predict the horizon with the emulator (chunk by chunk when it is longer than `seq_out`), reduce the prediction to the
scalar objective of the scenario (`envs/scenario/astlingen.py:75-99`), and differentiate it with respect to the decision
vector (the reverse mode runs through the HIP backward kernels).  A whole population is ONE batched forward.

The GA / optimiser drivers, SWMM coupling and scenario configuration stay in the reference; the performance targets are
passed as index / weight tensors.
"""
import torch


def expand_settings(y, n_step, n_act, r_step, horizon_steps):
    """Decision vectors (pop, n_step*n_act) -> per-step settings (pop, horizon_steps, n_act): every control step is held for
    r_step simulation steps and the last one to the end of the evaluation horizon (`mpc.py:552-558`)."""
    s = y.reshape(-1, n_step, n_act).repeat_interleave(r_step, dim=1)
    if s.shape[1] < horizon_steps:
        s = torch.cat([s, s[:, -1:].expand(-1, horizon_steps - s.shape[1], -1)], dim=1)
    return s[:, :horizon_steps]


def predict_horizon(emul, settings, state, runoff, edge_state):
    """`mpc_problem_gr.predict` (`mpc.py:565-582`): `predict_tf` on the whole horizon when it is one chunk, else chunk by
    chunk of seq_out steps, each fed with the previous prediction (flood bit = flooding volume > 0, `:572-574`)."""
    so = emul.seq_out
    n_chunk = settings.shape[1] // so
    if n_chunk <= 1:
        return emul.predict_tf(state, runoff, settings, edge_state)
    state, edge_state = state[:, -emul.seq_in:], edge_state[:, -emul.seq_in:]
    ys, es, perf = [], [], None
    for i in range(n_chunk):
        sl = slice(i * so, (i + 1) * so)
        ri, sett = runoff[:, sl], settings[:, sl]
        if emul.if_flood and i > 0:
            state = torch.cat([state[..., :-1], (perf > 0).float(), state[..., -1:]], dim=-1)
        y, ey = emul.predict_tf(state, ri, sett, edge_state)
        state, perf = torch.cat([y[..., :-2], ri], dim=-1), y[..., -1:]
        edge_state = torch.cat([ey, emul.get_edge_action(sett, True)], dim=-1)
        ys.append(y)
        es.append(ey)
    return torch.cat(ys, dim=1), torch.cat(es, dim=1)


def objective_pred(preds, state, flood_idx, flood_w, outflow_idx=None, outflow_w=None, smooth_idx=None, smooth_w=None, gamma=None):
    """`objective_pred_tf` (`envs/scenario/astlingen.py:75-99`) with the scenario's performance targets as tensors:
    flooding volume q_w at `flood_idx` ('cumflooding' targets), inflow at `outflow_idx` ('cuminflow' at the treatment plant),
    absolute inflow change between steps at `smooth_idx` (the `rough` term), each weighted, discounted by gamma (T,), summed
    over time -> (pop,)."""
    q_w = preds[..., -1]                                                   # (pop, T, N)
    q_in = torch.cat([state[:, -1:, :, 1], preds[..., 1]], dim=1)          # (pop, T+1, N)
    obj = (q_w[..., flood_idx] * flood_w).sum(-1)
    if outflow_idx is not None and len(outflow_idx):
        obj = obj + (q_in[:, 1:, outflow_idx] * outflow_w).sum(-1)
    if smooth_idx is not None and len(smooth_idx):
        obj = obj + ((q_in[:, 1:, smooth_idx] - q_in[:, :-1, smooth_idx]).abs() * smooth_w).sum(-1)
    if gamma is not None:
        obj = obj * gamma
    return obj.sum(-1)


def objective_and_gradient(emul, y, state, runoff, edge_state, n_step, n_act, r_step, targets, gamma=None):
    """`gradient_fn` (`mpc.py:600-610`): objective of every candidate in y (pop, n_step*n_act) and its gradient with respect
    to y.  state (T_in, N, C), runoff (T_h, N, b), edge_state (T_in, E, C) are shared by the population (`pre_state`)."""
    y = y.detach().clone().requires_grad_(True)
    pop = y.shape[0]
    settings = expand_settings(y, n_step, n_act, r_step, runoff.shape[0])
    rep = lambda t: t.unsqueeze(0).expand((pop,) + tuple(t.shape)).contiguous()
    st = rep(state)
    preds = predict_horizon(emul, settings, st, rep(runoff), rep(edge_state))
    obj = objective_pred(preds[0], st, gamma=gamma, **targets)
    (grad,) = torch.autograd.grad(obj.sum(), y)
    return obj.detach(), grad


def hessp(emul, y, p, state, runoff, edge_state, n_step, n_act, r_step, targets, gamma=None, eps=None):
    """`hessp_fn` (`mpc.py:616-624`, the Hessian-vector product `trust-constr` asks for in `run_ntopt`): H(y) p for every
    candidate, y and p (pop, n_step*n_act).  The reference nests two GradientTapes; the HIP backward operators are first-order
    (`torch.autograd.Function`s without a double backward), so the product is the central difference of the GRADIENT along p,
        H p ~ (grad f(y + eps p) - grad f(y - eps p)) / (2 eps),    eps = 1e-2 / max|p|  (settings live in [0, 1]),
    two gradient evaluations = two batched forward + backward passes through the same kernels.  The objective is piecewise
    smooth (relu, |.| of the roughness term, hard gates): like the exact second derivative, the difference is meaningful
    away from the kinks only.  Returns (pop, n_step*n_act)."""
    p = p.to(y.dtype)
    if eps is None:
        eps = 1e-2 / max(float(p.abs().max()), 1e-12)
    _, g_hi = objective_and_gradient(emul, y + eps * p, state, runoff, edge_state, n_step, n_act, r_step, targets, gamma)
    _, g_lo = objective_and_gradient(emul, y - eps * p, state, runoff, edge_state, n_step, n_act, r_step, targets, gamma)
    return (g_hi - g_lo) / (2.0 * eps)

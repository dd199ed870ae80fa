"""The model-based-RL virtual rollout on the engine: what `surrogate/mbrl.py:304-347` (`rollout`, a `tf.function`) does per
training episode -- alternate the policy and the emulator over the control horizon, every control step fed with the previous
prediction -- as one device-resident loop (no host round trip between the policy and `predict_tf`; with a policy that is
itself capture-safe the loop body can be replayed from a HIP graph like `Emulator.rollout_graphed`).

    xs, exs, settings, perfs = rollout(emul, policy, x, a, b, y, ex, n_step, r_step, node_attrs, link_attrs)

`policy(obs)`: any callable on the observation list the reference's graph agents see (`[x_obs, (b_obs,) e_obs]`, normalised,
one snapshot per sample: channel-wise the time SUM for cumulative / volume attributes and the last step otherwise,
`mbrl.py:317-323`) returning settings (batch, n_act) -- e.g. `ConvNet` (agent.py) plus an actor head.  The agent classes,
replay buffers and the RL update stay in the reference.
"""
import torch


def observe(dat, attrs):
    """(B, T, R, C) -> (B, R, C): sum over the window for attributes named '*cum*' / '*_vol*', last step for the others
    (`mbrl.py:318-320`)."""
    cols = [dat[..., i].sum(dim=1) if ('cum' in attr or '_vol' in attr) else dat[:, -1, :, i] for i, attr in enumerate(attrs)]
    return torch.stack(cols, dim=-1)


def rollout(emul, policy, x, a, b, y, ex, n_step, r_step, node_attrs, link_attrs, use_pred=False):
    """`rollout` of `mbrl.py:304-347` for the graph agents (`ctrl.conv`).  x (B, T_in, N, C), a (B, >= T_in, n_act) the settings
    that led to x, b (B, n_step * r_step, N, b_in) the boundary (rainfall) of the horizon, y (B, T_in, N, .) whose last channel is
    the performance (flooding) so far, ex (B, T_in, E, C).  Returns the concatenated trajectories
    [states (B, T_in + n_step r_step, N, C), link states, settings, performance]."""
    xs, exs, settings, perfs = [x], [ex], [a[:, :emul.seq_in]], [y[:, :emul.seq_in, :, -1:]]
    for i in range(n_step):
        bi = b[:, i * r_step:(i + 1) * r_step]
        x_obs = observe(emul.normalize(x, 'x'), node_attrs)
        e_obs = observe(emul.normalize(ex, 'e'), link_attrs)
        obs = [x_obs, e_obs]
        if use_pred:                                                  # :321-323: rainfall summed over the window, the rest at its last step
            bn = emul.normalize(bi, 'b')
            obs = [x_obs, torch.stack([bn[..., j].sum(dim=1) if j == 0 else bn[:, -1, :, j] for j in range(bn.shape[-1])], dim=-1), e_obs]
        setting = policy(obs)                                         # (B, n_act)
        setting = setting.unsqueeze(1).expand(-1, r_step, -1).contiguous()      # tf.repeat(setting[:, None], r_step, axis=1)
        settings.append(setting)
        preds, edge_preds = emul.predict_tf(x, bi, setting, ex)
        if emul.if_flood:
            x = torch.cat([preds[..., :-2], (preds[..., -2:-1] > 0.5).to(preds.dtype), bi], dim=-1)
        else:
            x = torch.cat([preds[..., :-1], bi], dim=-1)
        ex = torch.cat([edge_preds, emul.get_edge_action(setting, True)], dim=-1)
        xs.append(x)
        exs.append(ex)
        perfs.append(preds[..., -1:])
    return [torch.cat(t, dim=1) for t in (xs, exs, settings, perfs)]

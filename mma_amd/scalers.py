"""Degree scalers of the node-classification path (reference node_classification/scalers.py:10-64, PNA).

Same function names and signatures as the reference.  As invoked from MMA.forward (layers.py:856) the
`add_all` argument is really the SPARSE adjacency tensor, whose rows all have len() == N, so every
"degree" equals N and amplification = attenuation = 1.0 to an ulp.  `scaler_factors`
evaluates exactly those fp32 expressions; MMA.forward folds them into the dense tail."""
import torch


def avg_d_log(all_degrees):
    return torch.mean(torch.log(all_degrees + 1))                      # scalers.py:10-14


def avg_d_exp(all_degrees):
    return torch.mean(torch.exp(torch.div(1, all_degrees)) - 1)        # scalers.py:17-19


def _degrees(add_all, device):
    return torch.tensor([len(node_nei) for node_nei in add_all], device=device)


def _tile(scale, num_aggregators):
    # scalers.py:33-40 only handles up to 4 aggregators; beyond that the reference fails to broadcast.
    return torch.cat((scale,) * num_aggregators, 0) if num_aggregators > 1 else scale


def scale_identity(input, add_all, num_aggregators, avg_d=None):
    return input


def scale_amplification(input, add_all, num_aggregators, avg_d=None):
    all_degrees = _degrees(add_all, input.device)
    scale = (torch.log(all_degrees + 1) / avg_d_log(all_degrees)).unsqueeze(-1)
    return torch.mul(_tile(scale, num_aggregators), input)


def scale_attenuation(input, add_all, num_aggregators, avg_d=None):
    all_degrees = _degrees(add_all, input.device)
    scale = (avg_d_log(all_degrees) / torch.log(all_degrees + 1)).unsqueeze(-1)
    return torch.mul(_tile(scale, num_aggregators), input)


SCALERS = {'identity': scale_identity, 'amplification': scale_amplification, 'attenuation': scale_attenuation}


def scaler_factors(N, device):
    """(amplification, attenuation) row factors, each (N,1), as MMA.forward evaluates them: all degrees == N."""
    all_degrees = torch.full((N,), N, dtype=torch.int64, device=device)
    lg = torch.log(all_degrees + 1)
    avg = torch.mean(lg)
    return (lg / avg).unsqueeze(-1), (avg / lg).unsqueeze(-1)


_ROW_FACTOR = {}


def scaler_row_factor(N, device):
    """1 + amplification + attenuation as ONE (1,1) tensor.  Every row of scaler_factors(N) holds the same value (all
    "degrees" are N, quirk Q1), so it is evaluated once per (N, device) with exactly the operations above - the mean over N
    equal logs included, whose rounding depends on N - and multiplied in by broadcast: bit-identical to the per-row form,
    without six N-element kernels per forward."""
    key = (int(N), str(device))
    c = _ROW_FACTOR.get(key)
    if c is None:
        amp, att = scaler_factors(N, device)
        c = _ROW_FACTOR[key] = (1.0 + amp[:1] + att[:1]).detach()
    return c


# ---- strict_reference=False: degree scalers evaluated with the TRUE degrees (extension, SURVEY 7 "Quirk fidelity vs. sanity")
TRUE_DEGREE_SCALERS = ("identity", "amplification", "attenuation", "linear", "inverse_linear")   # mma_conv.py:181-192


def true_degree_row_factor(deg, scalers, compound, avg_d=None):
    """The per-row factor R(i) that the scaler stage reduces to when the layer weight is stacked once per scaler:
        cat_s(c_s(i) * m[i]) @ [W; ...; W]  ==  (sum_s c_s(i)) * (m[i] @ W),
    with c_s from the node's true degree d_i = len(add_all[i]) clamped to >= 1 (mma_conv.py:179):
      compound=False  the node-classification scalers (scalers.py:22-62) as PNA meant them:  c_s = f_s(d_i);
      compound=True   the graph-regression form (mma_conv.py:181-196, quirk G7):             c_s = prod_{q<=s} f_q(d_i),
    f = 1 | log(d+1)/avg_log | avg_log/log(d+1) | d/avg_lin | avg_lin/d.   avg_d: {'log','lin'} (default: the means over
    `deg`, i.e. PNA's delta; the sharded layer passes the GLOBAL means).  deg: (N,) tensor.  Returns (N,1) fp32."""
    d = deg.to(torch.float32).clamp(min=1)
    lg = torch.log(d + 1)
    avg_log = lg.mean() if avg_d is None else torch.as_tensor(avg_d["log"], dtype=torch.float32, device=d.device)
    avg_lin = d.mean() if avg_d is None else torch.as_tensor(avg_d["lin"], dtype=torch.float32, device=d.device)
    f = {"identity": torch.ones_like(d), "amplification": lg / avg_log, "attenuation": avg_log / lg,
         "linear": d / avg_lin, "inverse_linear": avg_lin / d}
    total, run = torch.zeros_like(d), torch.ones_like(d)
    for name in scalers:
        if name not in f:
            raise ValueError('Unknown scaler "%s".' % name)          # mma_conv.py:193-194
        run = run * f[name] if compound else f[name]
        total = total + run
    return total.unsqueeze(-1)

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

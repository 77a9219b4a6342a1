"""Oracle: invariant-point-attention block and encoder (torch CPU fp32).  Test infrastructure only.

Reference: /root/reference/AbDock/src/modules/encoders/ga.py (byte-identical to AbDesign's):
  _alpha_from_logits :11-26     _node_logits :81-86      _pair_logits :88-90
  _spatial_logits :92-112       _pair_aggregation :114-118   _node_aggregation :120-125
  _spatial_aggregation :127-147 GABlock.forward :149-178  GAEncoder.forward :190-193
  LayerNorm: /root/reference/AbDock/src/modules/common/layers.py:146-155

`sd` is a state_dict-like mapping; `pre` the key prefix of one block, e.g.
'diffusion.eps_net.encoder.blocks.0.'.  mode='ref' keeps the reference's op order
(broadcast product, then sum); mode='mm' contracts with matmul (same maths, faster, used for
big test sizes and reported separately by bench.py).
"""
import math
import torch
import torch.nn.functional as F
from .geometry import to_global, to_local, unit

H, D, P = 12, 32, 8     # heads, qk/value channels, points per head (ga.py:42-43)


def layer_norm(x, g, b, eps=1e-10):
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / (var + eps).sqrt() * g + b


def attention_logits(sd, pre, R, t, x, z, mode='ref'):
    N, L, _ = x.shape
    q = F.linear(x, sd[pre + 'proj_query.weight']).view(N, L, H, D)
    k = F.linear(x, sd[pre + 'proj_key.weight']).view(N, L, H, D)
    qp = to_global(R, t, F.linear(x, sd[pre + 'proj_query_point.weight']).view(N, L, H * P, 3)).reshape(N, L, H, P * 3)
    kp = to_global(R, t, F.linear(x, sd[pre + 'proj_key_point.weight']).view(N, L, H * P, 3)).reshape(N, L, H, P * 3)
    if mode == 'ref':
        l_node = (q.unsqueeze(2) * k.unsqueeze(1) * (1 / math.sqrt(D))).sum(-1)
        d2 = ((qp.unsqueeze(2) - kp.unsqueeze(1)) ** 2).sum(-1)
    else:
        l_node = torch.einsum('nihd,njhd->nijh', q, k) * (1 / math.sqrt(D))
        d2 = torch.stack([((qp[:, :, None, h] - kp[:, None, :, h]) ** 2).sum(-1) for h in range(H)], dim=-1)
    l_pair = F.linear(z, sd[pre + 'proj_pair_bias.weight'])
    gamma = F.softplus(sd[pre + 'spatial_coef'])
    l_spat = d2 * ((-1 * gamma * math.sqrt(2 / (9 * P))) / 2)
    return l_node, l_pair, l_spat


def attention_weights(logits, mask, inf=1e5):
    """ga.py:11-26: additive -1e5 on masked pairs, softmax over j, zero masked query rows."""
    N, L = mask.shape
    mrow = mask.view(N, L, 1, 1).expand_as(logits)
    mpair = mrow * mrow.permute(0, 2, 1, 3)
    logits = torch.where(mpair, logits, logits - inf)
    alpha = torch.softmax(logits, dim=2)
    return torch.where(mrow, alpha, torch.zeros_like(alpha))


def aggregate(sd, pre, alpha, R, t, x, z, mode='ref'):
    N, L, _ = x.shape
    v = F.linear(x, sd[pre + 'proj_value.weight']).view(N, L, H, D)
    vp = to_global(R, t, F.linear(x, sd[pre + 'proj_value_point.weight']).view(N, L, H, P, 3))
    if mode == 'ref':
        f_pair = (alpha.unsqueeze(-1) * z.unsqueeze(-2)).sum(dim=2)                 # (N,L,H,C)
        f_node = (alpha.unsqueeze(-1) * v.unsqueeze(1)).sum(dim=2)                   # (N,L,H,D)
        agg = (alpha.reshape(N, L, L, H, 1, 1) * vp.unsqueeze(1)).sum(dim=2)         # (N,L,H,P,3)
    else:
        f_pair = torch.einsum('nijh,nijc->nihc', alpha, z)
        f_node = torch.einsum('nijh,njhd->nihd', alpha, v)
        agg = torch.einsum('nijh,njhpa->nihpa', alpha, vp)
    loc = to_local(R, t, agg)
    dist = loc.norm(dim=-1)
    direc = unit(loc, eps=1e-4)
    feat = torch.cat([f_pair.reshape(N, L, -1), f_node.reshape(N, L, -1),
                      loc.reshape(N, L, -1), dist.reshape(N, L, -1), direc.reshape(N, L, -1)], dim=-1)
    return feat


def ga_block(sd, pre, R, t, x, z, mask, mode='ref', parts=None):
    """One GABlock.forward (ga.py:149-178).  `parts`, if a dict, receives intermediates."""
    l_node, l_pair, l_spat = attention_logits(sd, pre, R, t, x, z, mode)
    alpha = attention_weights((l_node + l_pair + l_spat) * math.sqrt(1 / 3), mask)
    feat = aggregate(sd, pre, alpha, R, t, x, z, mode)
    u = F.linear(feat, sd[pre + 'out_transform.weight'], sd[pre + 'out_transform.bias'])
    u = torch.where(mask.unsqueeze(-1), u, torch.zeros_like(u))
    y = layer_norm(x + u, sd[pre + 'layer_norm_1.gamma'], sd[pre + 'layer_norm_1.beta'])
    m = F.linear(y, sd[pre + 'mlp_transition.0.weight'], sd[pre + 'mlp_transition.0.bias']).relu()
    m = F.linear(m, sd[pre + 'mlp_transition.2.weight'], sd[pre + 'mlp_transition.2.bias']).relu()
    m = F.linear(m, sd[pre + 'mlp_transition.4.weight'], sd[pre + 'mlp_transition.4.bias'])
    out = layer_norm(y + m, sd[pre + 'layer_norm_2.gamma'], sd[pre + 'layer_norm_2.beta'])
    if parts is not None:
        parts.update(l_node=l_node, l_pair=l_pair, l_spat=l_spat, alpha=alpha, feat=feat)
    return out


def ga_encoder(sd, pre, R, t, x, z, mask, num_layers, mode='ref'):
    """GAEncoder.forward (ga.py:190-193): same R, t, z for every block."""
    for i in range(num_layers):
        x = ga_block(sd, f'{pre}blocks.{i}.', R, t, x, z, mask, mode)
    return x

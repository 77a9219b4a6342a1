"""Oracle: SO(3) maps and rigid-frame algebra (torch CPU fp32).  Test infrastructure only.

Reference (D/ = /root/reference/AbDock/src/):
  exp map        D/modules/common/so3.py:33-57
  log map        D/modules/common/so3.py:10-30,60-63
  quaternions    D/modules/common/geometry.py:148-175 (general), :215-233 ((1,b,c,d) form)
  frames         D/modules/common/geometry.py:32-33,47-69,72-117
"""
import math
import torch


def hat(w):
    """so(3) vector -> the reference's 3x3 'skew' layout (so3.py:33-41).

    Note the layout is the transpose of the textbook hat map: row0 = (0, z, -y).
    """
    x, y, z = w[..., 0], w[..., 1], w[..., 2]
    o = torch.zeros_like(x)
    rows = [o, z, -y, -z, o, x, y, -x, o]
    return torch.stack(rows, dim=-1).reshape(w.shape[:-1] + (3, 3))


def vee(S):
    """Inverse of hat(): reads (S12, S20, S01)  (so3.py:25-30)."""
    return torch.stack([S[..., 1, 2], S[..., 2, 0], S[..., 0, 1]], dim=-1)


def so3_exp(w):
    """Rodrigues with the reference's epsilons (so3.py:44-57)."""
    S = hat(w)
    th = torch.linalg.norm(w, dim=-1)
    b = (torch.sin(th) + 1e-8) / (th + 1e-8)
    c = (1 - torch.cos(th) + 1e-8) / (th ** 2 + 2e-8)
    eye = torch.eye(3, dtype=w.dtype).expand(S.shape)
    return eye + b[..., None, None] * S + c[..., None, None] * (S @ S)


def so3_log(R, grad_mode=False):
    """Rotation matrix -> so(3) vector (so3.py:10-22,60-63).

    grad_mode mirrors `torch.is_grad_enabled()` in the reference: the cosine is clamped
    at -0.999 when autograd is on and at -1.0 under no_grad.
    """
    tr = R[..., 0, 0] + R[..., 1, 1] + R[..., 2, 2]
    cmin = -0.999 if grad_mode else -1.0
    cos_t = ((tr - 1) / 2).clamp_min(cmin)
    sin_t = torch.sqrt(1 - cos_t ** 2)
    th = torch.acos(cos_t)
    coef = (th + 1e-8) / (2 * sin_t + 2e-8)
    return vee(coef[..., None, None] * (R - R.transpose(-1, -2)))


def quat_to_rot(q):
    """General quaternion (real first) -> R, normalising first (geometry.py:148-175)."""
    q = torch.nn.functional.normalize(q, dim=-1)
    r, i, j, k = q.unbind(-1)
    two_s = 2.0 / (q * q).sum(-1)
    m = [1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
         two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
         two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)]
    return torch.stack(m, -1).reshape(q.shape[:-1] + (3, 3))


def quat1ijk_to_rot(e):
    """(1 + b i + c j + d k) -> R (geometry.py:215-233)."""
    b, c, d = e.unbind(-1)
    s = torch.sqrt(1 + b ** 2 + c ** 2 + d ** 2)
    a, b, c, d = 1 / s, b / s, c / s, d / s
    m = [a ** 2 + b ** 2 - c ** 2 - d ** 2, 2 * b * c - 2 * a * d, 2 * b * d + 2 * a * c,
         2 * b * c + 2 * a * d, a ** 2 - b ** 2 + c ** 2 - d ** 2, 2 * c * d - 2 * a * b,
         2 * b * d - 2 * a * c, 2 * c * d + 2 * a * b, a ** 2 - b ** 2 - c ** 2 + d ** 2]
    return torch.stack(m, -1).reshape(e.shape[:-1] + (3, 3))


def unit(v, eps=1e-6):
    """v / (|v| + eps)  (geometry.py:32-33)."""
    return v / (torch.linalg.norm(v, ord=2, dim=-1, keepdim=True) + eps)


def frames_from_backbone(ca, c, n):
    """Gram-Schmidt frame with columns [e1 e2 e3] (geometry.py:47-69)."""
    e1 = unit(c - ca)
    v2 = n - ca
    e2 = unit(v2 - (e1 * v2).sum(-1, keepdim=True) * e1)
    e3 = torch.cross(e1, e2, dim=-1)
    return torch.stack([e1, e2, e3], dim=-1)


def to_global(R, t, p):
    """q = R p + t for points p (N,L,...,3) (geometry.py:72-91)."""
    shp = p.shape
    N, L = shp[0], shp[1]
    pp = p.reshape(N, L, -1, 3).transpose(-1, -2)
    q = torch.matmul(R, pp) + t.unsqueeze(-1)
    return q.transpose(-1, -2).reshape(shp)


def to_local(R, t, q):
    """p = R^T (q - t) (geometry.py:94-113)."""
    shp = q.shape
    N, L = shp[0], shp[1]
    qq = q.reshape(N, L, -1, 3).transpose(-1, -2)
    p = torch.matmul(R.transpose(-1, -2), qq - t.unsqueeze(-1))
    return p.transpose(-1, -2).reshape(shp)


def rotate(R, p):
    """R p (geometry.py:116-117)."""
    return to_global(R, torch.zeros_like(p), p)


def dihedral(p0, p1, p2, p3):
    """Signed dihedral of four points (geometry.py:255-273)."""
    v0, v1, v2 = p2 - p1, p0 - p1, p3 - p2
    u1 = torch.cross(v0, v1, dim=-1)
    n1 = u1 / torch.linalg.norm(u1, dim=-1, keepdim=True)
    u2 = torch.cross(v0, v2, dim=-1)
    n2 = u2 / torch.linalg.norm(u2, dim=-1, keepdim=True)
    sgn = torch.sign((torch.cross(v1, v2, dim=-1) * v0).sum(-1))
    ang = sgn * torch.acos((n1 * n2).sum(-1).clamp(min=-0.999999, max=0.999999))
    return torch.nan_to_num(ang)


PI = math.pi

"""Reference state-dict -> packed device tensors.

The loader accepts the reference's key layout verbatim and performs, once at load time, the
layout work that would otherwise cost HBM passes per call: weight_norm folding (g*v/||v||), conv weights to the
tap-major [N][taps*Cin] image the implicit-GEMM kernel walks, ConvTranspose1d to its phase-packed stride-1 form,
fused QKV / gate-up row blocks, zero-padded input channels where Cin is not a multiple of 16.
"""
import torch


def fold_weight_norm(sd, prefix):
    """w = g * v / ||v|| with the norm over all dims but 0 (torch.nn.utils.parametrizations.weight_norm, dim=0).
    Accepts the parametrization keys, the legacy weight_g / weight_v keys, or a plain weight."""
    if prefix + ".weight" in sd:
        return sd[prefix + ".weight"].float()
    if prefix + ".parametrizations.weight.original0" in sd:
        g, v = sd[prefix + ".parametrizations.weight.original0"], sd[prefix + ".parametrizations.weight.original1"]
    else:
        g, v = sd[prefix + ".weight_g"], sd[prefix + ".weight_v"]
    g, v = g.float(), v.float()
    return g * v / v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))


def pack_conv(w, cin_pad=None):
    """torch Conv1d weight (Cout, Cin, k) -> (Cout, k*Cin_pad) with k-major, channel-minor columns."""
    cout, cin, k = w.shape
    cp = cin_pad or cin
    out = torch.zeros(cout, k, cp, dtype=torch.float32)
    out[:, :, :cin] = w.float().permute(0, 2, 1)
    return out.reshape(cout, k * cp).contiguous()


def pack_conv_transpose(w, bias, stride, padding):
    """ConvTranspose1d weight (Cin, Cout, k) -> stride-1 conv with 3 taps (input offsets -1,0,+1) producing
    stride*Cout columns: column r*Cout+co of output row t is output sample t*stride + r.  Returns (Wp, bias_p)."""
    cin, cout, k = w.shape
    s, p = stride, padding
    wp = torch.zeros(s * cout, 3, cin, dtype=torch.float32)
    for r in range(s):
        for d in (-1, 0, 1):
            j = r + p - d * s
            if 0 <= j < k:
                wp[r * cout:(r + 1) * cout, d + 1, :] = w[:, :, j].float().t()
    n_cov = sum(1 for r in range(s) for d in (-1, 0, 1) if 0 <= r + p - d * s < k)
    assert n_cov == k, f"ConvTranspose1d(k={k}, s={s}, p={p}) needs more than 3 input taps"
    bp = None if bias is None else bias.float().repeat(s)
    return wp.reshape(s * cout, 3 * cin).contiguous(), bp


def pack_swiglu(gate, up):
    """(F, D) gate and up -> (2F, D): blocks of [32 gate rows | 32 up rows] so that one wave's two 32-column MFMA
    tiles hold silu-input and multiplier of the same 32 features."""
    f, d = gate.shape
    g = gate.float().view(f // 32, 32, d)
    u = up.float().view(f // 32, 32, d)
    return torch.cat([g, u], dim=1).reshape(2 * f, d).contiguous()


def pad_cols(w, to):
    out = torch.zeros(w.shape[0], to, dtype=torch.float32)
    out[:, : w.shape[1]] = w
    return out

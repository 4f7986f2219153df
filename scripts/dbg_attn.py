import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chatterbox_amd import ops
dev = torch.device("cuda:0")
def pe(x):
    h = x.half(); l = ((x.double() - h.double()) * 2048.0).half(); return h.double() + l.double() / 2048.0
for ver in (1, 2):
    ops.lib.cbx_set_attn_planes_version(ver)
    for (Z, T) in [(1, 64), (1, 128), (1, 192), (1, 256), (1, 320), (2, 1000)]:
        H, Tp = 8, (T + 7) // 8 * 8
        g = torch.Generator().manual_seed(1)
        q, k, v = [torch.randn(Z, T, H, 64, generator=g) for _ in range(3)]
        qk = ops.Planes(Z * T, 1024, dev)
        ops.split_planes(q.reshape(Z * T, 512).to(dev), qk.cols(0, 512)); ops.split_planes(k.reshape(Z * T, 512).to(dev), qk.cols(512, 512))
        vt = ops.Planes(Z * 512, Tp, dev, zero=True)
        vtt = torch.zeros(Z, 512, Tp); vtt[:, :, :T] = v.reshape(Z, T, 512).transpose(1, 2)
        ops.split_planes(vtt.reshape(Z * 512, Tp).to(dev), vt)
        out = ops.Planes(Z * T, 512, dev)
        ops.flash_attn_planes(qk.cols(0, 512), qk.cols(512, 512), vt, out, Z=Z, H=H, T=T, vt_sb=512 * vt.ld, scale=0.125)
        s = torch.einsum("zqhd,zkhd->zhqk", pe(q), pe(k)) * 0.125
        ref = torch.einsum("zhqk,zkhd->zqhd", torch.softmax(s, -1), pe(v)).reshape(Z * T, 512)
        err = (out.float().cpu().double() - ref).abs()
        # error by query block of 32 and by head
        eq = err.view(Z, T, 8, 64).amax(dim=(0, 2, 3))
        print(f"v{ver} Z={Z} T={T}: max err {err.max():.3e}; per 32-query block: " + " ".join(f"{eq[i:i+32].max():.1e}" for i in range(0, min(T, 512), 32)), flush=True)

"""On-disk formats: checkpoints and voices without torch pickles, and a cache of the packed device layout.

  read_safetensors / write_safetensors   the safetensors container (8-byte little-endian header length, JSON header with dtype /
        shape / data_offsets per tensor, raw little-endian data) parsed directly: tensors are zero-copy views of ONE read-only
        memory map, so a 2 GB checkpoint costs no host copy before the H2D transfer.  Interoperable with the `safetensors` package
        (the reference's loader, tts.py:139-160) in both directions.
  save_conds / load_conds                `Conditionals` (reference tts.py:64-103 stores them as a torch pickle, conds.pt) in a safetensors
        container: T3Cond fields under `t3.*`, the S3Gen reference dict under `gen.*`; `None` fields are simply absent.
  save_packed / load_packed              the tensors an engine holds AFTER load-time packing (lane-ordered GEMV images, fused q/k/v,
        SwiGLU interleave, folded weight norm, Toeplitz convolutions ...) keyed by a fingerprint of the source checkpoint, so that a
        later start maps the packed image straight to the device instead of re-deriving it.
"""
import hashlib
import json
import mmap
import os
import struct

import numpy as np
import torch

_DT = {"F64": torch.float64, "F32": torch.float32, "F16": torch.float16, "BF16": torch.bfloat16, "I64": torch.int64, "I32": torch.int32,
       "I16": torch.int16, "I8": torch.int8, "U8": torch.uint8, "BOOL": torch.bool}
_DT_INV = {v: k for k, v in _DT.items()}


def read_safetensors(path, with_metadata=False):
    """-> {name: tensor} (read-only views of a memory map; `.clone()` or `.to(device)` before writing to them)."""
    f = open(path, "rb")
    n = struct.unpack("<Q", f.read(8))[0]
    header = json.loads(f.read(n).decode("utf-8"))
    meta = header.pop("__metadata__", {}) or {}
    mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
    base = 8 + n
    out = {}
    for name, info in header.items():
        lo, hi = info["data_offsets"]
        dt = _DT[info["dtype"]]
        if hi == lo:
            out[name] = torch.empty(info["shape"], dtype=dt)
            continue
        buf = np.frombuffer(mm, dtype=np.uint8, count=hi - lo, offset=base + lo)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # "the given NumPy array is not writable": the views are documented read-only
            t = torch.from_numpy(buf)
        out[name] = t.view(dt).view(info["shape"])
    return (out, meta) if with_metadata else out


def write_safetensors(tensors, path, metadata=None):
    """{name: tensor} -> file (tensors are written contiguous, CPU, in name order; 8-byte aligned header like the reference writer)."""
    header, off, blobs = {}, 0, []
    if metadata:
        header["__metadata__"] = {str(k): str(v) for k, v in metadata.items()}
    for name in sorted(tensors):
        t = tensors[name].detach().cpu().contiguous()
        nbytes = t.numel() * t.element_size()
        header[name] = {"dtype": _DT_INV[t.dtype], "shape": list(t.shape), "data_offsets": [off, off + nbytes]}
        blobs.append(t)
        off += nbytes
    hj = json.dumps(header, separators=(",", ":")).encode("utf-8")
    hj += b" " * ((8 - len(hj) % 8) % 8)
    tmp = str(path) + ".tmp"
    with open(tmp, "wb") as f:
        f.write(struct.pack("<Q", len(hj)))
        f.write(hj)
        for t in blobs:
            if t.numel():  # reshape(-1) first: a 0-d tensor (num_batches_tracked, a scalar emotion_adv) cannot be viewed as bytes
                t = t.reshape(-1)
                f.write(t.view(torch.uint8).numpy().tobytes() if t.dtype != torch.bool else t.to(torch.uint8).numpy().tobytes())
    os.replace(tmp, path)


# ----------------------------------------------------------------------------- voices


def save_conds(conds, path):
    """Conditionals -> safetensors container (no pickle)."""
    t = {}
    for k, v in conds.t3.__dict__.items():
        if torch.is_tensor(v):
            t["t3." + k] = v
        elif isinstance(v, (int, float)):
            t["t3." + k] = torch.tensor(float(v))
    for k, v in conds.gen.items():
        if torch.is_tensor(v):
            t["gen." + k] = v
    write_safetensors(t, path, metadata={"format": "chatterbox-conds", "version": 1})


def load_conds(path, device="cpu"):
    from .api import Conditionals, T3Cond
    t, meta = read_safetensors(path, with_metadata=True)
    assert meta.get("format") == "chatterbox-conds", f"{path} is not a conds container"
    t3 = {k[3:]: v.clone().to(device) for k, v in t.items() if k.startswith("t3.")}
    gen = {k[4:]: v.clone().to(device) for k, v in t.items() if k.startswith("gen.")}
    gen.setdefault("prompt_feat_len", None)
    return Conditionals(T3Cond(**t3), gen)


# ----------------------------------------------------------------------------- packed device layout cache


def fingerprint(sd):
    """Cheap, order-independent identity of a checkpoint: names, shapes, dtypes and a strided sample of the bytes of every tensor."""
    h = hashlib.sha256()
    for k in sorted(sd):
        v = sd[k]
        h.update(k.encode())
        h.update(str(tuple(v.shape)).encode())
        h.update(str(v.dtype).encode())
        flat = v.detach().reshape(-1)
        n = flat.numel()
        if n:  # the first and last 4 KiB in full + 1024 strided samples: an edit that misses all of them has to be adversarial
            step = max(1, n // 1024)
            for piece in (flat[:1024], flat[max(0, n - 1024):], flat[::step][:1024]):
                h.update(piece.to(torch.float64).cpu().numpy().tobytes())
    return h.hexdigest()[:32]


PACKED_LAYOUT_VERSION = 2  # bump whenever an engine changes what export_packed() holds or how an image is laid out


def packed_kind(base):
    """`kind` of a packed image: the engine flavour + everything the packed layout depends on besides the checkpoint -- the C ABI version
    (lane orders are part of it) and this module's layout version -- so that a library / layout change never maps a stale image."""
    from ._lib import ABI_VERSION
    return f"{base}-abi{ABI_VERSION}-pl{PACKED_LAYOUT_VERSION}"


def save_packed(tensors, path, source_fingerprint, kind):
    """tensors: flat {name: device tensor} as exported by an engine (`export_packed()`)."""
    write_safetensors(tensors, path, metadata={"format": "chatterbox-packed", "version": PACKED_LAYOUT_VERSION, "kind": kind,
                                               "source": source_fingerprint})


def load_packed(path, source_fingerprint, kind):
    """-> {name: tensor view} or None when the file is missing / belongs to another checkpoint or engine kind."""
    if not os.path.exists(path):
        return None
    t, meta = read_safetensors(path, with_metadata=True)
    if (meta.get("format") != "chatterbox-packed" or meta.get("kind") != kind or meta.get("source") != source_fingerprint
            or str(meta.get("version")) != str(PACKED_LAYOUT_VERSION)):
        return None
    return t

"""Route `chatterbox_amd.ops` to the SIMT-emulated library on CPU tensors -- for tests only (see simt_emu.h).

    with emulated():
        ops.linear(x_cpu, w_cpu, out_cpu)      # runs chatterbox_amd/csrc/gemm_f32.hip on the emulator

The product wrappers are not changed: the context manager swaps the ctypes handle (`_lib.lib`, `ops.lib`), the stream getter and the
"is a CUDA fp32 tensor" assertion for the duration, and restores them afterwards.  Nothing under chatterbox_amd/ imports this module.
"""
import contextlib
import ctypes
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

_EMU = None


def load_emu():
    global _EMU
    if _EMU is None:
        import build_emu
        from chatterbox_amd import _lib
        lib = ctypes.CDLL(build_emu.build())
        for name, (args, res) in _lib._SIGS.items():
            fn = getattr(lib, name)  # the emulated library exports the whole ABI: a missing symbol is an error
            fn.argtypes, fn.restype = args, res
        assert lib.cbx_abi_version() == _lib.ABI_VERSION
        _EMU = lib
    return _EMU


CPU = torch.device("cpu")


@contextlib.contextmanager
def emulated():
    from chatterbox_amd import _lib, ops
    emu = load_emu()

    def f32(t, name):
        assert t.dtype == torch.float32 and t.device.type == "cpu", f"{name}: the emulator takes CPU fp32 tensors, got {t.dtype} on {t.device}"
        return t

    def planes_init(self, rows, C, device, zero=False, t=None, c0=0, width=None):
        # ops.Planes.__init__ minus its is_cuda assertion; fresh planes are filled with a NaN pattern instead of torch.empty's leftovers
        self.t = t if t is not None else (torch.zeros(rows, 2 * C, dtype=torch.float16) if zero else torch.full((rows, 2 * C), float("nan"), dtype=torch.float16))
        assert self.t.dtype == torch.float16 and self.t.stride(1) == 1
        self.rows, self.Call, self.c0 = self.t.shape[0], self.t.shape[1] // 2, c0
        self.C = self.Call - c0 if width is None else width

    def enable_range_flag(device=None):
        if 0 not in ops._RANGE_FLAGS:  # [two words, index of the registered one]: the layout of ops.enable_range_flag (round 5)
            ops._RANGE_FLAGS[0] = [torch.zeros(2, dtype=torch.int32), 0]
            _lib.check(emu.cbx_set_range_flag(ops._RANGE_FLAGS[0][0].data_ptr()), "cbx_set_range_flag")
        return ops._RANGE_FLAGS[0][0][:1]

    da_ws = (torch.empty(128 * 8 * 66), torch.zeros(128, dtype=torch.int32))

    def ensure_decode_attn_workspace(device):
        _lib.check(emu.cbx_set_decode_attn_workspace(da_ws[0].data_ptr(), da_ws[1].data_ptr(), 128), "cbx_set_decode_attn_workspace")
        ops._DA_WS[0] = da_ws

    # the engines' own device plumbing: no hipGraph capture, no streams, no device switch on the emulator
    import functools

    from chatterbox_amd import t3 as _t3, t3_turbo as _t3t
    eng_saved = (_t3.T3Engine.generate, _t3t.T3TurboEngine.generate, torch.cuda.current_stream, torch.cuda.set_device)

    def no_graph(fn):
        @functools.wraps(fn)
        def wrapper(self, *a, **k):
            k["use_graph"] = False
            return fn(self, *a, **k)
        return wrapper

    _t3.T3Engine.generate, _t3t.T3TurboEngine.generate = no_graph(eng_saved[0]), no_graph(eng_saved[1])
    torch.cuda.current_stream = lambda *a, **k: type("HostStream", (), {"cuda_stream": None, "synchronize": lambda self: None})()
    torch.cuda.set_device = lambda *a, **k: None
    saved_da = dict(ops._DA_WS)
    saved = dict(erf=ops.enable_range_flag, eda=ops.ensure_decode_attn_workspace, lib_l=_lib.lib, lib_o=ops.lib, stream=ops._stream, f32=ops._f32, pinit=ops.Planes.__init__, sync=torch.cuda.synchronize,
                 flags=dict(ops._RANGE_FLAGS), dev_index=ops._dev_index, cur_dev=torch.cuda.current_device)
    _lib.lib = ops.lib = emu
    ops._stream = lambda: None
    ops._f32 = f32
    ops.Planes.__init__ = planes_init
    ops._dev_index = lambda device=None: 0
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.current_device = lambda: 0
    ops._RANGE_FLAGS.clear()
    ops.enable_range_flag, ops.ensure_decode_attn_workspace = enable_range_flag, ensure_decode_attn_workspace
    try:
        yield emu
    finally:
        emu.cbx_set_range_flag(None)
        emu.cbx_set_decode_attn_workspace(None, None, 0)
        _lib.lib, ops.lib, ops._stream, ops._f32 = saved["lib_l"], saved["lib_o"], saved["stream"], saved["f32"]
        ops.Planes.__init__, torch.cuda.synchronize = saved["pinit"], saved["sync"]
        ops._dev_index, torch.cuda.current_device = saved["dev_index"], saved["cur_dev"]
        ops.enable_range_flag, ops.ensure_decode_attn_workspace = saved["erf"], saved["eda"]
        ops._RANGE_FLAGS.clear()
        ops._RANGE_FLAGS.update(saved["flags"])
        ops._DA_WS.clear()
        ops._DA_WS.update(saved_da)
        _t3.T3Engine.generate, _t3t.T3TurboEngine.generate, torch.cuda.current_stream, torch.cuda.set_device = eng_saved

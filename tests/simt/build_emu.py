"""Build tests/simt/libcbx_emu.so: the kernel sources of chatterbox_amd/csrc compiled for the x86 host against the SIMT emulator.

TEST INFRASTRUCTURE.  The product library is chatterbox_amd/libcbx_hip.so (hipcc, gfx950); nothing under chatterbox_amd/ knows this file.

The sources are used as they are, except for two mechanical rewrites done on a COPY (tests/simt/_gen/):
  * inline asm: `s_waitcnt` / `s_nop` / empty optimisation barriers are dropped; `v_max3_f32 d, |a|, |b|, d` and the
    `v_fma_mix{lo,hi}_f16` pairs become calls of the equivalent C functions of simt_emu.h; `global_store_dword{,x2,x4} … sc1` (the write-through
    stores of cbx_common.h) become plain stores; any OTHER asm statement is an error;
  * `extern __shared__ T name[];` becomes a pointer to the launch's dynamic LDS block.

    python tests/simt/build_emu.py [--force]
"""
import hashlib
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
CSRC = os.path.join(ROOT, "chatterbox_amd", "csrc")
# CBX_EMU_ASAN=1: an AddressSanitizer build (libcbx_emu_asan.so): out-of-range reads / writes of the kernels on host buffers become
# reports.  Run the Python process with LD_PRELOAD=$(clang++ -print-file-name=libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0
ASAN = os.environ.get("CBX_EMU_ASAN") == "1"
LIB = os.path.join(HERE, "libcbx_emu_asan.so" if ASAN else "libcbx_emu.so")
GEN = os.path.join(HERE, "_gen_asan" if ASAN else "_gen")
CLANG = os.environ.get("CBX_EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")
FLAGS = ["-x", "c++", "-std=c++17", "-O2", "-g0", "-fPIC", "-fno-strict-aliasing", "-Wno-everything", "-ffp-contract=on", "-mfma",  # same contraction rule as the gfx950 build (chatterbox_amd/build.py FLAGS)
         "-I", os.path.join(HERE, "shim"), "-I", HERE] + (["-fsanitize=address", "-fsanitize-recover=address", "-fno-omit-frame-pointer", "-g"] if os.environ.get("CBX_EMU_ASAN") == "1" else [])


def _match_paren(s, i):
    """s[i] == '(' -> index one past its matching ')', skipping string literals."""
    depth, j, n = 0, i, len(s)
    while j < n:
        c = s[j]
        if c == '"':
            j += 1
            while s[j] != '"':
                j += 2 if s[j] == "\\" else 1
        elif c == "(":
            depth += 1
        elif c == ")":
            depth -= 1
            if depth == 0:
                return j + 1
        j += 1
    raise ValueError("unbalanced parenthesis")


def _split_top(s, sep):
    """Split at top-level `sep` characters (outside parentheses, brackets and string literals)."""
    out, depth, cur, j = [], 0, [], 0
    while j < len(s):
        c = s[j]
        if c == '"':
            k = j + 1
            while s[k] != '"':
                k += 2 if s[k] == "\\" else 1
            cur.append(s[j:k + 1])
            j = k + 1
            continue
        if c in "([":
            depth += 1
        elif c in ")]":
            depth -= 1
        if c == sep and depth == 0:
            out.append("".join(cur))
            cur = []
        else:
            cur.append(c)
        j += 1
    out.append("".join(cur))
    return out


def _operands(section):
    """'"+v"(amax), "v"(v[0])' -> ['amax', 'v[0]']"""
    ops = []
    for part in _split_top(section, ","):
        part = part.strip()
        if not part:
            continue
        m = re.match(r'"[^"]*"\s*\((.*)\)\s*$', part, re.S)
        if not m:
            raise ValueError(f"asm operand not understood: {part!r}")
        ops.append(m.group(1).strip())
    return ops


def _rewrite_asm(body, where):
    """body: the text between the parentheses of one asm statement -> replacement C expression statement (without ';')."""
    secs = _split_top(body, ":")
    tmpl = "".join(re.findall(r'"((?:[^"\\]|\\.)*)"', secs[0]))
    outs = _operands(secs[1]) if len(secs) > 1 else []
    ins = _operands(secs[2]) if len(secs) > 2 else []
    first = tmpl.strip().split()[0] if tmpl.strip() else ""
    if first == "s_waitcnt":
        m = re.match(r"s_waitcnt vmcnt\((%0|\d+)\)\s*$", tmpl.strip())
        if m:  # the explicit vector-memory waits: they decide when a deferred LDS-DMA lands (simt_emu.h, CBX_EMU_DMA=deferred)
            return f"(simt::g_dma_deferred ? simt::vm_wait({ins[0] if m.group(1) == '%0' else m.group(1)}) : (void)0)"
        assert re.match(r"s_waitcnt (lgkmcnt|vmcnt)\(", tmpl.strip()), (where, tmpl)
        return "((void)0)"
    if first in ("", "s_nop", ";"):
        return "((void)0)"
    if first == "v_max3_f32":
        assert re.match(r"v_max3_f32 %0, \|%1\|, \|%2\|, %0", tmpl), (where, tmpl)
        return f"({outs[0]} = simt_max3_abs({outs[0]}, {ins[0]}, {ins[1]}))"
    if first in ("v_fma_mixlo_f16", "v_fma_mixhi_f16"):
        hi = 1 if first.endswith("hi_f16") else 0
        m = re.match(r"v_fma_mix(?:lo|hi)_f16 %0, %1, (%2|-1\.0), (%2|%3) op_sel:\[(\d),0,0\] op_sel_hi:\[1,0,0\]", tmpl)
        assert m and int(m.group(3)) == hi, (where, tmpl)
        h = ins[0]
        if m.group(1) == "-1.0":
            s, t = "-1.0f", ins[1]
        else:
            s, t = ins[1], ins[2]
        return f"({outs[0]} = simt_fma_mix_f16({outs[0] if hi else '0u'}, {h}, {s}, {t}, {hi}))"
    if first in ("global_store_dword", "global_store_dwordx2", "global_store_dwordx4"):  # write-through stores (cbx_common.h cbx_store_out*): plain stores here
        assert re.match(r"global_store_dword(x2|x4)? %0, %1, off( sc0)?( sc1)?(\\n\\ts_nop \d)?\s*$", tmpl.strip()) and len(ins) == 2 and not outs, (where, tmpl)
        return f"simt_store_wt((void*)({ins[0]}), {ins[1]})"
    raise ValueError(f"{where}: asm statement without an emulation: {tmpl!r}")


def transform(src, name):
    out, i = [], 0
    pat = re.compile(r"\basm\b(\s+volatile)?\s*\(")
    while True:
        m = pat.search(src, i)
        if not m:
            out.append(src[i:])
            break
        out.append(src[i:m.start()])
        end = _match_paren(src, m.end() - 1)
        line = src.count("\n", 0, m.start()) + 1
        out.append(_rewrite_asm(src[m.end():end - 1], f"{name}:{line}"))
        i = end
    s = "".join(out)
    # extern __shared__ [attrs] T name[];  ->  T* name = (T*)simt::dyn_lds();
    s = re.sub(r"extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?([A-Za-z_][\w ]*?)\s+(\w+)\[\];",
               lambda m: f"{m.group(1)}* {m.group(2)} = ({m.group(1)}*)simt::dyn_lds();", s)
    assert "extern __shared__" not in s, f"{name}: an extern __shared__ declaration was not rewritten"
    return s


def _digest(files):
    h = hashlib.sha256()
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=False):
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    deps = srcs + [os.path.join(CSRC, "cbx_common.h"), os.path.join(ROOT, "include", "cbx.h"), os.path.join(HERE, "simt_emu.h"),
                   os.path.join(HERE, "simt_core.cpp"), os.path.abspath(__file__)]
    dig = _digest(deps)
    stamp = os.path.join(GEN, ".stamp")
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return LIB
    if not os.path.exists(CLANG):
        raise RuntimeError(f"{CLANG} not found (the emulator build needs ROCm's clang++ as an x86 host compiler)")
    os.makedirs(os.path.join(GEN, "csrc"), exist_ok=True)
    # keep the sources' relative include of ../../include/cbx.h working: _gen/csrc/x.hip -> _gen/../../include does not exist, so copy
    with open(os.path.join(ROOT, "include", "cbx.h")) as f:
        cbx_h = f.read()
    gen_common = transform(open(os.path.join(CSRC, "cbx_common.h")).read(), "cbx_common.h").replace('"../../include/cbx.h"', '"cbx.h"')
    with open(os.path.join(GEN, "csrc", "cbx_common.h"), "w") as f:
        f.write(gen_common)
    with open(os.path.join(GEN, "csrc", "cbx.h"), "w") as f:
        f.write(cbx_h)
    objs, procs = [], []
    for src in srcs:
        name = os.path.basename(src)
        gen = os.path.join(GEN, "csrc", name + ".cpp")
        with open(gen, "w") as f:
            f.write(transform(open(src).read(), name))
        obj = gen + ".o"
        objs.append(obj)
        cmd = [CLANG, *FLAGS, "-c", gen, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((name, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    core = os.path.join(GEN, "simt_core.o")
    procs.append(("simt_core.cpp", subprocess.Popen([CLANG, "-std=c++17", "-O2", "-fPIC", "-Wno-everything", *(["-fsanitize=address", "-g"] if ASAN else []), "-I", os.path.join(HERE, "shim"), "-I", HERE, "-c",
                                                     os.path.join(HERE, "simt_core.cpp"), "-o", core], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    fail = False
    for name, pr in procs:
        o, _ = pr.communicate()
        if pr.returncode != 0:
            fail = True
            sys.stderr.write(f"--- {name} failed:\n{o.decode()[:6000]}\n")
    if fail:
        raise RuntimeError("emulator build failed")
    subprocess.check_call([CLANG, "-shared", "-fPIC", *(["-fsanitize=address", "-shared-libasan"] if ASAN else []), *objs, core, "-o", LIB])
    with open(stamp, "w") as f:
        f.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
